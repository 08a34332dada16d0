#!/bin/bash
# round-3 probe A (GPU box, repo root): CU-mask behaviour + first multi-rank RCCL run on one GPU
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/probe_a
mkdir -p $O
ip addr show lo > $O/lo.txt 2>&1
timeout -k 10 120 tools/bin/cumask_lab map > $O/cumask_map.txt 2>&1; echo "map rc=$?" | tee -a $O/rc.txt
timeout -k 10 60 tools/bin/cumask_lab reserve 2 > $O/cumask_reserve2.txt 2>&1; echo "reserve rc=$?" | tee -a $O/rc.txt
timeout -k 10 60 tools/bin/cumask_lab reserve 4 >> $O/cumask_reserve2.txt 2>&1
timeout -k 10 120 tools/bin/cumask_lab stream 6 > $O/cumask_stream.txt 2>&1; echo "stream rc=$?" | tee -a $O/rc.txt
tail -3 $O/cumask_map.txt; cat $O/cumask_reserve2.txt $O/cumask_stream.txt
NCCL_DEBUG=INFO timeout -k 10 400 python3 -m pytest tests/test_gpu_rccl.py -x -q -k "jacobi_f32_1x2x1" > $O/rccl_first.log 2>&1; echo "rccl first rc=$?" | tee -a $O/rc.txt
tail -40 $O/rccl_first.log
