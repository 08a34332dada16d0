#!/bin/bash
# GPU-side cost of the exchange kernels per face orientation (two 512^3 ranks as threads of one process on ONE GPU, LOCAL transport)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/comm_cost
mkdir -p $O
for d in "2 1 1" "1 2 1" "1 1 2"; do
  t=$(echo $d | tr -d ' ')
  rocprofv3 --kernel-trace -d $O/kt_$t -- python3 tools/comm_kernels_cost.py $d > $O/run_$t.log 2>&1 || { tail -5 $O/run_$t.log; exit 1; }
  db=$(ls $O/kt_$t/*/*.db | head -1)
  echo "== division $d" >> $O/summary.txt
  python3 tools/rocprof_kernels.py $db | grep -E "kernel|box_copy|pair_shell|jacobi2p|copyBuffer" >> $O/summary.txt
done
cat $O/summary.txt
