#!/bin/bash
# round-3 probe C (GPU box, repo root)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/probe_c
mkdir -p $O
step() {  # name, seconds, command...: a step that runs into its time limit ends the call (no further GPU work after a hang)
  local name=$1 lim=$2; shift 2
  timeout -k 10 $lim "$@" > $O/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a $O/rc.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; tail -20 $O/$name.log; exit 1; fi
}
: > $O/rc.txt
step pytest_lines 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -q -p no:cacheprovider -k "length_limit or few_resident or cu_reservation or long_lines or void"
tail -15 $O/pytest_lines.log
for v in pair_lab pair_lab_nores pair_lab_grp; do
  step lab_$v 120 tools/bin/$v 512 30 0 0 0 0 1024x27 512x30
  step lab_rb_$v 120 tools/bin/$v 512 30 1 0 0 0 1024x27
done
grep -h "tj" $O/lab_*.log
step pcr_ring_default 200 python3 bench.py --solver pcr --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline
CZHIP_PCR_SLOTS=8 step pcr_ring_8 200 python3 bench.py --solver pcr --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline
for f in pcr_ring_default pcr_ring_8; do python3 -c "import json,sys; d=json.loads(open('$O/$f.log').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
step tl_a 330 tools/rccl_overlap_timeline.sh a_k0_f32_1x2x1 0 f32 1 2 1
step tl_b 330 tools/rccl_overlap_timeline.sh b_k2_f32_1x2x1 2 f32 1 2 1
step tl_c 330 tools/rccl_overlap_timeline.sh c_k2_f32_2x1x1 2 f32 2 1 1
step tl_d 330 tools/rccl_overlap_timeline.sh d_k2_f64_1x1x2 2 f64 1 1 2
step tl_e 330 tools/rccl_overlap_timeline.sh e_k0_f64_1x1x2 0 f64 1 1 2
for t in a b c d e; do echo "=== tl_$t"; head -40 $O/tl_$t.log; done
