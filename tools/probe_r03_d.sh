#!/bin/bash
# round-3 probe D (GPU box, repo root): whole GPU suite on the refactored driver + RCCL timelines with / without the CU reservation
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/probe_d
mkdir -p $O
step() {
  local name=$1 lim=$2; shift 2
  timeout -k 10 $lim "$@" > $O/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a $O/rc.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; tail -20 $O/$name.log; exit 1; fi
}
: > $O/rc.txt
step pytest_gpu 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider
tail -12 $O/pytest_gpu.log
step tl_a 330 tools/rccl_overlap_timeline.sh a_k0_f32_1x2x1 0 f32 1 2 1
step tl_b 330 tools/rccl_overlap_timeline.sh b_k2_f32_1x2x1 2 f32 1 2 1
step tl_c 330 tools/rccl_overlap_timeline.sh c_k2_f32_2x1x1 2 f32 2 1 1
step tl_d 330 tools/rccl_overlap_timeline.sh d_k2_f64_1x1x2 2 f64 1 1 2
step tl_e 330 tools/rccl_overlap_timeline.sh e_k0_f64_1x1x2 0 f64 1 1 2
step tl_f 330 tools/rccl_overlap_timeline.sh f_k4_f32_1x2x1 4 f32 1 2 1
for t in a b c d e f; do echo "=== tl_$t"; grep -v "^W2026\|simple_timer" $O/tl_$t.log | head -42 | cut -c1-200; done
du -sh gpurun_out
