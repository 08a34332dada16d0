#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/probe_f
mkdir -p $O
step() {
  local name=$1 lim=$2; shift 2
  timeout -k 10 $lim "$@" > $O/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a $O/rc.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; tail -20 $O/$name.log; exit 1; fi
}
: > $O/rc.txt
step pytest_comm 900 python3 -m pytest tests/test_gpu_decomp.py tests/test_gpu_rccl.py tests/test_gpu_kernels.py -q -p no:cacheprovider -k "not vs_oracle and not golden"
tail -5 $O/pytest_comm.log
step cu_cost 900 python3 tools/cu_reserve_cost.py
grep -v "^{" $O/cu_cost.log
step tl_g 330 tools/rccl_overlap_timeline.sh g_soft2_f32_1x2x1 2 f32 1 2 1
step tl_h 330 tools/rccl_overlap_timeline.sh h_soft2_f64_1x1x2 2 f64 1 1 2
for t in g h; do echo "=== tl_$t"; grep -v "^W2026\|simple_timer" $O/tl_$t.log | head -42 | cut -c1-200; done
step profile 900 tools/profile_r03.sh
tail -30 $O/profile.log | cut -c1-600
du -sh gpurun_out
