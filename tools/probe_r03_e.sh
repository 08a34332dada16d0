#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/probe_e
mkdir -p $O
step() {
  local name=$1 lim=$2; shift 2
  timeout -k 10 $lim "$@" > $O/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a $O/rc.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; tail -20 $O/$name.log; exit 1; fi
}
: > $O/rc.txt
step fetch_lab 300 tools/fetch_lab_run.sh
cat gpurun_out/fetch_lab/table.txt
step cu_cost 300 python3 tools/cu_reserve_cost.py
grep -v "^{" $O/cu_cost.log
step profile 900 tools/profile_r03.sh
tail -30 $O/profile.log | cut -c1-400
du -sh gpurun_out
