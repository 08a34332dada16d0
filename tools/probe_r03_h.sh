#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/probe_h
mkdir -p $O
step() {
  local name=$1 lim=$2; shift 2
  timeout -k 10 $lim "$@" > $O/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a $O/rc.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; tail -20 $O/$name.log; exit 1; fi
}
: > $O/rc.txt
step psor_a 60 tools/bin/psor_lab 40 36 44 2 0
step psor_b 60 tools/bin/psor_lab 70 50 40 2 0
step psor_c 60 tools/bin/psor_lab 70 50 40 2 1
step psor_d 60 tools/bin/psor_lab64 41 37 45 2 0
step psor_e 60 tools/bin/psor_lab 128 128 128 3 0
step psor_f 90 tools/bin/psor_lab 512 512 512 4 0
step psor_g 90 tools/bin/psor_lab 512 512 512 4 0 2
step psor_h 90 tools/bin/psor_lab 512 512 512 4 0 8
step psor_i 90 tools/bin/psor_lab64 512 512 512 3 0
step psor_j 90 tools/bin/psor_lab 512 512 512 3 1
step psor_k 90 tools/bin/psor_lab 256 256 256 4 0
cat $O/psor_*.log
