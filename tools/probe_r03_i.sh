#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/probe_i
mkdir -p $O
for a in "18 18 512 3 0" "34 18 512 3 0" "130 18 512 3 0" "34 34 512 3 0" "70 50 40 2 0" "70 50 40 2 1" "128 128 128 3 0" "256 256 256 3 0" "512 512 512 4 0 2" "512 512 512 4 0 3" "512 512 512 3 1 2"; do
  timeout -k 10 60 tools/bin/psor_lab $a || exit 1
done 2>&1 | tee $O/psor_chain.log
for a in "41 37 45 2 0" "18 18 512 3 0" "512 512 512 3 0 2"; do
  timeout -k 10 60 tools/bin/psor_lab64 $a || exit 1
done 2>&1 | tee -a $O/psor_chain.log
