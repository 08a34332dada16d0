#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/probe_k
mkdir -p $O
(for b in psor_lab psor_lab_g8; do for a in "70 50 40 2 0" "36 52 20 2 0" "256 256 256 3 0" "512 512 512 4 0"; do timeout -k 5 60 tools/bin/$b $a || exit 1; done; done) 2>&1 | tee $O/psor_g8.log
