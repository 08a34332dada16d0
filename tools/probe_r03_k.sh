#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/probe_k
mkdir -p $O
(for a in "36 52 20" "18 18 18" "33 17 50" "20 20 20" "36 36 34" "36 36 20" "24 20 36"; do tools/bin/psor_lab $a 2 0; done; tools/bin/psor_lab64 36 52 20 2 0; tools/bin/psor_lab64 24 20 36 2 0) 2>&1 | tee $O/psor_small.log
