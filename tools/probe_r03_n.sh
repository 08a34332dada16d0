#!/bin/bash
# shorter exact divisions by the loop-invariant diagonal: exhaustive comparison (div_lab) and what they are worth in the two-stage pass (pair_lab)
cd "$(dirname "$0")/.."
O=gpurun_out/probe_n; rm -rf $O; mkdir -p $O
timeout -k 10 120 tools/bin/div_lab > $O/div_lab.txt 2>&1 || exit 1
for b in pair_lab pair_lab_medium pair_lab_short pair_lab pair_lab_medium pair_lab_short; do
  echo "== $b" >> $O/pair.txt
  timeout -k 10 120 tools/bin/$b 512 30 0 0 0 0 1024x27 >> $O/pair.txt 2>&1 || exit 1
done
cat $O/div_lab.txt $O/pair.txt
