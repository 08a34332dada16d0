#!/bin/bash
# jacobi2p_k: nontemporal stores of the output / nontemporal loads of the right-hand side (pair_lab, random fields)
cd "$(dirname "$0")/.."
O=gpurun_out/probe_r; rm -rf $O; mkdir -p $O
for r in 1 2 3; do
for b in pair_lab pair_lab_nts pair_lab_ntb pair_lab_ntsb; do
  echo "== $b" >> $O/pair.txt
  timeout -k 10 120 tools/bin/$b 512 30 0 0 0 0 1024x27 2>&1 | grep "tj" >> $O/pair.txt || exit 1
  timeout -k 10 120 tools/bin/$b 512 30 1 0 0 0 1024x27 2>&1 | grep "tj" >> $O/pair.txt || exit 1
done
done
cat $O/pair.txt
