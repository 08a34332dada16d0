#!/bin/bash
# nontemporal output stores of jacobi2p_k at sizes that fit the caches (pair_lab)
cd "$(dirname "$0")/.."
O=gpurun_out/probe_s; rm -rf $O; mkdir -p $O
for r in 1 2; do
for n in 64 128 192 256 384; do
for b in pair_lab pair_lab_nts; do
  echo "== $b $n" >> $O/pair.txt
  timeout -k 10 120 tools/bin/$b $n 200 0 0 0 0 512x16 1024x16 2>&1 | grep "tj" >> $O/pair.txt || exit 1
done
done
done
cat $O/pair.txt
