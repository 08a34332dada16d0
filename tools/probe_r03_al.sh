#!/bin/bash
# the 8f solvers over small and non-cubic boxes through the CLI (GPU time per iteration), looking for holes
cd "$(dirname "$0")/.."
O=gpurun_out/probe_al; rm -rf $O; mkdir -p $O; cd $O
for a in "64 64 64" "128 128 128" "256 256 256" "300 200 100" "100 200 300" "1024 64 64" "64 1024 64" "40 2000 40" "512 512 16" "511 255 127"; do
  for s in "psor 40 1.2" "pcr_rb 40 1.2" "pcr 20 1.2" "pcr_j_esa 40 0.9" "sor2sma 40 1.5" "jacobi 40 0.8"; do
    set -- $s
    r=$(timeout -k 10 100 ../../cubez_amd/cz_f32 $a $s 2>&1 | grep -E "Iter =|GPU time" | tr '\n' ' ' | sed -e 's/=====*//g')
    echo "cz_f32 $a $s | $r" | tee -a times.txt
  done
done
