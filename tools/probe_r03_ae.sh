#!/bin/bash
# non-cubic boxes through the CLI: the two-stage pass (default) against single sweeps (CZHIP_T2=0)
cd "$(dirname "$0")/.."
O=gpurun_out/probe_ae; rm -rf $O; mkdir -p $O; cd $O
for a in "1024 64 64" "64 1024 64" "64 64 1024" "256 256 1024" "1024 256 256" "256 1024 256" "512 512 16" "512 16 512" "16 512 512" "2000 40 40" "40 40 2000" "40 2000 40" "300 200 100" "100 200 300" "37 1000 53" "1000 37 53" "640 480 24" "48 48 3000"; do
  for p in f32 f64; do
    for t2 in 1 0; do
      r=$(CZHIP_T2=$t2 timeout -k 10 100 ../../cubez_amd/cz_$p $a jacobi 200 0.8 2>&1 | grep "GPU time" | sed -e 's/.*GPU time = //')
      echo "cz_$p $a jacobi 200 0.8  CZHIP_T2=$t2  $r" | tee -a times.txt
    done
  done
done
