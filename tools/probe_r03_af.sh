#!/bin/bash
# balanced XCD map also for fewer than eight segments: parity tests, non-cubic boxes and small cubes again
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/probe_af; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "fused or two_stage or pair or rows or made or jacobi2 or rbsor2 or decomp or stationary or zero or bicg or full" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -3 $O/pytest.log
grep -q "rc=0" $O/rc.txt || exit 1
cd $O
for a in "1024 64 64" "64 1024 64" "64 64 1024" "512 512 16" "512 16 512" "16 512 512" "2000 40 40" "40 2000 40" "37 1000 53" "1000 37 53" "640 480 24" "300 200 100" "64 64 64" "96 96 96" "128 128 128" "192 192 192" "256 256 256" "384 384 384" "512 512 512"; do
  for p in f32 f64; do
    r=$(timeout -k 10 100 ../../cubez_amd/cz_$p $a jacobi 200 0.8 2>&1 | grep "GPU time" | sed -e 's/.*GPU time = //')
    echo "cz_$p $a jacobi 200 0.8   $r" | tee -a times.txt
  done
done
