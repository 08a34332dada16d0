#!/bin/bash
# the 8f solvers at a k extent that is no multiple of the vector width, and configs[0] (128^3 FP64 BiCGSTAB to convergence) through the CLI
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/probe_ab; rm -rf $O; mkdir -p $O
for n in 512 511; do
  for s in pcr_rb pcr_rb_esa pcr_j_esa pcr psor psor_maf pcr_rb_maf sor2sma_maf; do
    timeout -k 10 200 python3 bench.py --cells $n --solver $s --steps 12 --warmup 3 --repeats 2 --no-cpu-baseline --settle 0 > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; exit 1; }
    python3 -c "
import json;d=json.load(open('$O/b.json'))
print('%4d^3 f32 %-12s %9.0f MLUPS  %.4f ms/step' % ($n, '$s', d['value'], d['ms_per_step']))" | tee -a $O/times.txt
  done
done
cd $O
for a in "128 128 128 pbicgstab 1000 0.8 jacobi" "127 127 127 pbicgstab 1000 0.8 jacobi" "64 64 64 jacobi 100000 0.8" "128 128 128 sor2sma 100000 1.5"; do
  echo "== cz_f64 $a" | tee -a cli.txt
  timeout -k 10 200 ../../cubez_amd/cz_f64 $a 2>&1 | grep -E "Iter =|GPU time|Error max" | tee -a cli.txt
done
