#!/bin/bash
# chunk lengths from 2 planes up in the launch model: parity tests, then the size sweep again
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/probe_ad; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "fused or two_stage or pair or rows or made or jacobi2 or rbsor2 or decomp or stationary or zero or bicg or full" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -3 $O/pytest.log
grep -q "rc=0" $O/rc.txt || exit 1
for p in f32 f64; do
for n in 64 96 128 160 192 256 384 512; do
  for s in jacobi sor2sma; do
    timeout -k 10 100 python3 bench.py --cells $n --solver $s --prec $p --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --settle 0.05 > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; exit 1; }
    python3 -c "
import json;d=json.load(open('$O/b.json'))
print('%4d^3 $p %-8s %9.0f MLUPS  %.4f ms/step  kernel %.4f ms per pass' % ($n, '$s', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))" | tee -a $O/times.txt
  done
done
done
for a in "128 128 128 pbicgstab 1000 0.8 jacobi" "64 64 64 jacobi 100000 0.8" "128 128 128 sor2sma 100000 1.5"; do
  echo "== cz_f64 $a" | tee -a $O/times.txt
  (cd $O && timeout -k 10 200 ../../cubez_amd/cz_f64 $a 2>&1 | grep -E "Iter =|GPU time") | tee -a $O/times.txt
done
