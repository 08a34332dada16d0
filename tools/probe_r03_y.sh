#!/bin/bash
# sizes whose padded k extent is not a multiple of the vector width: what the sweeps fall back to and what it costs
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/probe_y; rm -rf $O; mkdir -p $O
for n in 512 511 510 509 508 500 384 383; do
  for s in jacobi sor2sma; do
    timeout -k 10 120 python3 bench.py --cells $n --solver $s --steps 40 --warmup 6 --repeats 3 --no-cpu-baseline --settle 0.05 > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; exit 1; }
    python3 -c "
import json;d=json.load(open('$O/b.json'))
print('%4d^3 f32 %-8s %9.0f MLUPS  %.4f ms/step  kernel: %s' % ($n, '$s', d['value'], d['ms_per_step'], d['roofline']['kernel'][:40]))" | tee -a $O/times.txt
  done
done
for n in 511 510 509; do
  timeout -k 10 120 python3 bench.py --cells $n --solver jacobi --prec f64 --steps 40 --warmup 6 --repeats 3 --no-cpu-baseline --settle 0.05 > $O/b.json 2>$O/b.err || exit 1
  python3 -c "
import json;d=json.load(open('$O/b.json'))
print('%4d^3 f64 jacobi   %9.0f MLUPS  %.4f ms/step  kernel: %s' % ($n, d['value'], d['ms_per_step'], d['roofline']['kernel'][:40]))" | tee -a $O/times.txt
done
