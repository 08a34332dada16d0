#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/probe_ag; rm -rf $O; mkdir -p $O
for map in 1 0 1 0; do
for p in f32 f64; do
for n in 64 96 128 160 192 224; do
    CZHIP_T2_MAP=$map timeout -k 10 100 python3 bench.py --cells $n --solver jacobi --prec $p --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --settle 0.05 > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; exit 1; }
    python3 -c "
import json;d=json.load(open('$O/b.json'))
print('map $map %4d^3 $p jacobi %9.0f MLUPS  %.4f ms/step  kernel %.4f ms per pass' % ($n, d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))" | tee -a $O/times.txt
done
done
done
