#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/probe_ai; rm -rf $O; mkdir -p $O
for pc in jacobi sor2sma none; do
  for p in f64 f32; do
    timeout -k 10 200 python3 bench.py --solver pbicgstab --precond $pc --prec $p --steps 10 --warmup 2 --repeats 2 --no-cpu-baseline > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; continue; }
    python3 -c "
import json;d=json.load(open('$O/b.json'))
print('512^3 $p pbicgstab + %-8s %.3f ms per iteration' % ('$pc', d['ms_per_step']))" | tee -a $O/times.txt
  done
done
for s in jacobi sor2sma jacobi_maf sor2sma_maf; do
  timeout -k 10 200 python3 bench.py --solver $s --prec f64 --steps 40 --warmup 6 --repeats 3 --no-cpu-baseline > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; continue; }
  python3 -c "
import json;d=json.load(open('$O/b.json'))
print('512^3 f64 %-12s %9.0f MLUPS  %.4f ms/step  kernel %.4f' % ('$s', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))" | tee -a $O/times.txt
done
