#!/bin/bash
# BiCGSTAB with alpha / omega kept on the device (one host wait per iteration instead of three): parity, then time per iteration over sizes
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/probe_am; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "bicg or precond or decomp or rccl or cli or smoke" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -3 $O/pytest.log
grep -q "rc=0" $O/rc.txt || exit 1
for r in 1 2; do
for n in 64 128 256 512; do
  for f in 1 0; do
    CZ_BICG_FUSE=$f timeout -k 10 200 python3 bench.py --cells $n --solver pbicgstab --prec f64 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; exit 1; }
    python3 -c "
import json;d=json.load(open('$O/b.json'))
print('%4d^3 f64 bicgstab+jacobi CZ_BICG_FUSE=$f %.4f ms per iteration' % ($n, d['ms_per_step']))" | tee -a $O/times.txt
  done
done
done
