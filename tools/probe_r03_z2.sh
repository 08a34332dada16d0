#!/bin/bash
# every vector kernel on rows that are no multiple of the vector width: full GPU suite, then rates against the old rule (CZHIP_T2_ROWS=0)
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/probe_z2; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -5 $O/pytest.log
grep -q "rc=0" $O/rc.txt || exit 1
for n in 511 510 509 512; do
  for rows in 1 0; do
    CZHIP_T2_ROWS=$rows timeout -k 10 200 python3 bench.py --cells $n --solver pbicgstab --prec f64 --steps 10 --warmup 2 --repeats 2 --no-cpu-baseline > $O/b.json 2>$O/b.err || exit 1
    python3 -c "
import json;d=json.load(open('$O/b.json'))
print('%4d^3 f64 bicgstab CZHIP_T2_ROWS=$rows %.3f ms per iteration' % ($n, d['ms_per_step']))" | tee -a $O/times.txt
    CZHIP_T2_ROWS=$rows timeout -k 10 200 python3 bench.py --cells $n --solver pbicgstab --prec f32 --steps 10 --warmup 2 --repeats 2 --no-cpu-baseline > $O/b.json 2>$O/b.err || exit 1
    python3 -c "
import json;d=json.load(open('$O/b.json'))
print('%4d^3 f32 bicgstab CZHIP_T2_ROWS=$rows %.3f ms per iteration' % ($n, d['ms_per_step']))" | tee -a $O/times.txt
  done
done
for n in 511 255 127; do
  for s in jacobi sor2sma jacobi_maf; do
    for rows in 1 0; do
      CZHIP_T2_ROWS=$rows timeout -k 10 120 python3 bench.py --cells $n --solver $s --steps 40 --warmup 6 --repeats 3 --no-cpu-baseline --settle 0.05 > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; exit 1; }
      python3 -c "
import json;d=json.load(open('$O/b.json'))
print('%4d^3 f32 %-10s CZHIP_T2_ROWS=$rows %9.0f MLUPS  %.4f ms/step  %s' % ($n, '$s', d['value'], d['ms_per_step'], d['roofline']['kernel'][:28]))" | tee -a $O/times.txt
    done
  done
done
