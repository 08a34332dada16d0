#!/bin/bash
# the two-stage pass on rows that are no multiple of the vector width: parity tests, then rates against the old rule (CZHIP_T2_ROWS=0)
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/probe_z; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "fused or two_stage or pair or rows or made or jacobi2 or rbsor2 or decomp or stationary or zero" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -5 $O/pytest.log
grep -q "rc=0" $O/rc.txt || exit 1
for n in 512 511 510 509; do
  for rows in 1 0; do
    for s in jacobi sor2sma; do
      CZHIP_T2_ROWS=$rows timeout -k 10 120 python3 bench.py --cells $n --solver $s --steps 40 --warmup 6 --repeats 3 --no-cpu-baseline --settle 0.05 > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; exit 1; }
      python3 -c "
import json;d=json.load(open('$O/b.json'))
print('%4d^3 f32 %-8s CZHIP_T2_ROWS=$rows %9.0f MLUPS  %.4f ms/step  %s' % ($n, '$s', d['value'], d['ms_per_step'], d['roofline']['kernel'][:28]))" | tee -a $O/times.txt
    done
  done
done
for n in 511 509; do
  for rows in 1 0; do
    CZHIP_T2_ROWS=$rows timeout -k 10 120 python3 bench.py --cells $n --solver jacobi --prec f64 --steps 40 --warmup 6 --repeats 3 --no-cpu-baseline --settle 0.05 > $O/b.json 2>$O/b.err || exit 1
    python3 -c "
import json;d=json.load(open('$O/b.json'))
print('%4d^3 f64 jacobi   CZHIP_T2_ROWS=$rows %9.0f MLUPS  %.4f ms/step  %s' % ($n, d['value'], d['ms_per_step'], d['roofline']['kernel'][:28]))" | tee -a $O/times.txt
    CZHIP_T2_ROWS=$rows timeout -k 10 200 python3 bench.py --cells $n --solver pbicgstab --prec f64 --steps 10 --warmup 2 --repeats 2 --no-cpu-baseline > $O/b.json 2>$O/b.err || exit 1
    python3 -c "
import json;d=json.load(open('$O/b.json'))
print('%4d^3 f64 bicgstab CZHIP_T2_ROWS=$rows %.3f ms per iteration' % ($n, d['ms_per_step']))" | tee -a $O/times.txt
  done
done
