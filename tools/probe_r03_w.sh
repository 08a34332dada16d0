#!/bin/bash
# BiCGSTAB with the vector updates made inside the first preconditioner pass: parity, then time with and without
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/probe_w; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "bicg or precond" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -5 $O/pytest.log
grep -q "rc=0" $O/rc.txt || exit 1
for r in 1 2 3; do
  for f in 1 0; do
    CZ_BICG_FUSE=$f timeout -k 10 200 python3 bench.py --solver pbicgstab --prec f64 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; exit 1; }
    python3 -c "import json;d=json.load(open('$O/b.json'));print('CZ_BICG_FUSE=$f  %.3f ms per iteration  all %s'%(d['ms_per_step'],[round(x,3) for x in d['ms_per_step_all']]))" | tee -a $O/times.txt
  done
done
CZ_BICG_FUSE=1 timeout -k 10 200 python3 bench.py --solver pbicgstab --prec f32 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline > $O/b.json 2>$O/b.err || exit 1
python3 -c "import json;d=json.load(open('$O/b.json'));print('f32 fused %.3f ms per iteration'%d['ms_per_step'])" | tee -a $O/times.txt
CZ_BICG_FUSE=0 timeout -k 10 200 python3 bench.py --solver pbicgstab --prec f32 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline > $O/b.json 2>$O/b.err || exit 1
python3 -c "import json;d=json.load(open('$O/b.json'));print('f32 unfused %.3f ms per iteration'%d['ms_per_step'])" | tee -a $O/times.txt
