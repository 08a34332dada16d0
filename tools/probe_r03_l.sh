#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/probe_l
mkdir -p $O
step() {
  local name=$1 lim=$2; shift 2
  timeout -k 10 $lim "$@" > $O/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a $O/rc.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; tail -20 $O/$name.log; exit 1; fi
}
: > $O/rc.txt
step pytest_psor 600 python3 -m pytest tests -m gpu -q -p no:cacheprovider -k "psor or ragged or preconditioners_on_small or cli or block_local"
tail -8 $O/pytest_psor.log
step bench_psor 200 python3 bench.py --solver psor --steps 20 --warmup 4 --repeats 3 --no-cpu-baseline
step bench_psor64 200 python3 bench.py --solver psor --prec f64 --steps 20 --warmup 4 --repeats 3 --no-cpu-baseline
step bench_psor_maf 200 python3 bench.py --solver psor_maf --steps 20 --warmup 4 --repeats 3 --no-cpu-baseline
CZHIP_PSOR=0 step bench_psor_tiles 200 python3 bench.py --solver psor --steps 20 --warmup 4 --repeats 3 --no-cpu-baseline
for f in bench_psor bench_psor64 bench_psor_maf bench_psor_tiles; do python3 -c "import json; d=json.loads(open('$O/$f.log').read().strip().splitlines()[-1]); print('$f', round(d['value']), 'MLUPS', round(d['ms_per_step'],4), 'ms/sweep; kernel', round(d['roofline']['kernel_avg_ms'],4), d['roofline']['kernel_launches_timed'])"; done
