#!/bin/bash
# red-black line SOR (pcr_line_reg_k): 2-D tiles of lines per workgroup step against runs along i -- parity tests, time, HBM reads
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/probe_o; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "pcr or line" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -3 $O/pytest.log
grep -q "rc=0" $O/rc.txt || exit 1
for t in 1 0 1 0; do
  for s in pcr_rb pcr_rb_esa pcr_j_esa; do
    CZHIP_PCR_TILE=$t timeout -k 10 120 python3 bench.py --solver $s --steps 20 --warmup 4 --repeats 3 --no-cpu-baseline > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; exit 1; }
    python3 -c "import json;d=json.load(open('$O/b.json'));print('tile $t %-10s f32 %8.0f MLUPS  %.4f ms per iteration'%('$s',d['value'],d['ms_per_step']))" | tee -a $O/times.txt
  done
  CZHIP_PCR_TILE=$t timeout -k 10 120 python3 bench.py --solver pcr_rb --prec f64 --steps 20 --warmup 4 --repeats 3 --no-cpu-baseline > $O/b.json 2>$O/b.err || exit 1
  python3 -c "import json;d=json.load(open('$O/b.json'));print('tile $t pcr_rb     f64 %8.0f MLUPS  %.4f ms per iteration'%(d['value'],d['ms_per_step']))" | tee -a $O/times.txt
done
for t in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    CZHIP_PCR_TILE=$t rocprofv3 --pmc $c -d $O/pmc_${t}_$c --output-format csv -- python3 bench.py --solver pcr_rb --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --settle 0 > $O/pmc.log 2>&1 || { tail -3 $O/pmc.log; exit 1; }
  done
  python3 - "$O" $t <<'PY' | tee -a $O/traffic.txt
import csv, glob, sys
O, t = sys.argv[1], sys.argv[2]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(f"{O}/pmc_{t}_{c}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "pcr_line_reg_k" in r["Kernel_Name"]: v.append(float(r["Counter_Value"]))
    print("tile %s %s pcr_line_reg_k mean %.1f MiB raw per launch (n=%d)%s" % (t, c, sum(v) / len(v) / 1024, len(v), "  (x2 for bytes)" if c == "FETCH_SIZE" else ""))
PY
  rm -rf $O/pmc_${t}_*
done
