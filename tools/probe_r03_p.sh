#!/bin/bash
# red-black line SOR (pcr_line_reg_k) with the mask kept as bits between source term and relaxation: parity tests, time, HBM reads
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/probe_p; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "pcr or line" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -3 $O/pytest.log
grep -q "rc=0" $O/rc.txt || exit 1
for r in 1 2; do
  for s in pcr_rb pcr_rb_esa pcr_j_esa pcr; do
    timeout -k 10 120 python3 bench.py --solver $s --steps 20 --warmup 4 --repeats 3 --no-cpu-baseline > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; exit 1; }
    python3 -c "import json;d=json.load(open('$O/b.json'));print('%-10s f32 %8.0f MLUPS  %.4f ms per iteration'%('$s',d['value'],d['ms_per_step']))" | tee -a $O/times.txt
  done
  timeout -k 10 120 python3 bench.py --solver pcr_rb --prec f64 --steps 20 --warmup 4 --repeats 3 --no-cpu-baseline > $O/b.json 2>$O/b.err || exit 1
  python3 -c "import json;d=json.load(open('$O/b.json'));print('pcr_rb     f64 %8.0f MLUPS  %.4f ms per iteration'%(d['value'],d['ms_per_step']))" | tee -a $O/times.txt
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_$c --output-format csv -- python3 bench.py --solver pcr_rb --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --settle 0 > $O/pmc.log 2>&1 || { tail -3 $O/pmc.log; exit 1; }
done
python3 - "$O" <<'PY' | tee -a $O/traffic.txt
import csv, glob, sys
O = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(f"{O}/pmc_{c}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "pcr_line_reg_k" in r["Kernel_Name"]: v.append(float(r["Counter_Value"]))
    print("%s pcr_line_reg_k mean %.1f MiB raw per launch (n=%d)%s" % (c, sum(v) / len(v) / 1024, len(v), "  (x2 for bytes)" if c == "FETCH_SIZE" else ""))
PY
python3 tools/summarize_pmc.py pcr_rb_512_f32 pcr_line_reg_k $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/hbm_traffic.json > $O/hbm.txt 2>&1 || true
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
