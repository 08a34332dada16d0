#!/bin/bash
# kernel trace of a few lexicographic sweeps (psor, pcr) at 512^3 FP32: per-launch durations in launch order
# usage (GPU box, repo root): tools/trace_lex.sh
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/lex
mkdir -p $O
for s in psor; do
  rocprofv3 --kernel-trace -d $O/kt_$s --output-format csv -- python3 bench.py --solver $s --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > $O/kt_$s.log 2>&1 || { tail -5 $O/kt_$s.log; exit 1; }
  python3 - $O/kt_$s > $O/launches_$s.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows[-140:]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f us  gap %6.1f  dur %7.1f us  grid %s wg %s  %s" % ((st - t0) / 1e3, (st - prev_end) / 1e3, (en - st) / 1e3, r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "?"), r.get("Workgroup_Size_X", "?"), r["Kernel_Name"][:50]))
    prev_end = en
PY
  tail -110 $O/launches_$s.txt
done
