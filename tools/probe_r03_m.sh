#!/bin/bash
# BiCGSTAB 512^3 FP64: the launches of one iteration in order, with the idle gaps between them (rocprofv3 --kernel-trace + --memory-copy-trace)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/probe_m
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt --output-format csv -- python3 bench.py --solver pbicgstab --prec f64 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --settle 0 > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
ev = []
for f in glob.glob(O + "/kt/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:46]))
for f in glob.glob(O + "/kt/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# the last complete iteration: from the second-to-last ewise_k<2, 1> (bicg_1) to the last one
idx = [i for i, e in enumerate(ev) if e[2].startswith("ewise_k<2, 1>")]
a, b = idx[-2], idx[-1]
t0 = ev[a][0]
prev_end = None
busy = 0
with open(O + "/bicgstab_iteration_timeline.txt", "w") as o:
    o.write("one BiCGSTAB iteration, 512^3 FP64 (bicg_1 to the next bicg_1): start [us], duration [us], idle gap before [us], launch\n")
    for s, e, n in ev[a:b]:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        o.write("%9.1f %8.1f %7.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
        prev_end = max(prev_end, e) if prev_end else e
        busy += e - s
    tot = ev[b][0] - t0
    o.write("iteration %.1f us, launches busy %.1f us, idle %.1f us (%.1f %%), %d launches\n" % (tot / 1e3, busy / 1e3, (tot - busy) / 1e3, 100.0 * (tot - busy) / tot, b - a))
print(open(O + "/bicgstab_iteration_timeline.txt").read())
PY
rm -rf $O/kt
tail -2 $O/bench.log | cut -c1-300
