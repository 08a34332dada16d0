#!/bin/bash
# FETCH_SIZE per access shape against a known byte count (tools/fetch_lab.hip) -> one table
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/fetch_lab
rm -rf $O; mkdir -p $O
tools/bin/fetch_lab 6 > $O/timing.txt 2>&1 || { cat $O/timing.txt; exit 1; }
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc --output-format csv -- tools/bin/fetch_lab 2 > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/pmc2 --output-format csv -- tools/bin/fetch_lab 2 > $O/pmc2.log 2>&1 || echo "(TCC_EA0_RDREQ pass failed)"
python3 - "$O" <<'PY' | tee $O/table.txt
import csv, glob, collections, sys
O = sys.argv[1]
known = 516 ** 3 * 4
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("bytes read by every kernel (known): %d = %.1f MB" % (known, known / 1e6))
print("%-28s %14s %10s   %s" % ("kernel", "FETCH_SIZE KiB", "x bytes", "other counters (mean)"))
for k in sorted(acc):
    fs = acc[k].get("FETCH_SIZE", [])
    m = sum(fs) / len(fs) if fs else float("nan")
    other = {c: sum(v) / len(v) for c, v in acc[k].items() if c != "FETCH_SIZE"}
    print("%-28s %14.1f %10.3f   %s" % (k, m, m * 1024 / known, other))
PY
cat $O/timing.txt
rm -rf $O/pmc $O/pmc2
