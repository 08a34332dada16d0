#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/probe_j
rm -rf $O; mkdir -p $O
for pc in 1 2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 120 rocprofv3 --pmc $c -d $O/pmc_${c}_$pc --output-format csv -- tools/bin/psor_lab 512 512 512 2 0 $pc > $O/log_${c}_$pc.txt 2>&1
    echo "rc $? $c $pc"; ls $O/pmc_${c}_$pc/* | head -3
  done
done
python3 - "$O" <<'PY' | tee $O/traffic.txt
import csv, glob, collections, sys
O = sys.argv[1]
for pc in (1, 2):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{O}/pmc_{c}_{pc}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                acc["psor_col_k" if "psor_col_k" in r["Kernel_Name"] else "psor_tile_k" if "psor_tile_k" in r["Kernel_Name"] else "other"].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if "psor" in k:
                print(f"wg/cu {pc} {c:10s} {k:14s} mean {sum(v)/len(v)/1024:10.1f} MiB per launch, {sum(v)/1024:10.1f} MiB over {len(v)} launches")
PY
rm -rf $O/pmc_*
