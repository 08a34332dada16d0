#!/bin/bash
# round-2 kernel lab: TJ sweeps + PMC passes of jacobi2_k vs jacobi2p_k (run on the GPU box from the repo root)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/lab2
mkdir -p $O
L=./tools/bin/pair_lab
timeout -k 10 120 $L 512 30 0 0 0 0 512x16 512x18 512x22 512x23 512x26 512x28 512x30 512x32 512x34 512x43 1024x16 1024x19 1024x22 1024x27 1024x32 1024x43 1024x64 > $O/sweep_f32.txt 2>&1 || exit 1
timeout -k 10 120 $L 512 30 1 0 0 0 512x16 512x30 1024x27 > $O/sweep_rb_f32.txt 2>&1 || exit 1
timeout -k 10 120 ./tools/bin/pair_lab64 512 20 0 0 0 0 1024x16 1024x32 1024x64 1024x43 1024x27 > $O/sweep_f64.txt 2>&1 || exit 1
timeout -k 10 120 $L 256 50 0 200 180 252 512x16 512x30 > $O/sweep_small.txt 2>&1 || exit 1
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $pass -d $O/pmc_$tag --output-format csv -- $L 512 4 0 0 0 0 512x16 > $O/pmc_$tag.log 2>&1 || { echo "pmc $tag failed"; tail -5 $O/pmc_$tag.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/lab2/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "jacobi2" not in k: continue
        name = "v2 jacobi2p_k" if "jacobi2p_k" in k else "v1 jacobi2_k"
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/lab2/pmc_summary.txt", "w") as o:
    for name in sorted(acc):
        o.write(name + "\n")
        for c in sorted(acc[name]):
            v = acc[name][c]
            o.write("  %-24s mean %16.1f  (n=%d)\n" % (c, sum(v) / len(v), len(v)))
print(open("gpurun_out/lab2/pmc_summary.txt").read())
PY
cat $O/sweep_f32.txt $O/sweep_rb_f32.txt $O/sweep_f64.txt $O/sweep_small.txt
