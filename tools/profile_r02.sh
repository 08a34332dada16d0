#!/bin/bash
# Round-2 evidence for profiles/r02: bench lines, rocprofv3 --kernel-trace --stats of the same commands, HBM counters (separate --pmc passes).
# usage (GPU box, repo root): tools/profile_r02.sh <tag>
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-a}
O=gpurun_out/prof_$TAG
mkdir -p $O
python3 bench.py > $O/bench_jacobi512_f32.json 2> $O/bench_jacobi.err || { tail -5 $O/bench_jacobi.err; exit 1; }
python3 bench.py --solver sor2sma --no-cpu-baseline > $O/bench_rbsor512_f32.json 2> $O/bench_rb.err || exit 1
python3 bench.py --solver pbicgstab --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline > $O/bench_pbicgstab512_f64.json 2> $O/bench_bicg.err || { tail -5 $O/bench_bicg.err; exit 1; }
# the same commands under the profiler (program directly behind "--")
rocprofv3 --kernel-trace --stats -d $O/kt_jacobi --output-format csv -- python3 bench.py --no-cpu-baseline --repeats 2 > $O/kt_jacobi.log 2>&1 || { tail -5 $O/kt_jacobi.log; exit 1; }
rocprofv3 --kernel-trace --stats -d $O/kt_rbsor --output-format csv -- python3 bench.py --solver sor2sma --no-cpu-baseline --repeats 2 > $O/kt_rbsor.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $O/kt_bicg --output-format csv -- python3 bench.py --solver pbicgstab --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline > $O/kt_bicg.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_jac_$c --output-format csv -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --repeats 1 > $O/pmc_jac_$c.log 2>&1 || exit 1
  rocprofv3 --pmc $c -d $O/pmc_rb_$c --output-format csv -- python3 bench.py --solver sor2sma --no-cpu-baseline --steps 8 --warmup 2 --repeats 1 > $O/pmc_rb_$c.log 2>&1 || exit 1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/pmc_jac_SQ --output-format csv -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --repeats 1 > $O/pmc_jac_SQ.log 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum -d $O/pmc_jac_LDS --output-format csv -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --repeats 1 > $O/pmc_jac_LDS.log 2>&1 || exit 1
cp profiles/hbm_traffic.json $O/hbm_traffic.json
python3 tools/summarize_pmc.py jacobi2_512_f32 jacobi2p_k $O/pmc_jac_FETCH_SIZE $O/pmc_jac_WRITE_SIZE $O/hbm_traffic.json > $O/hbm_jac.txt || exit 1
python3 tools/summarize_pmc.py rbsor2_512_f32 jacobi2p_k $O/pmc_rb_FETCH_SIZE $O/pmc_rb_WRITE_SIZE $O/hbm_traffic.json > $O/hbm_rb.txt || exit 1
python3 - "$O" <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(O + "/pmc_jac_SQ/*/*_counter_collection.csv") + glob.glob(O + "/pmc_jac_LDS/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "jacobi2p_k" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O + "/pmc_jacobi2p_512_f32_SQ_LDS.txt", "w") as o:
    for c in sorted(acc):
        o.write("%-24s mean %16.1f  (n=%d)\n" % (c, sum(acc[c]) / len(acc[c]), len(acc[c])))
print(open(O + "/pmc_jacobi2p_512_f32_SQ_LDS.txt").read())
PY
for k in jacobi rbsor bicg; do f=$(ls $O/kt_$k/*/*_kernel_stats.csv | head -1); cp $f $O/kernel_stats_$k.csv; head -8 $f | cut -c1-200; done
cat $O/bench_jacobi512_f32.json $O/bench_rbsor512_f32.json $O/bench_pbicgstab512_f64.json $O/hbm_jac.txt
