#!/bin/bash
# Round-4 evidence for profiles/r04 (GPU box, repo root): the bench line, rocprofv3 --kernel-trace --stats of the same command, HBM counters
# (separate --pmc passes, one counter per pass: MI355X_MICROARCH.md, HBM section) of the fused FP32 kernels AND of every kernel of the FP64
# BiCGSTAB iteration (configs[3]) on THIS build -> gpurun_out/prof_r04/ (small files only: the traces are deleted).
#   tools/profile_r04.sh [bench|pmc32|pmc64|sq|all]   (default: all)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/prof_r04
what=${1:-all}
mkdir -p $O
[ -f $O/hbm_traffic.json ] || cp profiles/hbm_traffic.json $O/hbm_traffic.json
PMCARGS="--no-cpu-baseline --no-configs --steps 8 --warmup 2 --repeats 1 --settle 0"
if [ $what = bench ] || [ $what = all ]; then
  python3 bench.py > $O/bench_default.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 bench.py --no-cpu-baseline --repeats 2 > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
  cp $(ls $O/kt/*/*_kernel_stats.csv | head -1) $O/bench_default_kernel_stats.csv
  rm -rf $O/kt
  head -8 $O/bench_default_kernel_stats.csv | cut -c1-200
fi
if [ $what = pmc32 ] || [ $what = all ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $O/pmc_jac_$c --output-format csv -- python3 bench.py $PMCARGS > $O/pmc_jac_$c.log 2>&1 || exit 1
    rocprofv3 --pmc $c -d $O/pmc_rb_$c --output-format csv -- python3 bench.py --solver sor2sma $PMCARGS > $O/pmc_rb_$c.log 2>&1 || exit 1
  done
  python3 tools/summarize_pmc.py jacobi2_512_f32 jacobi2p_k $O/pmc_jac_FETCH_SIZE $O/pmc_jac_WRITE_SIZE $O/hbm_traffic.json > $O/hbm_jac.txt || exit 1
  python3 tools/summarize_pmc.py rbsor4_512_f32 rb4_k $O/pmc_rb_FETCH_SIZE $O/pmc_rb_WRITE_SIZE $O/hbm_traffic.json > $O/hbm_rb.txt || exit 1
  rm -rf $O/pmc_jac_* $O/pmc_rb_*
  cat $O/hbm_jac.txt $O/hbm_rb.txt
fi
if [ $what = pmc64 ] || [ $what = all ]; then
  # configs[3]: one warm-up solve of 2 iterations + one timed solve of 6 = 8 iterations in the profiled command; FP64 Jacobi pair beside it
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $O/pmc_bicg_$c --output-format csv -- python3 bench.py --solver pbicgstab --prec f64 --no-cpu-baseline --steps 6 --warmup 2 --repeats 1 --settle 0 > $O/pmc_bicg_$c.log 2>&1 || exit 1
    rocprofv3 --pmc $c -d $O/pmc_jac64_$c --output-format csv -- python3 bench.py --prec f64 $PMCARGS > $O/pmc_jac64_$c.log 2>&1 || exit 1
  done
  python3 tools/summarize_pmc.py '--all=jacobi2p_k;stencil_k<2, 512, 2, 0, 2;ewise_k;triad_dots_k' bicg_512_f64 8 $O/pmc_bicg_FETCH_SIZE $O/pmc_bicg_WRITE_SIZE $O/hbm_traffic.json > $O/hbm_bicg.txt || exit 1
  python3 tools/summarize_pmc.py jacobi2_512_f64 jacobi2p_k $O/pmc_jac64_FETCH_SIZE $O/pmc_jac64_WRITE_SIZE $O/hbm_traffic.json > $O/hbm_jac64.txt || exit 1
  rm -rf $O/pmc_bicg_*/*/*.db $O/pmc_jac64_*  # (the counter csv of the BiCGSTAB passes stays for re-summarising)
  cat $O/hbm_bicg.txt $O/hbm_jac64.txt
fi
if [ $what = sq ] || [ $what = all ]; then
  SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
  rocprofv3 --pmc $SQ -d $O/pmc_rb_SQ --output-format csv -- python3 bench.py --solver sor2sma $PMCARGS > $O/pmc_rb_SQ.log 2>&1 || exit 1
  rocprofv3 --pmc $SQ -d $O/pmc_jac_SQ --output-format csv -- python3 bench.py $PMCARGS > $O/pmc_jac_SQ.log 2>&1 || exit 1
  python3 - "$O" <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
for tag in ("rb", "jac"):
    acc = collections.defaultdict(list)
    for f in glob.glob(O + f"/pmc_{tag}_SQ/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "jacobi2p_k" in r["Kernel_Name"] or "rb4_k" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(O + f"/pmc_jacobi2p_512_f32_{tag}_SQ.txt", "w") as o:
        for c in sorted(acc):
            o.write("%-24s mean %16.1f  (n=%d)\n" % (c, sum(acc[c]) / len(acc[c]), len(acc[c])))
    print(tag, open(O + f"/pmc_jacobi2p_512_f32_{tag}_SQ.txt").read())
PY
  rm -rf $O/pmc_*_SQ
fi
[ -f $O/bench_default.json ] && cat $O/bench_default.json
exit 0
