#!/bin/bash
# Evidence for the one-launch lexicographic line SOR (pcr_lex_wg_k), 512^3: bench lines with the kernel on / off (launch per diagonal) on the
# same box, the strip profile (CZHIP_PCR_PIPE_PROF) of the shapes that were compared, and the kernel trace.
# usage (GPU box, repo root): tools/pcr_lex_run.sh      (the strip profile needs a library built with -DCZ_LEX_PROF: make CXXFLAGS_EXTRA=-DCZ_LEX_PROF)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/pcr_lex
mkdir -p $O
{
  for prec in f32 f64; do
    for s in pcr pcr_esa; do
      for pipe in 1 0; do
        echo "# CZHIP_PCR_PIPE=$pipe --solver $s --prec $prec"
        CZHIP_PCR_PIPE=$pipe timeout -k 10 300 python3 bench.py --solver $s --prec $prec --steps 6 --warmup 2 --repeats 3 --no-cpu-baseline 2>/dev/null || exit 1
      done
    done
  done
  for s in pcr_maf pcr_eda_maf; do
    for pipe in 1 0; do
      echo "# CZHIP_PCR_PIPE=$pipe --solver $s --prec f32"
      CZHIP_PCR_PIPE=$pipe timeout -k 10 300 python3 bench.py --solver $s --steps 4 --warmup 1 --repeats 3 --no-cpu-baseline 2>/dev/null || exit 1
    done
  done
} > $O/bench_pcr_lex_512_on_off.jsonl || { tail -3 $O/bench_pcr_lex_512_on_off.jsonl; exit 1; }
{
  for cfg in 0,1 2,1 0,2; do
    echo "# CZHIP_PCR_PIPE=1,2,$cfg (groups per workgroup [0 = launcher's choice], rows per thread)"
    CZHIP_PCR_PIPE=1,2,$cfg timeout -k 10 300 python3 bench.py --solver pcr --steps 6 --warmup 2 --repeats 3 --no-cpu-baseline 2>/dev/null | cut -c1-260 || exit 1
    CZHIP_PCR_PIPE=1,2,$cfg CZHIP_PCR_PIPE_PROF=64 timeout -k 10 300 python3 bench.py --solver pcr --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline 2>&1 >/dev/null | grep -A8 "pcr_lex_wg_k" | head -9
  done
} > $O/pcr_lex_strip_profile_512_f32.txt || exit 1
rocprofv3 --kernel-trace --stats -d $O/kt_pcr --output-format csv -- python3 bench.py --solver pcr --steps 6 --warmup 2 --repeats 2 --no-cpu-baseline > $O/kt_pcr.log 2>&1 || { tail -5 $O/kt_pcr.log; exit 1; }
f=$(ls $O/kt_pcr/*/*_kernel_stats.csv | head -1); cp $f $O/kernel_stats_pcr.csv
head -6 $O/kernel_stats_pcr.csv | cut -c1-220
cat $O/bench_pcr_lex_512_on_off.jsonl | cut -c1-150
cat $O/pcr_lex_strip_profile_512_f32.txt | cut -c1-230
