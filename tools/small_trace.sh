#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace --stats) of the fused pass on small grids, preloaded form on / off: how much of a pass is kernel,
# how much the gap between two dependent launches.   tools/small_trace.sh  -> gpurun_out/small_trace/
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/small_trace
rm -rf $O; mkdir -p $O
for n in 64 128; do
  for pre in 1 0; do
    rocprofv3 --kernel-trace --stats -d $O/kt_${n}_$pre --output-format csv -- python3 tools/rate.py "f32:$n,$n,$n:jacobi@notime=1;CZHIP_T2_PRE=$pre" > $O/run_${n}_$pre.log 2>&1 || { tail -5 $O/run_${n}_$pre.log; exit 1; }
    f=$(ls $O/kt_${n}_$pre/*/*_kernel_stats.csv | head -1)
    echo "== $n^3 f32 jacobi, preload $pre: $(grep MLUPS $O/run_${n}_$pre.log)"
    python3 -c "
import csv,sys
for r in list(csv.DictReader(open('$f')))[:3]:
    n=r['Name'].replace('void (anonymous namespace)::','').split('(')[0]
    print('   %-44s calls %6s  avg %9.0f ns  min %9s  max %9s' % (n[:44], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
"
    rm -rf $O/kt_${n}_$pre
  done
done 2>&1 | tee $O/summary.txt
