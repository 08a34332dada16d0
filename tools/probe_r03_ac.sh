#!/bin/bash
# small grids: what a fused pass costs (kernel vs wall) and how the shape / chunk length moves it
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/probe_ac; rm -rf $O; mkdir -p $O
for n in 64 128 192; do
  for t2 in "1" "1,512,2,2" "1,512,2,3" "1,512,2,4" "1,512,2,6" "1,512,2,8" "1,512,2,16" "1,1024,2,2" "1,1024,2,4" "1,1024,2,8"; do
    CZHIP_T2=$t2 timeout -k 10 100 python3 bench.py --cells $n --solver jacobi --prec f64 --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --settle 0 > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; continue; }
    python3 -c "
import json;d=json.load(open('$O/b.json'))
print('%4d^3 f64 jacobi CZHIP_T2=%-12s %.4f ms/step  kernel %.4f ms per pass (%d launches)' % ($n, '$t2', d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['kernel_launches_timed']))" | tee -a $O/times.txt
  done
done
