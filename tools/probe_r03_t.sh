#!/bin/bash
# streaming (nontemporal) hint on the output stores: the three single-GPU configs with the hint off / by array size, pair_lab with the run-time flag
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/probe_t; rm -rf $O; mkdir -p $O
for r in 1 2 3; do
  for nt in 0 -1; do
    CZHIP_NT=$nt timeout -k 10 200 python3 bench.py --no-cpu-baseline --repeats 3 > $O/b.json 2>$O/b.err || { tail -3 $O/b.err; exit 1; }
    python3 - $O/b.json $nt <<'PY' | tee -a $O/times.txt
import json, sys
d = json.load(open(sys.argv[1])); c = d["configs"]
k = list(c)
print("CZHIP_NT=%-2s jacobi f32 %.4f ms/step (kernel %.4f ms)  |  rbsor f32 %.4f ms/iteration (kernel %.4f)  |  bicgstab f64 %.3f ms/iteration" % (
    sys.argv[2], d["ms_per_step"], d["roofline"]["kernel_avg_ms"], c[k[0]]["ms_per_step"], c[k[0]]["roofline"]["kernel_avg_ms"], c[k[1]]["ms_per_step"]))
PY
  done
done
for nt in 0 1 0 1; do
  echo "== LAB_NT=$nt" >> $O/pair.txt
  LAB_NT=$nt timeout -k 10 120 tools/bin/pair_lab 512 30 0 0 0 0 1024x27 2>&1 | grep "tj" >> $O/pair.txt || exit 1
  LAB_NT=$nt timeout -k 10 120 tools/bin/pair_lab 512 30 1 0 0 0 1024x27 2>&1 | grep "tj" >> $O/pair.txt || exit 1
done
cat $O/pair.txt | cut -c1-120
