#!/bin/bash
# the two-stage pass over grid sizes: the launcher's own choice of shape / chunk length against fixed ones
cd "$(dirname "$0")/.."
for n in 128 192 256 320 384 448 512 640 768 1024; do
  for t in auto 1,512,2,16 1,1024,2,16 1,512,2,32 1,1024,2,32; do
    if [ $t = auto ]; then unset CZHIP_T2; else export CZHIP_T2=$t; fi
    python3 bench.py --n $n --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('n %4d  %-12s %9.0f MLUPS  %.4f ms/launch' % ($n, '$t', d['value'], d['roofline']['kernel_avg_ms']))"
  done
done
