cd "$(dirname "$0")/.."
python3 - <<'PY'
import ctypes, os
PY
for rep in 1 2; do
for t in auto 1,1024,2,27 1,1024,2,43 1,1024,2,85 1,1024,2,73 1,1024,2,102 1,512,2,30; do
  if [ $t = auto ]; then unset CZHIP_T2; else export CZHIP_T2=$t; fi
  python3 bench.py --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s %9.0f MLUPS  %.4f ms/launch' % ('$t', d['value'], d['roofline']['kernel_avg_ms']))"
done
done
