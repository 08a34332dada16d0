#!/bin/bash
# A/B of two library builds on one box through the BiCGSTAB leg of bench.py: B = the tree's libczhip_*.so, A = tools/bin/buildA/*.so
cd "$(dirname "$0")/.."
mkdir -p /tmp/B && cp cubez_amd/libczhip_f32.so cubez_amd/libczhip_f64.so /tmp/B/
run() {
  for pc in jacobi sor2sma; do
    python3 bench.py --solver pbicgstab --precond $pc --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pbicgstab+$pc f64 %.3f ms/iteration' % d['ms_per_step'], d.get('ms_per_step_all'))"
  done
}
for rep in 1 2; do
  echo "== build B (tree)"; cp /tmp/B/*.so cubez_amd/; run
  echo "== build A (tools/bin/buildA)"; cp tools/bin/buildA/*.so cubez_amd/; run
done
cp /tmp/B/*.so cubez_amd/
