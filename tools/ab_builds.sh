#!/bin/bash
# A/B of two library builds on one box: B = the tree's libczhip_*.so, A = tools/bin/buildA/*.so
cd "$(dirname "$0")/.."
O=gpurun_out/ab
mkdir -p $O /tmp/B
cp cubez_amd/libczhip_f32.so cubez_amd/libczhip_f64.so /tmp/B/
run() {
  for s in "jacobi f32" "sor2sma f32" "jacobi f64" "pcr_rb f32" "psor f32" "jacobi_maf f32" "pcr_rb_maf f32"; do
    set -- $s
    python3 bench.py --solver $1 --prec $2 --steps 40 --warmup 6 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s %s %10.0f MLUPS  kernel %.4f ms' % ('$1','$2',d['value'],d['roofline']['kernel_avg_ms']))"
  done
  python3 bench.py --solver pbicgstab --steps 10 --warmup 2 --repeats 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pbicgstab f64 %.3f ms/iteration' % d['ms_per_step'])"
}
for rep in 1 2; do
  echo "== build B (tree)"; cp /tmp/B/*.so cubez_amd/; run
  echo "== build A (tools/bin/buildA)"; cp tools/bin/buildA/*.so cubez_amd/; run
done 2>&1 | tee $O/ab.txt
cp /tmp/B/*.so cubez_amd/
