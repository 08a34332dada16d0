#!/bin/bash
# compare the residual histories of the one-launch lexicographic line SOR and the launch-per-diagonal path through the CLI
# (long lines: the coefficient table in global memory, the literal kernel, global scratch beyond LDS -- chosen by the launcher in the default mode,
# round 3; the run of round 2 that failed, `f64 40 36 1024 pcr`, is among the cases)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/cli && cd gpurun_out/cli
rc=0
for cfg in "f32 300 200 400 pcr 12 1.2" "f64 300 200 400 pcr_esa 8 1.2" "f32 40 36 1024 pcr 6 1.2" "f64 40 36 1024 pcr 6 1.2" "f64 36 40 900 pcr_rb_esa 6 1.2" "f32 130 70 260 pcr_eda 10 1.2" "f32 130 70 260 pcr_maf 10 1.2" "f64 64 300 100 pcr_eda_maf 10 1.2" "f32 513 33 65 pcr 8 1.2" \
           "f64 40 36 1024 pcr_j_esa 6 0.9" "f64 40 36 1024 pcr_esa 6 1.2" "f64 24 20 2048 pcr_rb 4 1.2" "f32 20 16 6000 pcr 3 1.2" "f64 20 16 4000 pcr_j_esa 3 0.9" "f64 20 16 6000 pcr_rb_esa 3 1.2"; do
  set -- $cfg
  prec=$1; shift
  for pipe in 1 0; do
    rm -f *.txt
    CZHIP_PCR_PIPE=$pipe timeout -k 10 120 ../../cubez_amd/cz_$prec "$@" > out_$pipe.log 2>&1 || { echo "FAILED: $cfg pipe=$pipe"; tail -3 out_$pipe.log; rc=1; }
    cp "$4.txt" hist_$pipe.dat 2>/dev/null   # the residual history the reference CLI writes (the profile file carries wall times)
  done
  if cmp -s hist_1.dat hist_0.dat && [ -s hist_1.dat ]; then echo "same history ($(wc -l < hist_1.dat) lines): $cfg"; else echo "DIFFERENT: $cfg"; diff hist_1.dat hist_0.dat | head -5; rc=1; fi
done
# the lexicographic point SOR: one launch per sweep (psor_col_k) against a launch per hyperplane of tiles
for cfg in "f32 200 150 300 psor 10 1.2" "f64 130 70 260 psor 10 1.2" "f32 130 70 260 psor_maf 10 1.2" "f32 17 300 40 psor 8 1.2"; do
  set -- $cfg
  prec=$1; shift
  for one in 1 0; do
    rm -f *.txt
    CZHIP_PSOR=$one timeout -k 10 120 ../../cubez_amd/cz_$prec "$@" > out_$one.log 2>&1 || { echo "FAILED: $cfg one_launch=$one"; tail -3 out_$one.log; rc=1; }
    cp "$4.txt" hist_$one.dat 2>/dev/null
  done
  if cmp -s hist_1.dat hist_0.dat && [ -s hist_1.dat ]; then echo "same history ($(wc -l < hist_1.dat) lines): $cfg"; else echo "DIFFERENT: $cfg"; diff hist_1.dat hist_0.dat | head -5; rc=1; fi
done
exit $rc
