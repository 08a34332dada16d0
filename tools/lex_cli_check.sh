#!/bin/bash
# compare the residual histories of the one-launch lexicographic line SOR and the launch-per-diagonal path through the CLI
# (FP64 lines beyond ~640 unknowns with the 4x4 final stage take the literal per-line kernel in both: their coefficient table does not fit LDS)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/cli && cd gpurun_out/cli
rc=0
for cfg in "f32 300 200 400 pcr 12 1.2" "f64 300 200 400 pcr_esa 8 1.2" "f32 40 36 1024 pcr 6 1.2" "f64 40 36 1024 pcr 6 1.2" "f64 36 40 900 pcr_rb_esa 6 1.2" "f32 130 70 260 pcr_eda 10 1.2" "f32 130 70 260 pcr_maf 10 1.2" "f64 64 300 100 pcr_eda_maf 10 1.2" "f32 513 33 65 pcr 8 1.2"; do
  set -- $cfg
  prec=$1; shift
  for pipe in 1 0; do
    rm -f *.txt
    CZHIP_PCR_PIPE=$pipe timeout -k 10 120 ../../cubez_amd/cz_$prec "$@" > out_$pipe.log 2>&1 || { echo "FAILED: $cfg pipe=$pipe"; tail -3 out_$pipe.log; rc=1; }
    cp "$4.txt" hist_$pipe.dat 2>/dev/null   # the residual history the reference CLI writes (the profile file carries wall times)
  done
  if cmp -s hist_1.dat hist_0.dat && [ -s hist_1.dat ]; then echo "same history ($(wc -l < hist_1.dat) lines): $cfg"; else echo "DIFFERENT: $cfg"; diff hist_1.dat hist_0.dat | head -5; rc=1; fi
done
exit $rc
