#!/bin/bash
# the Laplace evidence again on the final sources (a comment in a fingerprinted file changed), and the matrix-core rate with VGPR accumulators
bash tools/r03_final_profiles_laplace.sh || exit 1
timeout -k 5 60 tools/microbench/_bin/mfma_f64_rate > gpurun_out/r03final/mfma_f64_rate.txt 2>&1; echo "mfma rate rc=$?"; cat gpurun_out/r03final/mfma_f64_rate.txt
