#!/bin/bash
# round 3, last session: the GPU suite on the frozen code, then the Laplace evidence (bench line, rocprofv3 stats, PMC passes, build timeline)
export TMPDIR=/tmp
O=gpurun_out/r03v
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1
rc=$?; echo "suite rc=$rc"; tail -n 4 $O/gpu_suite.log
[ $rc -eq 0 ] || exit 1
bash tools/r03_final_profiles_laplace.sh
