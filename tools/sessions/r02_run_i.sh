#!/bin/bash
# round-2 GPU session I: build timeline + kernel trace + ACA traffic counters, grouped-phase-A test
set -e
export TMPDIR=/tmp
O=gpurun_out/r02i
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_hmatrix.py -m gpu -x -q -k "grouped_phase_a" > $O/tests.log 2>&1
echo "tests ok"
timeout -k 10 300 python tools/buildprof.py laplace 1000000 3 > $O/buildprof.log 2>&1
echo "buildprof ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_build -o kt -- python3 tools/buildprof.py laplace 1000000 2 > $O/kt_build.log 2>&1
echo "kt ok"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_build -o f -- python3 tools/buildprof.py laplace 1000000 1 > $O/fetch_build.log 2>&1
echo "fetch ok"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_build -o w -- python3 tools/buildprof.py laplace 1000000 1 > $O/write_build.log 2>&1
echo "write ok"
