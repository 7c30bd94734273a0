#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02k
mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_native_build.py tests/test_gpu_symmetric_storage.py -m gpu -x -q > $O/tests.log 2>&1
echo "tests ok"
timeout -k 10 300 python tools/buildprof.py laplace 1000000 4 > $O/buildprof.log 2>&1
echo "buildprof ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_build -o kt -- python3 tools/buildprof.py laplace 1000000 2 > $O/kt_build.log 2>&1
echo "kt ok"
