#!/bin/bash
# pilot round overlapped with the dense helper; the queue dealt out over the expected rounds (Helmholtz): timelines, then the build / boundary / transposed tests
export TMPDIR=/tmp
O=gpurun_out/r02z
mkdir -p $O
timeout -k 10 200 python tools/buildprof.py laplace 1000000 4 2> $O/bp.log || exit 1
grep -E "native build timing|timeline" $O/bp.log | tail -n 4
timeout -k 10 300 python tools/buildprof.py helmholtz 1000000 3 2> $O/bph.log || exit 1
grep -E "native build timing|ACA round" $O/bph.log | tail -n 8
timeout -k 10 900 python -m pytest tests/test_gpu_native_build.py tests/test_gpu_boundary.py tests/test_gpu_transposed.py tests/test_gpu_hmatrix.py -m gpu -q > $O/tests.log 2>&1
echo "tests rc=$?"
tail -n 5 $O/tests.log
PYTHONPATH=. timeout -k 10 300 python tools/fuzz.py 200 31 > $O/fuzz31.log 2>&1
echo "fuzz rc=$?"; tail -n 1 $O/fuzz31.log; grep "FAIL" $O/fuzz31.log | cut -c1-500 | head -n 5
