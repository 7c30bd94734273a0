#!/bin/bash
# leaf records of a pack uploaded from a pinned stage: build tests, then the pack timings of four builds
O=gpurun_out/r03aj; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_native_build.py tests/test_gpu_symmetric_storage.py -x -q -k "not full_size" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/buildprof.py laplace 1000000 4 > $O/buildprof.log 2>&1; echo "rc=$?"
grep -E "pack batch|^rep " $O/buildprof.log | cut -c1-200
grep -E "total [0-9.]+ s" $O/buildprof.log | sed 's/.*ACA kernels/ACA/' | cut -c1-200
