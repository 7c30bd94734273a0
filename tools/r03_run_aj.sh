#!/bin/bash
# the queues of a build freed by a helper thread: build tests, then the scope-exit marks again
O=gpurun_out/r03aj; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_native_build.py -x -q -k "not full_size" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/buildprof.py laplace 1000000 4 > $O/buildprof.log 2>&1; echo "rc=$?"
grep -E "scope exit|^rep " $O/buildprof.log | cut -c1-160
grep -E "total [0-9.]+ s" $O/buildprof.log | sed 's/.*coordinates + device set-up/setup/' | cut -c1-80
