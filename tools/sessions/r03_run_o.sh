#!/bin/bash
# round 3, session o: tests of the last changes (scratch stride of the lockstep kernels, shifted LU always on the device), replay of the one fuzz failure next to the CPU oracle
export TMPDIR=/tmp
O=gpurun_out/r03o
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_native_build.py tests/test_gpu_hmatrix.py -m gpu -x -q -k "lockstep or dense_factor or one_level or jacobi or confirmation or device_aca or C3" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -n 4 $O/tests.log
HTOOL_ACA_STEP_MIN=200 PYTHONPATH=. timeout -k 10 300 python tools/fuzz.py 100000 72 193 > $O/fuzz_72_193_step200.log 2>&1; echo "replay (lockstep > 200) rc=$?"; tail -n 3 $O/fuzz_72_193_step200.log | cut -c1-200
PYTHONPATH=. timeout -k 10 300 python tools/fuzz.py 100000 72 193 > $O/fuzz_72_193_default.log 2>&1; echo "replay (default) rc=$?"; tail -n 3 $O/fuzz_72_193_default.log | cut -c1-200
FUZZ_CONFIRM=1 PYTHONPATH=. timeout -k 10 300 python tools/fuzz.py 100000 72 193 > $O/fuzz_72_193_confirm1.log 2>&1; echo "replay (confirm 1) rc=$?"; tail -n 3 $O/fuzz_72_193_confirm1.log | cut -c1-200
