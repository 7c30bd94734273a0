#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02r
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_transposed.py tests/test_gpu_capi_ctypes.py -m gpu -q > $O/tests_t.log 2>&1
echo "transposed rc=$?"
tail -n 15 $O/tests_t.log
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_transposed.py > $O/tests.log 2>&1
echo "suite rc=$?"
tail -n 5 $O/tests.log
