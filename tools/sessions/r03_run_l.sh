#!/bin/bash
# round 3, session l: the distributed transposed product (2 / 3 gloo ranks, one-rank RCCL through the C ABI), then the final benches (part 2)
export TMPDIR=/tmp
O=gpurun_out/r03l
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_distributed.py tests/test_gpu_capi_ctypes.py -m gpu -x -q > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -n 8 $O/tests.log
[ $rc -eq 0 ] || exit 1
bash tools/r03_final_benches.sh
