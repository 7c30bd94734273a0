#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02aa
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_transposed.py -m gpu -q > $O/tests_t.log 2>&1
echo "transposed rc=$?"; tail -n 6 $O/tests_t.log | cut -c1-300
