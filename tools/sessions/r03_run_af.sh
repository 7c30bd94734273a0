#!/bin/bash
# dense(H) leaf by leaf: the GPU suite (to_dense is used all over it), then the dense device LU at 62 500 unknowns
O=gpurun_out/r03af; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/dense_lu_62k.py 500000 > $O/dense_lu_62k.log 2>&1; echo "dense lu rc=$?"; tail -8 $O/dense_lu_62k.log
