#!/bin/bash
# LU requests for symmetric operators try Cholesky first: the dense-factor tests, the 62 500-unknown block both ways, then the round-end rehearsal
O=gpurun_out/r03ag; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_hmatrix.py -x -q -k "factor or cholesky or lu or jacobi or one_level or dense" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/dense_lu_62k.py 500000 S > $O/dense_lu_62k_sym.log 2>&1; echo "dense lu sym rc=$?"; tail -4 $O/dense_lu_62k_sym.log
timeout -k 10 300 python tools/dense_lu_62k.py 500000 > $O/dense_lu_62k.log 2>&1; echo "dense lu rc=$?"; tail -3 $O/dense_lu_62k.log
bash tools/r03_run_w.sh
