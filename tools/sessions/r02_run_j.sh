#!/bin/bash
# round-2 GPU session J: block ACA kernel with fused dot products, build host-side trims; parity tests, timeline, counters
set -e
export TMPDIR=/tmp
O=gpurun_out/r02j
mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_native_build.py tests/test_gpu_symmetric_storage.py -m gpu -x -q > $O/tests.log 2>&1
echo "tests ok"
timeout -k 10 300 python tools/buildprof.py laplace 1000000 3 > $O/buildprof.log 2>&1
timeout -k 10 300 python tools/buildprof.py helmholtz 1000000 2 > $O/buildprof_helm.log 2>&1
echo "buildprof ok"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_build -o f -- python3 tools/buildprof.py laplace 1000000 1 > $O/fetch_build.log 2>&1
echo "fetch ok"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/sq_build -o s -- python3 tools/buildprof.py laplace 1000000 1 > $O/sq_build.log 2>&1
echo "sq ok"
timeout -k 10 300 python bench.py --kernel helmholtz --rhs 1 --no-cpu-baseline > $O/bench_helm_rhs1.json 2> $O/bench_helm_rhs1.err
timeout -k 10 300 python bench.py --kernel helmholtz --rhs 8 --no-cpu-baseline > $O/bench_helm_rhs8.json 2> $O/bench_helm_rhs8.err
echo "helm ok"
