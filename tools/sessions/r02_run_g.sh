#!/bin/bash
# round-2 GPU session G: new ACA kernels (four waves per leaf), restructured build, restructured 16-wide sweep
set -e
export TMPDIR=/tmp
O=gpurun_out/r02g
mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_native_build.py tests/test_gpu_symmetric_storage.py -m gpu -x -q > $O/tests_build.log 2>&1
echo "build tests ok"
timeout -k 10 300 python tools/buildprof.py laplace 1000000 3 > $O/buildprof.log 2>&1
timeout -k 10 300 python tools/buildprof.py helmholtz 1000000 2 > $O/buildprof_helm.log 2>&1
echo "buildprof ok"
timeout -k 10 500 python -m pytest tests/test_gpu_hmatrix.py tests/test_gpu_boundary.py -m gpu -x -q > $O/tests_hm.log 2>&1
echo "hm tests ok"
for r in 1 16 32; do timeout -k 10 200 python bench.py --rhs $r --no-cpu-baseline > $O/bench_rhs$r.json 2> $O/bench_rhs$r.err; done
echo "rhs ok"
timeout -k 10 300 python bench.py --points 125000 --steps 300 --warmup 20 --no-cpu-baseline --no-phase-timing > $O/bench_125k_graph.json 2> $O/bench_125k_graph.err
HTOOL_PRODUCT_GRAPH=0 timeout -k 10 300 python bench.py --points 125000 --steps 300 --warmup 20 --no-cpu-baseline --no-phase-timing > $O/bench_125k_eager.json 2> $O/bench_125k_eager.err
echo "125k ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_build -o kt -- python3 tools/buildprof.py laplace 1000000 2 > $O/kt_build.log 2>&1
echo "kt ok"
