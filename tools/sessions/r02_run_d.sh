#!/bin/bash
# round-2 GPU session D: new distributed path tests, build kernel trace, 16-wide sweep counters
set -e
export TMPDIR=/tmp
O=gpurun_out/r02d
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_capi_ctypes.py tests/test_gpu_bench_contract.py tests/test_distributed.py tests/test_gpu_examples.py -m gpu -x -q > $O/tests1.log 2>&1
echo "tests1 ok"
timeout -k 10 600 python -m pytest tests/test_gpu_hmatrix.py -m gpu -x -q > $O/tests2.log 2>&1
echo "tests2 ok"
timeout -k 10 300 python bench.py --force-dist --points 200000 --steps 10 --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err
echo "force-dist ok"
timeout -k 10 300 python bench.py --points 500000 --gmres 50 --no-cpu-baseline > $O/bench_gmres.json 2> $O/bench_gmres.err
echo "gmres ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_build -o kt -- python3 tools/buildprof.py laplace 1000000 2 > $O/kt_build.log 2>&1
echo "kt build ok"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch16 -o f -- python3 bench.py --rhs 16 --steps 3 --warmup 1 --no-cpu-baseline > $O/fetch16.json 2> $O/fetch16.err
echo "fetch16 ok"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write16 -o w -- python3 bench.py --rhs 16 --steps 3 --warmup 1 --no-cpu-baseline > $O/write16.json 2> $O/write16.err
echo "write16 ok"
