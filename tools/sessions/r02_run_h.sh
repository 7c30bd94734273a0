#!/bin/bash
# round-2 GPU session H: grouped phase A -- full GPU suite, then benches at leaf 100 / 16 / 10, 16 right-hand sides, build times
set -e
export TMPDIR=/tmp
O=gpurun_out/r02h
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
echo "tests ok"
for leaf in 100 16 10; do timeout -k 10 300 python bench.py --leaf $leaf --no-cpu-baseline > $O/bench_leaf$leaf.json 2> $O/bench_leaf$leaf.err; done
echo "leaf ok"
HTOOL_PHASE_A_GROUP=1 timeout -k 10 300 python bench.py --leaf 100 --no-cpu-baseline > $O/bench_leaf100_nogroup.json 2> $O/bench_leaf100_nogroup.err
HTOOL_PHASE_A_GROUP=1 timeout -k 10 300 python bench.py --leaf 16 --no-cpu-baseline > $O/bench_leaf16_nogroup.json 2> $O/bench_leaf16_nogroup.err
echo "nogroup ok"
timeout -k 10 200 python bench.py --rhs 16 --no-cpu-baseline > $O/bench_rhs16.json 2> $O/bench_rhs16.err
timeout -k 10 300 python tools/buildprof.py laplace 1000000 3 > $O/buildprof.log 2>&1
echo "done"
