#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02n
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
echo "tests ok"
for leaf in 10 16 100; do timeout -k 10 300 python bench.py --leaf $leaf --no-cpu-baseline > $O/bench_leaf$leaf.json 2> $O/bench_leaf$leaf.err; done
echo "leaf ok"
