#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02o
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc"
tail -n 3 $O/tests.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
for leaf in 10 16 100; do timeout -k 10 300 python bench.py --leaf $leaf --no-cpu-baseline > $O/bench_leaf$leaf.json 2> $O/bench_leaf$leaf.err || exit 1; done
echo "leaf ok"
