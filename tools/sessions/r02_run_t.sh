#!/bin/bash
# complex 16-wide matrix-core sweep: tests, then Helmholtz 1 M with 1 / 8 / 16 right-hand sides on one box
export TMPDIR=/tmp
O=gpurun_out/r02t
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_hmatrix.py -m gpu -q -k "sixteen or multi_rhs" > $O/tests.log 2>&1
echo "tests rc=$?"
tail -n 12 $O/tests.log
for r in 1 8 16; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --kernel helmholtz --rhs $r > $O/bench_helm_rhs$r.json 2> $O/bench_helm_rhs$r.err || exit 1
done
echo done
