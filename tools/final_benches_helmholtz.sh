#!/bin/bash
# final evidence of the round, part 3: BASELINE config C3 (1 M-point Helmholtz, complex128) with 1 / 8 / 16 right-hand sides and conjugate-transposed
set -e
export TMPDIR=/tmp
O=gpurun_out/final2
mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" --no-cpu-baseline > $O/$name.json 2> $O/$name.err; echo "$name done"; }
run bench_c3_helmholtz --kernel helmholtz --kappa 10
run bench_helm_rhs8 --kernel helmholtz --kappa 10 --rhs 8
run bench_helm_rhs16 --kernel helmholtz --kappa 10 --rhs 16
run bench_helm_trans_C --kernel helmholtz --kappa 10 --trans C
echo all done
