#!/bin/bash
# the 16-column one-triangle / transposed bench lines on the final sources, then the round-end rehearsal (smoke + default line)
O2=gpurun_out/r03final2; mkdir -p $O2
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" --no-cpu-baseline > $O2/$name.json 2> $O2/$name.err; echo "$name done rc=$?"; }
run bench_sym_rhs16 --symmetric one-triangle --rhs 16
run bench_trans_T_rhs16 --trans T --rhs 16
run bench_helm_sym_rhs16 --kernel helmholtz --kappa 10 --symmetric one-triangle --rhs 16
bash tools/r03_run_w.sh
