#!/bin/bash
# evidence on the final sources: the Laplace line + rocprofv3 + PMC passes, the 16-column one-triangle / transposed lines and their kernel summary
bash tools/r03_final_profiles_laplace.sh || exit 1
O2=gpurun_out/r03final2
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" --no-cpu-baseline > $O2/$name.json 2> $O2/$name.err; echo "$name done rc=$?"; }
run bench_sym_rhs16 --symmetric one-triangle --rhs 16
run bench_trans_T_rhs16 --trans T --rhs 16
run bench_helm_sym_rhs16 --kernel helmholtz --kappa 10 --symmetric one-triangle --rhs 16
rm -rf gpurun_out/r03final/kt_sym16
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03final/kt_sym16 -o kt -- python3 bench.py --symmetric one-triangle --rhs 16 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r03final/bench_sym_rhs16_under_rocprof.json 2> gpurun_out/r03final/kt_sym16.err; echo "kt sym16 rc=$?"
echo all done
