#!/bin/bash
# evidence on the code with the 16-wide one-triangle / transposed sweeps: the Laplace bench line + rocprofv3 + PMC passes, a fuzz run,
# the matrix-core rate, the 16-column benches
O2=gpurun_out/r03final2; mkdir -p $O2 gpurun_out/r03final
bash tools/r03_final_profiles_laplace.sh || exit 1
export TMPDIR=/tmp
PYTHONPATH=. timeout -k 10 300 python tools/fuzz.py 200 73 > gpurun_out/r03final/fuzz_73.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r03final/fuzz_73.log | cut -c1-300
timeout -k 5 60 tools/microbench/_bin/mfma_f64_rate > gpurun_out/r03final/mfma_f64_rate.txt 2>&1; echo "mfma rate rc=$?"
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" --no-cpu-baseline > $O2/$name.json 2> $O2/$name.err; echo "$name done rc=$?"; }
run bench_rhs16 --rhs 16
run bench_helm_rhs16 --kernel helmholtz --kappa 10 --rhs 16
run bench_sym_rhs16 --symmetric one-triangle --rhs 16
run bench_trans_T_rhs16 --trans T --rhs 16
run bench_helm_sym_rhs16 --kernel helmholtz --kappa 10 --symmetric one-triangle --rhs 16
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03final/kt_sym16 -o kt -- python3 bench.py --symmetric one-triangle --rhs 16 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r03final/bench_sym_rhs16_under_rocprof.json 2> gpurun_out/r03final/kt_sym16.err; echo "kt sym16 rc=$?"
echo all done
