#!/bin/bash
# the finishing pass of the 16-wide fused sweeps as a gather: the tests of the one-triangle / transposed sweeps, then the two benches
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_symmetric_storage.py tests/test_gpu_transposed.py tests/test_gpu_hmatrix.py -x -q -k "sixteen or one_triangle or transposed or stress" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
for a in "--symmetric one-triangle" "--trans T"; do
  n=$(echo "$a" | tr -c 'a-z0-9' '_')
  timeout -k 10 400 python bench.py $a --rhs 16 --no-cpu-baseline > $O/bench_rhs16$n.json 2> $O/bench_rhs16$n.err; echo "bench [$a] rc=$?"
  tail -1 $O/bench_rhs16$n.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], d['rel_err_sampled_rows'], r['launch_us'], r['other_kernels_us'])"
done
