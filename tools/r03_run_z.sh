#!/bin/bash
# the 16-wide sweeps after the straight-line rewrite: their tests, then the 16-column benches (both storages)
O=gpurun_out/r03z; mkdir -p $O
rc=0

for a in "" "--symmetric one-triangle" "--kernel helmholtz --kappa 10"; do
  n=$(echo "$a" | tr -c 'a-z0-9' '_')
  timeout -k 10 400 python bench.py $a --rhs 16 --no-cpu-baseline > $O/bench_rhs16$n.json 2> $O/bench_rhs16$n.err; echo "bench [$a] rc=$?"
  tail -1 $O/bench_rhs16$n.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], d['rel_err_sampled_rows'], r['launch_us'], r['other_kernels_us'])"
done
