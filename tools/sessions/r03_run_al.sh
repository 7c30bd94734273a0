#!/bin/bash
# BASELINE config C3 on the final commit (its build after the queue-sort, pinned-stage and helper-thread changes)
O2=gpurun_out/r03final2; mkdir -p $O2
timeout -k 10 500 python bench.py --kernel helmholtz --kappa 10 --no-cpu-baseline > $O2/bench_c3_helmholtz.json 2> $O2/bench_c3_helmholtz.err; echo "c3 rc=$?"
tail -1 $O2/bench_c3_helmholtz.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], d['build_s'], d['build_cold_s'], r['frac'], r['phase_a_achieved'], d['rel_err_sampled_rows']); print(d['build_breakdown'])"
