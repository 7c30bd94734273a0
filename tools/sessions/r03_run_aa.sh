#!/bin/bash
# full GPU suite on the code with the 16-wide one-triangle / transposed sweeps, a short fuzz, the C3 build after the sort change
O=gpurun_out/r03aa; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/fuzz.py 60 71 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz.log
timeout -k 10 400 python bench.py --kernel helmholtz --kappa 10 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
tail -1 $O/bench_c3.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['build_s'], d['build_breakdown'])"
