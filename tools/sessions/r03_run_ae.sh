#!/bin/bash
# round-end rehearsal on the final code: smoke(), the GPU suite (with the slowest tests listed), the default bench line
O=gpurun_out/r03ae; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=25 > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; grep -E "passed|failed" $O/tests.log | tail -2
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -1 $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], d['build_s'], d['build_cold_s'], r['frac'], r['traffic'])"
