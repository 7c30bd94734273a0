#!/bin/bash
# one-triangle / transposed sweeps of sixteen on the matrix cores: the new tests, then timings at 1 M points under the profiler
O=gpurun_out/r03x; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_symmetric_storage.py tests/test_gpu_transposed.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o sym16 -- python3 bench.py --symmetric one-triangle --rhs 16 --no-cpu-baseline --steps 10 > $O/bench.json 2> $O/bench.err; echo "rc=$?"
tail -1 $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['rel_err_sampled_rows'])"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    if '16' in r['Name'] : print(r['Name'][:80], r['Calls'], r['AverageNs'], r['Percentage'])
PY
