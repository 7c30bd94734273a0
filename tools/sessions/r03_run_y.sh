#!/bin/bash
# per-kernel times of the one-triangle 16-wide sweep at 1 M points
O=gpurun_out/r03y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o sym16 -- python3 bench.py --symmetric one-triangle --rhs 16 --no-cpu-baseline --steps 10 > $O/bench.json 2> $O/bench.err; echo "rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]: print(r['Name'][:80], r['Calls'], r['AverageNs'], r['Percentage'])
PY
