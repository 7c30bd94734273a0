#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03t
mkdir -p $O
timeout -k 10 500 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03t/bench.json").read().strip().splitlines()[-1])
print("build_s", d["build_s"]); print(d["build_breakdown"])
PY
