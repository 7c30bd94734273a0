#!/bin/bash
# round 3, session s: build temporaries from one scratch slab (no hipMalloc / hipFree pairs inside a build): timing, build tests, leak check
export TMPDIR=/tmp
O=gpurun_out/r03s
mkdir -p $O
timeout -k 10 300 python tools/buildprof.py laplace 1000000 4 > $O/bp_laplace.log 2>&1 || { echo "laplace failed"; tail -n 5 $O/bp_laplace.log; exit 1; }
grep -E "^rep|native build timing" $O/bp_laplace.log | sed -e 's/.*block tree/block tree/' | tail -n 6
timeout -k 10 900 python -m pytest tests/test_gpu_native_build.py tests/test_gpu_symmetric_storage.py -m gpu -x -q -k "not full_size" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -n 4 $O/tests.log
PYTHONPATH=. timeout -k 10 200 python tools/leak_check.py > $O/leak.log 2>&1; echo "leak rc=$?"; tail -n 2 $O/leak.log | cut -c1-200
timeout -k 10 500 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03s/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("value", round(d["value"],1), "ms", round(d["ms_per_step"],3), "build_s", round(d["build_s"],3), "cold", round(d["build_cold_s"],2), "frac", round(r["frac"],4)); print(d["build_breakdown"])
PY
