#!/bin/bash
# round 3, session p: what the driver runs at round end, on the final code: build check, smoke, the GPU suite, the default bench line
export TMPDIR=/tmp
O=gpurun_out/r03u
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?"; tail -n 2 $O/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -n 4 $O/gpu_suite.log
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03u/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("value", round(d["value"],1), "ms", round(d["ms_per_step"],3), "build_s", round(d["build_s"],3), "cold", round(d["build_cold_s"],2), "frac", round(r["frac"],4), "traffic", r["traffic"], "phaseA", round(r["phase_a_achieved"]), "err", d["rel_err_sampled_rows"], "cpu", round(d["cpu_baseline"]["value"],1), d["cpu_baseline"]["cores"])
PY
