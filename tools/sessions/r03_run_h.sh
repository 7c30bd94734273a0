#!/bin/bash
# round 3, session h: dense device factorisation (tests incl. the 62 500-unknown block), changed solver tests, full-size gaps, build timing, benches
export TMPDIR=/tmp
O=gpurun_out/r03h
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_boundary.py tests/test_distributed.py -m gpu -x -q > $O/tests1.log 2>&1
rc=$?; echo "tests1 rc=$rc"; tail -n 6 $O/tests1.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_hmatrix.py tests/test_gpu_native_build.py -m gpu -x -q -k "jacobi or gmres or full_size" > $O/tests2.log 2>&1
rc=$?; echo "tests2 rc=$rc"; tail -n 6 $O/tests2.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/buildprof.py laplace 1000000 3 > $O/bp_laplace.log 2>&1; grep -E "native build timing" $O/bp_laplace.log | sed -e 's/.*block tree/block tree/' | tail -n 2; grep -E "pack batch 1" $O/bp_laplace.log | tail -n 1
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
run bench
run bench_c3 --kernel helmholtz --no-cpu-baseline
python - <<'PY'
import json
for n in ("bench","bench_c3"):
    d=json.loads(open(f"gpurun_out/r03h/{n}.json").read().strip().splitlines()[-1])
    print(n, "value", round(d["value"],1), "ms", round(d["ms_per_step"],3), "build_s", round(d["build_s"],3), "cold", round(d["build_cold_s"],3), "setup", round(d["setup_s"],3), "frac", d["roofline"]["frac"], "phaseA", d["roofline"]["phase_a_achieved"], "err", d["rel_err_sampled_rows"])
    print("   ", d["build_breakdown"])
PY
