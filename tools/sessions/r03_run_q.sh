#!/bin/bash
# round 3, session q: GMRES with the fused Gram-Schmidt kernels of the library: tests, then the per-iteration overhead
export TMPDIR=/tmp
O=gpurun_out/r03q
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_hmatrix.py tests/test_gpu_boundary.py tests/test_distributed.py -m gpu -x -q -k "gmres or jacobi or one_level or distributed or library" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -n 6 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-warm-build > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
run gmres_62k --points 62500 --gmres 50 --force-dist
run gmres_62k_b --points 62500 --gmres 50 --force-dist
run gmres_62k_single --points 62500 --gmres 50
run gmres_500k --points 500000 --gmres 50 --force-dist
python - <<'PY'
import json
for n in ("gmres_62k","gmres_62k_b","gmres_62k_single","gmres_500k"):
    d=json.loads(open(f"gpurun_out/r03q/{n}.json").read().strip().splitlines()[-1])
    r=d["roofline"]; ok=r.get("other_kernels_us")
    prod=(r["launch_us"]+sum(ok.values())) if ok else None
    print(n, "ms/it", round(d["ms_per_step"],4), "product kernels us", round(prod,1), "over", round(d["ms_per_step"]*1e3-prod,1), d["gmres"]["iterations_to_1e-6"], d["gmres"]["true_relative_residual"])
PY
