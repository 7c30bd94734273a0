#!/bin/bash
# 16-wide sweeps after the slab stride / finishing-pass changes, and the XCD-aware tile order for them
O=gpurun_out/r04w16
mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --no-warm-build --steps 10 "$@" > $O/$name.json 2> $O/$name.err || echo "FAILED $name"; }
run sym16 --symmetric one-triangle --rhs 16
run wide16 --rhs 16
run trans16 --trans T --rhs 16
HTOOL_TILE_ORDER=xcd run wide16_xcd --rhs 16
HTOOL_TILE_ORDER=xcd run sym16_xcd --symmetric one-triangle --rhs 16
python - <<PY
import json
for f in ("sym16","wide16","trans16","wide16_xcd","sym16_xcd"):
    try:
        d=json.load(open("$O/%s.json"%f)); r=d["roofline"]
        print(f, "ms/step %.2f"%d["ms_per_step"], "frac %.3f"%r["frac"], "launch_us", r["launch_us"], r.get("other_kernels_us"), "err %.2e"%d["rel_err_sampled_rows"])
    except Exception as e: print(f, "failed", e)
PY
