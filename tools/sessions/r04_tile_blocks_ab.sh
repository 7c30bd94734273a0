#!/bin/bash
# HTOOL_TILE_ORDER=blocks against the default (heaviest first): phase B of one column at leaf 100 / leaf 10, and the 16-wide sweep
O=gpurun_out/r04tb
mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --no-warm-build --steps 20 "$@" > $O/$name.json 2> $O/$name.err || echo "FAILED $name"; }
for ord in heavy blocks; do
  export HTOOL_TILE_ORDER=$ord
  run leaf100_$ord
  run leaf10_$ord --leaf 10
  run rhs16_$ord --rhs 16
  run leaf10_rhs16_$ord --leaf 10 --rhs 16
done
python - <<PY
import json
for f in ("leaf100","leaf10","rhs16","leaf10_rhs16"):
    for o in ("heavy","blocks"):
        try:
            d=json.load(open("$O/%s_%s.json"%(f,o))); r=d["roofline"]
            print(f, o, "ms/step %.3f"%d["ms_per_step"], "phase B %.1f us frac %.3f"%(r["launch_us"], r["frac"]), "phase A %.1f"%r["other_kernels_us"]["phase_a_tile_gemv_tall"])
        except Exception as e: print(f, o, "failed", e)
PY
