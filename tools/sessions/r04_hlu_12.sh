#!/bin/bash
# A/B of the splitting thresholds and the window size on the C5 block (hierarchical Cholesky)
O=gpurun_out/h16
mkdir -p $O
run() { name=$1; shift; env "$@" HLU_BENCH_REPS=2 timeout -k 10 100 python tools/hlu_bench.py 500000 100 1e-3 1S 8e-3 > $O/$name.json 2> $O/$name.err; python - <<PY
import json
d=json.load(open("$O/$name.json")); i=d["info"]
print("$name", d["lu_factorization_s_1"], "tasks", i["tasks"], "launches", i["launches"], "windows", i["windows"], "truncations", i["truncations"], "peak GB", round(i["peak_bytes"]/1e9,1), "err", d["solve_error_unrefined"])
PY
}
run default A=1
run split_4_2_32 HTOOL_HLU_SPLIT=4,2,32
run split_6_3_48 HTOOL_HLU_SPLIT=6,3,48
run split_16_8_16 HTOOL_HLU_SPLIT=16,8,16
run window_16g HTOOL_HLU_WINDOW_MB=16000
run window_1g HTOOL_HLU_WINDOW_MB=1000
echo done
