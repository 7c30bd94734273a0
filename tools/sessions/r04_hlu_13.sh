#!/bin/bash
# A/B of the chain fusion of the solve programs (rows of a subtree swept by one workgroup) on the C5 block, hierarchical Cholesky
# (the code of this experiment was taken out again: see profiles/r04_hlu_chain_fusion_ab.txt and NOTES.md; the script is kept as the record of the call)
O=gpurun_out/h17
mkdir -p $O
for fr in 0 256 512 1024; do
HTOOL_HLU_FUSE_ROWS=$fr HLU_BENCH_REPS=1 timeout -k 10 100 python tools/hlu_bench.py 500000 100 1e-3 1S 8e-3 > $O/fuse_$fr.json 2> $O/fuse_$fr.err
python - <<PY
import json
d=json.load(open("$O/fuse_$fr.json")); i=d["info"]
print("fuse_rows $fr: one application %.1f ms (1 column), %.1f ms (8 columns); solve tasks %d, launches %d; lu_solve on host vectors %.3f s" % (d["apply_ms_mu1"], d["apply_ms_mu8"], i["solve_tasks"], i["solve_launches"], d["lu_solve_host_s"]))
PY
done
echo done
