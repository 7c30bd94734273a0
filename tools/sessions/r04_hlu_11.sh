#!/bin/bash
# the whole symmetric operators again, now as hierarchical Cholesky factorisations with compacted factors: 250 000 and 500 000 unknowns (BASELINE C5), shifted system
O=gpurun_out/h15
mkdir -p $O
HLU_BENCH_REPS=1 HTOOL_HLU_REFINE=2 timeout -k 10 500 python tools/hlu_bench.py 250000 100 1e-3 S 8e-3 > $O/s250k.json 2> $O/s250k.err; cat $O/s250k.json | cut -c1-1300
HLU_BENCH_REPS=1 HTOOL_HLU_REFINE=2 timeout -k 10 700 python tools/hlu_bench.py 500000 100 1e-3 S 8e-3 > $O/s500k.json 2> $O/s500k.err; cat $O/s500k.json | cut -c1-1300; grep -v "hlu_bench\|amdgpu" $O/s500k.err | tail -3 | cut -c1-300
echo done
