#!/bin/bash
# the whole symmetric operator (one triangle stored) of BASELINE C5 -- 500 000 unknowns -- on the bench's shifted system: "factorisable at all"
O=gpurun_out/h13
mkdir -p $O
HLU_BENCH_REPS=1 HTOOL_HLU_REFINE=2 timeout -k 10 1050 python tools/hlu_bench.py 500000 100 1e-3 S 8e-3 > $O/s500k.json 2> $O/s500k.err; cat $O/s500k.json | cut -c1-1500; grep -v "hlu_bench\|amdgpu" $O/s500k.err | tail -3 | cut -c1-300
echo done
