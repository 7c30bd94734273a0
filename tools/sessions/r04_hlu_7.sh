#!/bin/bash
# (1) where the time of one application goes (12 000 unknowns); (2) the whole symmetric operator (one triangle stored) of the bench's
# shifted system at 250 000 unknowns: beyond what a dense copy can hold
O=gpurun_out/h11
mkdir -p $O
HTOOL_HLU_PROFILE=2 timeout -k 10 120 python tools/hlu_bench.py 12000 100 1e-3 0 > $O/b12k.json 2> $O/b12k.err; grep "solve profile" $O/b12k.err | tail -4
HLU_BENCH_REPS=1 HTOOL_HLU_REFINE=2 timeout -k 10 900 python tools/hlu_bench.py 250000 100 1e-3 S 8e-3 > $O/s250k.json 2> $O/s250k.err; cat $O/s250k.json | cut -c1-1500; grep -v "hlu_bench\|amdgpu" $O/s250k.err | tail -3 | cut -c1-300
echo done
