#!/bin/bash
O=gpurun_out/h7
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_hlu.py -x -q -m gpu -s -k "not per_gpu_block" > $O/test.log 2>&1; tail -12 $O/test.log | cut -c1-300
HTOOL_HLU_PROFILE=1 timeout -k 10 120 python tools/hlu_bench.py 12000 100 1e-3 0 > $O/b12k.json 2> $O/b12k.err; cat $O/b12k.json | cut -c1-1500; grep "hlu profile" $O/b12k.err | head -8
HTOOL_HLU_PROFILE=1 timeout -k 10 200 python tools/hlu_bench.py 500000 100 1e-3 1 > $O/c5block.json 2> $O/c5block.err; cat $O/c5block.json | cut -c1-1500; grep "hlu profile" $O/c5block.err | head -8
echo done
