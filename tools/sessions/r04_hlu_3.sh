#!/bin/bash
O=gpurun_out/h5
mkdir -p $O
HTOOL_HLU_PROFILE=1 timeout -k 10 120 python tools/hlu_bench.py 12000 100 1e-3 0 > $O/b12k.json 2> $O/b12k.err; cat $O/b12k.json | cut -c1-600; grep "hlu profile" $O/b12k.err | head -14
echo done
