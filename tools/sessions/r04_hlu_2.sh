#!/bin/bash
O=gpurun_out/h4
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_hlu.py -x -q -m gpu -s -k "not per_gpu_block" > $O/test.log 2>&1; tail -25 $O/test.log | cut -c1-400
timeout -k 10 200 python tools/hlu_bench.py 12000 100 1e-3 0 > $O/b12k.json 2> $O/b12k.err; cat $O/b12k.json; tail -2 $O/b12k.err
timeout -k 10 300 python tools/hlu_bench.py 500000 100 1e-3 1 > $O/c5block.json 2> $O/c5block.err; cat $O/c5block.json; tail -2 $O/c5block.err
echo done
