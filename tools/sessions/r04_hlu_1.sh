#!/bin/bash
# H-LU on the GPU: tests, then timings of the C5 block (62 500 unknowns) and a 12 000-point operator
O=gpurun_out/h3
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_hlu.py -x -q -m gpu -s > $O/test.log 2>&1; tail -5 $O/test.log | cut -c1-300
timeout -k 10 300 python tools/hlu_bench.py 12000 100 1e-3 0 > $O/b12k.json 2> $O/b12k.err; cat $O/b12k.json; tail -3 $O/b12k.err
timeout -k 10 500 python tools/hlu_bench.py 500000 100 1e-3 1 > $O/c5block.json 2> $O/c5block.err; cat $O/c5block.json; tail -3 $O/c5block.err
echo done
