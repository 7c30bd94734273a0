#!/bin/bash
# explicit inverse factors of the small diagonal blocks + private slots / REDUCE tasks in the sweeps: tests, then the application of the C5 block's factors
O=gpurun_out/h19
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_hlu.py -x -q -m gpu -k "not per_gpu_block" > $O/test.log 2>&1; tail -8 $O/test.log | cut -c1-400
HLU_BENCH_REPS=1 timeout -k 10 120 python tools/hlu_bench.py 500000 100 1e-3 1S 8e-3 > $O/c5block_sym.json 2> $O/c5block_sym.err; cat $O/c5block_sym.json | cut -c1-1300
HLU_BENCH_REPS=1 timeout -k 10 120 python tools/hlu_bench.py 12000 100 1e-3 0 > $O/b12k.json 2> $O/b12k.err; cat $O/b12k.json | cut -c900-1400
echo done
