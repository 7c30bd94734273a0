#!/bin/bash
# the last call of the round: the whole GPU suite on the final code, then the C5 block once more (LU and Cholesky, no profiling)
O=gpurun_out/final_suite
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log | cut -c1-300
timeout -k 10 120 python tools/hlu_bench.py 500000 100 1e-3 1S 8e-3 > $O/c5block_sym.json 2> $O/c5block_sym.err; cat $O/c5block_sym.json | cut -c1-900
timeout -k 10 120 python tools/hlu_bench.py 500000 100 1e-3 1 8e-3 > $O/c5block_lu.json 2> $O/c5block_lu.err; cat $O/c5block_lu.json | cut -c1-500
echo done
