#!/bin/bash
# the hierarchical Cholesky factorisation (symmetric operators): tests, then the C5 block declared symmetric and a 12 000-point symmetric operator
O=gpurun_out/h14
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_hlu.py -x -q -m gpu > $O/test.log 2>&1; tail -12 $O/test.log | cut -c1-400
timeout -k 10 120 python tools/hlu_bench.py 12000 100 1e-3 S > $O/b12k_sym.json 2> $O/b12k_sym.err; cat $O/b12k_sym.json | cut -c1-1300
HTOOL_HLU_PROFILE=1 timeout -k 10 200 python tools/hlu_bench.py 500000 100 1e-3 1S 8e-3 > $O/c5block_sym.json 2> $O/c5block_sym.err; cat $O/c5block_sym.json | cut -c1-1300; grep "hlu profile" $O/c5block_sym.err | head -8
echo done
