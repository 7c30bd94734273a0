#!/bin/bash
# Helmholtz lines, then the default bench line once more (it quotes the PMC traffic derived from this code's passes)
export TMPDIR=/tmp
bash tools/final_benches_helmholtz.sh || exit 1
timeout -k 10 400 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || exit 1
tail -c 400 gpurun_out/final/bench.json
