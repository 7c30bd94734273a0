#!/bin/bash
# first: how allocations behave on a fresh box (first process of the call); then BASELINE config C3 with 1 / 8 / 16 right-hand sides and conjugate-transposed
export TMPDIR=/tmp
mkdir -p gpurun_out/r02w
timeout -k 10 200 python tools/vram_first_touch.py 100 > gpurun_out/r02w/vram.log 2>&1
cat gpurun_out/r02w/vram.log
bash tools/final_benches_helmholtz.sh
