#!/bin/bash
# build timeline after the slim ACA records / fused layout passes, then the whole GPU suite
export TMPDIR=/tmp
O=gpurun_out/r02y
mkdir -p $O
timeout -k 10 200 python tools/buildprof.py laplace 1000000 4 2> $O/bp.log || exit 1
grep -E "native build timing|timeline" $O/bp.log | tail -n 4
timeout -k 10 300 python tools/buildprof.py helmholtz 1000000 2 2> $O/bph.log || exit 1
grep -E "native build timing" $O/bph.log | tail -n 1
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
echo "suite rc=$?"
tail -n 5 $O/tests.log
