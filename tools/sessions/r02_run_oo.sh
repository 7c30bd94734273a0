#!/bin/bash
# the occupancy variants as default + the pilot without the largest leaves: timeline, whole GPU suite, fuzz
export TMPDIR=/tmp
O=gpurun_out/r02oo
mkdir -p $O
timeout -k 10 200 python tools/buildprof.py laplace 1000000 4 2> $O/bp.log || exit 1
grep -E "native build timing|timeline|ACA pilot|ACA round" $O/bp.log | tail -n 5 | cut -c1-420
timeout -k 10 300 python tools/buildprof.py helmholtz 1000000 2 2> $O/bph.log || exit 1
grep -E "native build timing" $O/bph.log | tail -n 1 | cut -c1-200
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
echo "suite rc=$?"; tail -n 4 $O/tests.log
PYTHONPATH=. timeout -k 10 260 python tools/fuzz.py 200 71 > $O/fuzz71.log 2>&1
echo "fuzz rc=$?"; tail -n 1 $O/fuzz71.log; grep "FAIL" $O/fuzz71.log | cut -c1-400 | head -n 4
