#!/bin/bash
# randomised stress with the transposed product and the 16-wide sweeps in every case, then the whole GPU suite
export TMPDIR=/tmp
O=gpurun_out/r02u
mkdir -p $O
PYTHONPATH=. timeout -k 10 420 python tools/fuzz.py 330 11 > $O/fuzz11.log 2>&1
echo "fuzz rc=$?"
tail -n 2 $O/fuzz11.log
grep -c "^ok" $O/fuzz11.log
grep "FAIL" $O/fuzz11.log | cut -c1-600 | tail -n 5
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
echo "suite rc=$?"
tail -n 5 $O/tests.log
