#!/bin/bash
# round 3, session n: randomised stress of the final code: default settings; the lockstep ACA forced onto ordinary leaves; one confirmation step
export TMPDIR=/tmp
O=gpurun_out/r03n
mkdir -p $O
PYTHONPATH=. timeout -k 10 330 python tools/fuzz.py 280 71 > $O/fuzz71.log 2>&1; echo "fuzz71 rc=$?"; tail -n 1 $O/fuzz71.log; grep "FAIL" $O/fuzz71.log | cut -c1-500 | head -n 5
HTOOL_ACA_STEP_MIN=200 PYTHONPATH=. timeout -k 10 330 python tools/fuzz.py 280 72 > $O/fuzz72_step200.log 2>&1; echo "fuzz72 (lockstep > 200) rc=$?"; tail -n 1 $O/fuzz72_step200.log; grep "FAIL" $O/fuzz72_step200.log | cut -c1-500 | head -n 5
FUZZ_CONFIRM=1 PYTHONPATH=. timeout -k 10 330 python tools/fuzz.py 280 11 > $O/fuzz11_confirm1.log 2>&1; echo "fuzz11 (confirm 1; round 2: sheet misses) rc=$?"; tail -n 1 $O/fuzz11_confirm1.log; grep "FAIL" $O/fuzz11_confirm1.log | cut -c1-500 | head -n 5
PYTHONPATH=. timeout -k 10 200 python tools/leak_check.py > $O/leak.log 2>&1; echo "leak rc=$?"; tail -n 2 $O/leak.log | cut -c1-200
