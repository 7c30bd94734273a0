#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02bb
mkdir -p $O
PYTHONPATH=. timeout -k 10 460 python tools/fuzz.py 400 41 > $O/fuzz41.log 2>&1
echo "fuzz41 rc=$?"; tail -n 1 $O/fuzz41.log; grep "FAIL" $O/fuzz41.log | cut -c1-600 | head -n 5
FUZZ_MAX_LOG10_N=5.7 PYTHONPATH=. timeout -k 10 460 python tools/fuzz.py 400 42 > $O/fuzz42.log 2>&1
echo "fuzz42 rc=$?"; tail -n 1 $O/fuzz42.log; grep "FAIL" $O/fuzz42.log | cut -c1-600 | head -n 5
PYTHONPATH=. timeout -k 10 300 python tools/leak_check.py > $O/leak.log 2>&1
echo "leak rc=$?"; tail -n 3 $O/leak.log | cut -c1-300
