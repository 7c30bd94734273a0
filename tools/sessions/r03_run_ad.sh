#!/bin/bash
# fuzz on the final code (every case now also runs the 16-wide one-triangle / transposed sweeps), replay of the one failure of seed 73
export TMPDIR=/tmp
O=gpurun_out/r03ad; mkdir -p $O
PYTHONPATH=. timeout -k 10 300 python tools/fuzz.py 100000 73 80 > $O/fuzz_73_80_default.log 2>&1; echo "replay (default) rc=$?"; tail -n 3 $O/fuzz_73_80_default.log | cut -c1-260
FUZZ_CONFIRM=1 PYTHONPATH=. timeout -k 10 300 python tools/fuzz.py 100000 73 80 > $O/fuzz_73_80_confirm1.log 2>&1; echo "replay (confirm 1) rc=$?"; tail -n 3 $O/fuzz_73_80_confirm1.log | cut -c1-260
PYTHONPATH=. timeout -k 10 400 python tools/fuzz.py 330 74 > $O/fuzz_74.log 2>&1; echo "fuzz 74 rc=$?"; tail -n 1 $O/fuzz_74.log; grep FAIL $O/fuzz_74.log | cut -c1-400
FUZZ_CONFIRM=1 PYTHONPATH=. timeout -k 10 400 python tools/fuzz.py 250 75 > $O/fuzz_75_confirm1.log 2>&1; echo "fuzz 75 confirm rc=$?"; tail -n 1 $O/fuzz_75_confirm1.log; grep FAIL $O/fuzz_75_confirm1.log | cut -c1-400
