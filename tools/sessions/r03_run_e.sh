#!/bin/bash
# round 3, session e: confirmation steps of the ACA stopping test, lockstep thresholds after the register-fused dots, fuzz replays
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_native_build.py -m gpu -x -q -k "confirmation or lockstep or device_aca or helmholtz_complex or capacity or symmetric_operator" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -n 6 $O/tests.log
[ $rc -eq 0 ] || exit 1
for kt in "laplace 8192" "laplace 2048" "laplace 1024" "helmholtz 2048" "helmholtz 1024" "helmholtz 512"; do
  set -- $kt
  HTOOL_ACA_STEP_MIN=$2 timeout -k 10 300 python tools/buildprof.py $1 1000000 2 > $O/bp_$1_$2.log 2>&1 || { echo "$1 $2 failed"; tail -n 5 $O/bp_$1_$2.log; exit 1; }
  echo "== $1 step_min=$2"; grep -E "native build timing" $O/bp_$1_$2.log | sed -e 's/.*block tree/block tree/' | tail -n 1
done
# the sheet case of round 2 on the engine, without and with one confirming step (FUZZ_CONFIRM), next to the CPU oracle
FUZZ_MAX_LOG10_N=5.5 PYTHONPATH=. timeout -k 10 500 python tools/fuzz.py 100000 61 322 > $O/fuzz_61_322.log 2>&1; echo "fuzz 61/322 rc=$?"; tail -n 3 $O/fuzz_61_322.log | cut -c1-300
FUZZ_CONFIRM=1 FUZZ_MAX_LOG10_N=5.5 PYTHONPATH=. timeout -k 10 500 python tools/fuzz.py 100000 61 322 > $O/fuzz_61_322_confirm1.log 2>&1; echo "fuzz 61/322 confirm rc=$?"; tail -n 3 $O/fuzz_61_322_confirm1.log | cut -c1-300
# the case round 2's fuzz was killed in (seed 42, case 43): with the build's DEBUG log, its own time limit
FUZZ_DEBUG_LOG=1 FUZZ_MAX_LOG10_N=5.7 PYTHONPATH=. timeout -k 10 700 python tools/fuzz.py 100000 42 43 > $O/fuzz_42_43.log 2>&1; echo "fuzz 42/43 rc=$?"
grep -c "ACA round" $O/fuzz_42_43.log; grep -E "native build timing|^ok|^FAIL|oracle" $O/fuzz_42_43.log | cut -c1-400 | tail -n 4
