#!/bin/bash
# round 3, session m: rounds of a multi-round build as cross-sections of all sizes (A/B on the Helmholtz build, same box), C3 test
export TMPDIR=/tmp
O=gpurun_out/r03m
mkdir -p $O
for v in 0 1 0 1; do
  HTOOL_ACA_MIX_ROUNDS=$v timeout -k 10 300 python tools/buildprof.py helmholtz 1000000 2 > $O/bp_mix$v_$RANDOM.log 2>&1 || { echo "mix $v failed"; exit 1; }
  echo "== mix=$v"; grep -E "native build timing" $O/bp_mix$v_*.log | tail -n 1 | sed -e 's/.*block tree/block tree/'
done
ls $O
timeout -k 10 900 python -m pytest tests/test_gpu_native_build.py -m gpu -x -q -k "C3 or multi_round or capacity" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -n 4 $O/tests.log
