#!/bin/bash
# round 3, session f: dense evaluation beside the ACA rounds (timeline), lockstep defaults, first bench line of the round, GPU suite
export TMPDIR=/tmp
O=gpurun_out/r03f
mkdir -p $O
for v in default kc0; do
  if [ $v = default ]; then LP=""; else LP=$PWD/htool_python_amd/_variants/$v; fi
  LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH timeout -k 10 300 python tools/buildprof.py laplace 1000000 3 > $O/bp_laplace_$v.log 2>&1 || { echo "variant $v failed"; tail -n 5 $O/bp_laplace_$v.log; exit 1; }
  echo "== laplace $v"; grep -E "native build timing" $O/bp_laplace_$v.log | sed -e 's/.*block tree/block tree/' | tail -n 2
done
grep -E "timeline" $O/bp_laplace_default.log | tail -n 1 | cut -c1-900
for t in 2048 1024; do
  HTOOL_ACA_STEP_MIN=$t timeout -k 10 300 python tools/buildprof.py helmholtz 1000000 2 > $O/bp_helmholtz_$t.log 2>&1 || { echo "helmholtz $t failed"; exit 1; }
  echo "== helmholtz step_min=$t"; grep -E "native build timing" $O/bp_helmholtz_$t.log | sed -e 's/.*block tree/block tree/' | tail -n 1
done
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json | head -c 700; echo
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -n 5 $O/gpu_suite.log
