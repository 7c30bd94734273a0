#!/bin/bash
# round 3, session g: the build without the up-front sort (single-round builds), pack sides on two streams, packed stopping state;
# A/B of the stopping-rule state on the complex kernels; GPU suite
export TMPDIR=/tmp
O=gpurun_out/r03g
mkdir -p $O
timeout -k 10 300 python tools/buildprof.py laplace 1000000 4 > $O/bp_laplace.log 2>&1 || { echo "laplace failed"; tail -n 5 $O/bp_laplace.log; exit 1; }
echo "== laplace"; grep -E "native build timing" $O/bp_laplace.log | sed -e 's/.*block tree/block tree/' | tail -n 3
grep -E "timeline" $O/bp_laplace.log | tail -n 1 | cut -c1-900
for v in default nostop default; do
  if [ $v = default ]; then LP=""; else LP=$PWD/htool_python_amd/_variants/$v; fi
  LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH timeout -k 10 300 python tools/buildprof.py helmholtz 1000000 2 > $O/bp_helmholtz_$v.log 2>&1 || { echo "variant $v failed"; tail -n 5 $O/bp_helmholtz_$v.log; exit 1; }
  echo "== helmholtz $v"; grep -E "native build timing" $O/bp_helmholtz_$v.log | sed -e 's/.*block tree/block tree/' | tail -n 1
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -n 5 $O/gpu_suite.log
