#!/bin/bash
# round 3, session d: lockstep ACA of the largest leaves -- parity tests, then its threshold on both 1 M-point builds; fused GMRES tail
export TMPDIR=/tmp
O=gpurun_out/r03d
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_native_build.py -m gpu -x -q -k "lockstep or device_aca or helmholtz_complex or reqrank or capacity" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -n 6 $O/tests.log
[ $rc -eq 0 ] || exit 1
for k in laplace helmholtz; do
  for t in 0 8192 4096 2048; do
    HTOOL_ACA_STEP_MIN=$t timeout -k 10 300 python tools/buildprof.py $k 1000000 2 > $O/bp_${k}_$t.log 2>&1 || { echo "$k $t failed"; tail -n 5 $O/bp_${k}_$t.log; exit 1; }
    echo "== $k step_min=$t"; grep -E "native build timing" $O/bp_${k}_$t.log | sed -e 's/.*block tree/block tree/' | tail -n 1
    grep -E "ACA round" $O/bp_${k}_$t.log | tail -n 7 | cut -c1-200 | head -n 3
  done
done
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-warm-build > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
run gmres_62k --points 62500 --gmres 50 --force-dist
run gmres_500k --points 500000 --gmres 50 --force-dist
timeout -k 10 300 python -m pytest tests/test_gpu_hmatrix.py -m gpu -x -q -k "gmres or jacobi" > $O/tests2.log 2>&1
echo "tests2 rc=$?"; tail -n 3 $O/tests2.log
