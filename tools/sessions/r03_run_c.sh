#!/bin/bash
# round 3, session c: A/B of the complex ACA classes on the Helmholtz build (variants selected by LD_LIBRARY_PATH, nothing overwritten),
# then the pipelined GMRES: per-iteration overhead at 62 500 and 500 000 points, solver tests
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
for v in default c4 c4m3 c8 c12; do
  if [ $v = default ]; then LP=""; else LP=$PWD/htool_python_amd/_variants/$v; fi
  LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH timeout -k 10 300 python tools/buildprof.py helmholtz 1000000 2 > $O/bp_$v.log 2>&1 || { echo "variant $v failed"; tail -n 5 $O/bp_$v.log; exit 1; }
  echo "== $v"; grep -E "native build timing" $O/bp_$v.log | sed -e 's/.*block tree/block tree/' | tail -n 2
done
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-warm-build > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
run gmres_62k --points 62500 --gmres 50 --force-dist
run gmres_62k_nophase --points 62500 --gmres 50 --force-dist --no-phase-timing
run gmres_500k --points 500000 --gmres 50 --force-dist
timeout -k 10 600 python -m pytest tests/test_gpu_hmatrix.py tests/test_distributed.py tests/test_gpu_boundary.py -m gpu -x -q -k "gmres or jacobi or distributed or library or graph" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -n 5 $O/tests.log
