#!/bin/bash
# round 3, session b: baseline evidence before the build-kernel work: GMRES per-iteration overhead, C3 (Helmholtz) build under rocprofv3, the GPU suite
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" --no-cpu-baseline > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
run gmres_62k --points 62500 --gmres 50 --force-dist
run gmres_500k --points 500000 --gmres 50 --force-dist
tail -c 700 $O/gmres_62k.json; echo
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o kt -- python3 tools/buildprof.py helmholtz 1000000 2 > $O/buildprof_c3.log 2>&1
echo "c3 trace rc=$?"; grep -E "rep |native build timing" $O/buildprof_c3.log | tail -n 6
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -n 5 $O/gpu_suite.log
