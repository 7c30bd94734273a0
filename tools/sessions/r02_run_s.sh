#!/bin/bash
# A/B: eight waves per leaf for the 513..1024 class of the ACA; transposed product at 1 M points
export TMPDIR=/tmp
O=gpurun_out/r02s
mkdir -p $O
L=htool_python_amd/lib/libhtool_mi355x.so
cp $L /tmp/default.so
trap "cp /tmp/default.so $L" EXIT   # a failed variant must not stay installed
for v in e8 default; do
  if [ $v = default ]; then cp /tmp/default.so $L; else cp htool_python_amd/_variants/libhtool_mi355x.$v.so $L; fi
  timeout -k 10 200 python tools/buildprof.py laplace 1000000 3 2> $O/bp_$v.log || exit 1
  grep -E "native build timing" $O/bp_$v.log | tail -n 1
done
cp htool_python_amd/_variants/libhtool_mi355x.e8.so $L
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -o kt -- python3 $GRAFT_REPO_ROOT/tools/buildprof.py laplace 1000000 2 2> $GRAFT_REPO_ROOT/$O/kt.err || exit 1
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_native_build.py -m gpu -q -x -k "aca or panels or rectangular or helmholtz or reqrank or high_accuracy or c1" > $O/tests_e8.log 2>&1
echo "e8 tests rc=$?"
tail -n 3 $O/tests_e8.log
cp /tmp/default.so $L
timeout -k 10 300 python bench.py --no-cpu-baseline --trans T > $O/bench_T.json 2> $O/bench_T.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --trans T --rhs 4 > $O/bench_T4.json 2> $O/bench_T4.err || exit 1
timeout -k 10 400 python bench.py --no-cpu-baseline --kernel helmholtz --trans C > $O/bench_helm_C.json 2> $O/bench_helm_C.err || exit 1
echo done
