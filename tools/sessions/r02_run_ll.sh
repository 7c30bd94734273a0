#!/bin/bash
# A/B: 513..1024 class with 8 register terms and two workgroups per CU (254 registers) instead of 12 terms and one (342)
export TMPDIR=/tmp
O=gpurun_out/r02ll
mkdir -p $O
L=htool_python_amd/lib/libhtool_mi355x.so
cp $L /tmp/default.so
trap "cp /tmp/default.so $L" EXIT   # a failed variant must not stay installed
for v in cl8 default; do
  if [ $v = default ]; then cp /tmp/default.so $L; else cp htool_python_amd/_variants/libhtool_mi355x.$v.so $L; fi
  timeout -k 10 200 python tools/buildprof.py laplace 1000000 3 2> $O/bp_$v.log || exit 1
  grep -E "native build timing" $O/bp_$v.log | tail -n 2 | cut -c1-130
done
cp htool_python_amd/_variants/libhtool_mi355x.cl8.so $L
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -o kt -- python3 $GRAFT_REPO_ROOT/tools/buildprof.py laplace 1000000 2 2> $GRAFT_REPO_ROOT/$O/kt.err || exit 1
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_native_build.py -m gpu -q -x -k "aca or panels or high_accuracy or c1" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -n 2 $O/tests.log
cp /tmp/default.so $L
