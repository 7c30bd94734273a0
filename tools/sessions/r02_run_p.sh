#!/bin/bash
# A/B of the register ACA kernels: guard granularity x evaluation variant (tools/build_variants.sh made the libraries)
export TMPDIR=/tmp
O=gpurun_out/r02p
mkdir -p $O
L=htool_python_amd/lib/libhtool_mi355x.so
cp $L /tmp/default.so
trap "cp /tmp/default.so $L" EXIT   # a failed variant must not stay installed
for v in g1f0 g4f0 g1f1 default; do
  if [ $v = default ]; then cp /tmp/default.so $L; else cp htool_python_amd/_variants/libhtool_mi355x.$v.so $L; fi
  timeout -k 10 200 python tools/buildprof.py laplace 1000000 3 2> $O/bp_$v.log || exit 1
  grep -E "native build timing" $O/bp_$v.log | tail -n 1
done
for v in g1f0 default; do
  if [ $v = default ]; then cp /tmp/default.so $L; else cp htool_python_amd/_variants/libhtool_mi355x.$v.so $L; fi
  timeout -k 10 300 python tools/buildprof.py helmholtz 1000000 2 2> $O/bph_$v.log || exit 1
  grep -E "native build timing" $O/bph_$v.log | tail -n 1
done
cp /tmp/default.so $L
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -o kt -- python3 $GRAFT_REPO_ROOT/tools/buildprof.py laplace 1000000 2 2> $GRAFT_REPO_ROOT/$O/kt.err || exit 1
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_native_build.py tests/test_gpu_hmatrix.py -m gpu -q -x > $O/tests.log 2>&1
echo "tests rc=$?"
tail -n 3 $O/tests.log
