#!/bin/bash
# A/B of the occupancy variants of the register ACA kernels (tools/build_variants.sh t1 / t3 / t7: HTOOL_ACA_TUNE bit mask)
export TMPDIR=/tmp
O=gpurun_out/r02nn
mkdir -p $O
L=htool_python_amd/lib/libhtool_mi355x.so
cp $L /tmp/default.so
trap "cp /tmp/default.so $L" EXIT   # a failed variant must not stay installed
for v in t7 t15 t31 default; do
  if [ $v = default ]; then cp /tmp/default.so $L; else cp htool_python_amd/_variants/libhtool_mi355x.$v.so $L; fi
  timeout -k 10 200 python tools/buildprof.py laplace 1000000 3 2> $O/bp_$v.log || exit 1
  echo "$v: $(grep -E 'native build timing' $O/bp_$v.log | tail -n 2 | grep -oE 'ACA kernels [0-9.]+ s' | tr '\n' ' ')"
done
cp htool_python_amd/_variants/libhtool_mi355x.t31.so $L
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -o kt -- python3 $GRAFT_REPO_ROOT/tools/buildprof.py laplace 1000000 2 2> $GRAFT_REPO_ROOT/$O/kt.err || exit 1
cd $GRAFT_REPO_ROOT
cp /tmp/default.so $L
