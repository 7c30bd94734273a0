#!/bin/bash
# A/B: the 1025..2048 class on the workgroup kernel (arena history, three workgroups per CU) instead of the eight-wave register kernel
export TMPDIR=/tmp
O=gpurun_out/r02tt
mkdir -p $O
L=htool_python_amd/lib/libhtool_mi355x.so
cp $L /tmp/default.so
trap "cp /tmp/default.so $L" EXIT   # a failed variant must not stay installed
for v in r1024 default; do
  if [ $v = default ]; then cp /tmp/default.so $L; else cp htool_python_amd/_variants/libhtool_mi355x.$v.so $L; fi
  timeout -k 10 200 python tools/buildprof.py laplace 1000000 3 2> $O/bp_$v.log || exit 1
  echo "$v: $(grep -E 'native build timing' $O/bp_$v.log | grep -oE 'ACA kernels [0-9.]+ s' | tr '\n' ' ') $(grep -E 'ACA round 1' $O/bp_$v.log | tail -n 1 | grep -oE 'retried[^,]*,[^,]*' )"
done
cp htool_python_amd/_variants/libhtool_mi355x.r1024.so $L
cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/fetch -o fetch -- python3 $GRAFT_REPO_ROOT/tools/buildprof.py laplace 1000000 1 2> $GRAFT_REPO_ROOT/$O/fetch.err || exit 1
cd $GRAFT_REPO_ROOT
cp /tmp/default.so $L
