#!/bin/bash
# complex leaves: 8 register terms and two waves per SIMD / two workgroups per CU for the two smallest classes (Helmholtz 1 M)
export TMPDIR=/tmp
O=gpurun_out/r02pp
mkdir -p $O
L=htool_python_amd/lib/libhtool_mi355x.so
cp $L /tmp/default.so
for v in c3 default; do
  if [ $v = default ]; then cp /tmp/default.so $L; else cp htool_python_amd/_variants/libhtool_mi355x.$v.so $L; fi
  timeout -k 10 300 python tools/buildprof.py helmholtz 1000000 3 2> $O/bph_$v.log || exit 1
  echo "$v: $(grep -E 'native build timing' $O/bph_$v.log | grep -oE 'ACA kernels [0-9.]+ s' | tr '\n' ' ')"
done
cp htool_python_amd/_variants/libhtool_mi355x.c3.so $L
timeout -k 10 400 python -m pytest tests/test_gpu_native_build.py tests/test_gpu_hmatrix.py -m gpu -q -x -k "helmholtz or complex or C3" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -n 2 $O/tests.log
cp /tmp/default.so $L
