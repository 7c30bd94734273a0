#!/bin/bash
# A/B: two workgroups per CU for the four-wave ACA kernels (register spills) vs one; dense evaluation overlapped with the sort; warm-up build in bench.py
export TMPDIR=/tmp
O=gpurun_out/r02q
mkdir -p $O
L=htool_python_amd/lib/libhtool_mi355x.so
cp $L /tmp/default.so
for v in lb2 default; do
  if [ $v = default ]; then cp /tmp/default.so $L; else cp htool_python_amd/_variants/libhtool_mi355x.$v.so $L; fi
  timeout -k 10 200 python tools/buildprof.py laplace 1000000 3 2> $O/bp_$v.log || exit 1
  grep -E "native build timing" $O/bp_$v.log | tail -n 1
  timeout -k 10 300 python tools/buildprof.py helmholtz 1000000 2 2> $O/bph_$v.log || exit 1
  grep -E "native build timing" $O/bph_$v.log | tail -n 1
done
cp /tmp/default.so $L
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_native_build.py tests/test_gpu_boundary.py tests/test_gpu_capi_ctypes.py -m gpu -q -x > $O/tests.log 2>&1
echo "tests rc=$?"
tail -n 3 $O/tests.log
