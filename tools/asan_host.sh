#!/bin/bash
# Host core (cluster tree, block tree, tiles, layout, C ABI, the plan of the hierarchical LU) under AddressSanitizer + UBSan with the CPU tests.
# The kernels' object files are linked as built (GPU sanitizers are not available on the pool).
#   bash tools/asan_host.sh        (needs an up-to-date in-tree build: python -m htool_python_amd.build)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${TMPDIR:-/tmp}/htool_asan
rm -rf "$W" && mkdir -p "$W/pkg/htool_python_amd/lib" "$W/pkg/htool_python_amd/_obj"
CS=$ROOT/htool_python_amd/csrc
for f in util cluster blocktree layout build_host capi hlu_symbolic hlu_capi; do
  g++ -std=c++17 -O1 -g -fPIC -fopenmp -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -c "$CS/$f.cpp" -o "$W/$f.o"
done
g++ -shared -o "$W/pkg/htool_python_amd/lib/libhtool_mi355x.so" "$W"/*.o "$ROOT/htool_python_amd/_obj/device.hip.o" "$ROOT/htool_python_amd/_obj/dist_device.hip.o" \
    "$ROOT/htool_python_amd/_obj/krylov_device.hip.o" "$ROOT/htool_python_amd/_obj/dense_device.hip.o" "$ROOT/htool_python_amd/_obj/cluster_device.hip.o" "$ROOT/htool_python_amd/_obj/hlu_device.hip.o" \
    -fopenmp -fsanitize=address,undefined -L/opt/rocm/lib -lamdhip64 -lrccl -Wl,-rpath,/opt/rocm/lib
cp "$ROOT"/htool_python_amd/*.py "$ROOT"/htool_python_amd/Htool.cpython-*.so "$W/pkg/htool_python_amd/"
cp -r "$ROOT/htool_python_amd/csrc" "$W/pkg/htool_python_amd/"
cp "$ROOT"/htool_python_amd/_obj/*.o "$W/pkg/htool_python_amd/_obj/"
cp -r "$ROOT/Htool" "$ROOT/mpi4py" "$ROOT/oracle" "$ROOT/tests" "$ROOT/include" "$ROOT/tools" "$W/pkg/"
# keep the test fixture from rebuilding: the copies must look newer than the sources
touch "$W"/pkg/htool_python_amd/_obj/*.o; sleep 1; touch "$W/pkg/htool_python_amd/lib/libhtool_mi355x.so"; sleep 1; touch "$W"/pkg/htool_python_amd/Htool.cpython-*.so
cd "$W/pkg"
# libstdc++ has to be preloaded next to libasan, otherwise the first C++ exception trips an ASan-internal check
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0 PYTHONDONTWRITEBYTECODE=1 \
    python -m pytest tests/test_host_logic.py tests/test_independent_checks.py tests/test_hlu_cpu.py -q -p no:cacheprovider
