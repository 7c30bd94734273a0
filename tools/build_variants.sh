#!/bin/bash
# A/B builds of the device translation unit: tools/build_variants.sh NAME "-DFLAG=..." [NAME2 "-D..."] ...
# -> htool_python_amd/_variants/NAME/libhtool_mi355x.so (git-ignored; travels to the GPU box).  A GPU script selects one with
#    LD_LIBRARY_PATH=htool_python_amd/_variants/NAME python ...     (the pybind module finds the library through a RUNPATH, which
# LD_LIBRARY_PATH precedes) -- the installed library is never overwritten.  Also kept under the round-2 name
# _variants/libhtool_mi355x.NAME.so for the recorded r02 scripts.
set -e
cd "$(dirname "$0")/.."
P=htool_python_amd
mkdir -p $P/_variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -Wall -Wno-unused-result $flags -c $P/csrc/device.hip -o $P/_variants/device.$name.o
  g++ -shared -o $P/_variants/libhtool_mi355x.$name.so $P/_obj/util.cpp.o $P/_obj/cluster.cpp.o $P/_obj/blocktree.cpp.o $P/_obj/layout.cpp.o $P/_obj/build_host.cpp.o $P/_obj/capi.cpp.o \
      $P/_variants/device.$name.o $P/_obj/dist_device.hip.o $P/_obj/krylov_device.hip.o $P/_obj/dense_device.hip.o -fopenmp -L/opt/rocm/lib -lamdhip64 -lrccl -ldl -Wl,-rpath,/opt/rocm/lib
  rm -f $P/_variants/device.$name.o
  mkdir -p $P/_variants/$name
  cp $P/_variants/libhtool_mi355x.$name.so $P/_variants/$name/libhtool_mi355x.so
  echo "built $name ($flags)"
done
