"""Build driver: compiles the host core (g++), the gfx950 kernels (hipcc) and the pybind11 shim in-tree.

    python -m htool_python_amd.build [--force]

Outputs (git-ignored, shipped to the GPU box by gpurun):
    htool_python_amd/lib/libhtool_mi355x.so     C-ABI library (include/htool_mi355x.h)
    htool_python_amd/Htool.cpython-*.so         pybind11 module "Htool"
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libhtool_mi355x.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
ARCH = "gfx950"

HOST_SOURCES = ["util.cpp", "cluster.cpp", "blocktree.cpp", "layout.cpp", "build_host.cpp", "capi.cpp", "hlu_symbolic.cpp", "hlu_capi.cpp"]
HIP_SOURCES = ["device.hip", "dist_device.hip", "krylov_device.hip", "dense_device.hip", "cluster_device.hip", "hlu_device.hip"]
HEADERS = ["common.hpp", "cluster.hpp", "hmatrix.hpp", "device_internal.hpp", "capi_internal.hpp", "device_build.inc", "device_aca_wave.inc", "device_aca_steps.inc", "aca_stop.hpp", "device_recompress.inc", "device_expand.inc", "product_kernels.inc", "product_mfma.inc", "pack_kernels.inc", "device_memory.inc",
           "device_tables.inc", "hlu.hpp", os.path.join("..", "..", "include", "htool_mi355x.h")]


def ext_path():
    return os.path.join(HERE, "Htool" + sysconfig.get_config_var("EXT_SUFFIX"))


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(SRC, h) for h in HEADERS]
    objs = []
    for s in HOST_SOURCES:
        src, obj = os.path.join(SRC, s), os.path.join(OBJ, s + ".o")
        if force or _newer(obj, [src] + hdrs):
            _run(["g++", "-std=c++17", "-O2", "-fPIC", "-fopenmp", "-ffp-contract=off", "-Wall", "-c", src, "-o", obj])
        objs.append(obj)
    for s in HIP_SOURCES:
        src, obj = os.path.join(SRC, s), os.path.join(OBJ, s + ".o")
        if force or _newer(obj, [src] + hdrs):
            _run([HIPCC, "--offload-arch=" + ARCH, "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-result", "-c", src, "-o", obj])
        objs.append(obj)
    if force or _newer(LIB, objs):
        _run(["g++", "-shared", "-o", LIB] + objs + ["-fopenmp", "-L" + os.path.join(ROCM, "lib"), "-lamdhip64", "-lrccl", "-ldl", "-Wl,-rpath," + os.path.join(ROCM, "lib")])
    ext = ext_path()
    pysrc = os.path.join(SRC, "pyhtool.cpp")
    if force or _newer(ext, [pysrc, LIB, os.path.join(HERE, "..", "include", "htool_mi355x.h")]):
        import pybind11
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", pysrc, "-o", ext, "-I" + pybind11.get_include(),
              "-I" + sysconfig.get_paths()["include"], "-L" + LIBDIR, "-lhtool_mi355x", "-Wl,-rpath,$ORIGIN/lib"])
    return LIB, ext


if __name__ == "__main__":
    build(force="--force" in sys.argv)
