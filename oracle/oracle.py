"""ctypes front-end of the CPU restatement (oracle/hmat_oracle.cpp) + numpy dense oracle.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (htool_python_amd/, Htool/) never imports it.

Parity status: "parity unpinned" at leaf level (see hmat_oracle.cpp header): the reference's
engine lib/htool is not in /root/reference, so this oracle is pinned by the reference's own
tolerance assertions and by exact dense products from the kernel definition
(example/define_generators.py:14-17), not by golden leaf vectors.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libhmat_oracle.so")
_lib = None

K_INV_DELTA, K_LAPLACE, K_HELMHOLTZ = 0, 1, 2
PCA_REGULAR, PCA_GEOMETRIC, BBOX_REGULAR, BBOX_GEOMETRIC = 0, 1, 2, 3


def build(force=False):
    """Compile the oracle with the Makefile (g++)."""
    src = os.path.join(_HERE, "hmat_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, ci, cd, cc = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_char
        L.orc_cluster_build.restype = vp
        L.orc_cluster_build.argtypes = [vp, ci, ci, vp, vp, ci, ci, vp, ci, ci, ci]
        L.orc_cluster_free.argtypes = [vp]
        L.orc_cluster_n_nodes.argtypes = [vp]
        L.orc_cluster_perm.argtypes = [vp, vp]
        L.orc_cluster_nodes.argtypes = [vp, vp, vp]
        L.orc_cluster_partition_node.argtypes = [vp, ci]
        L.orc_blocktree.argtypes = [vp, vp, cd, cc, cc, ci, ci, ci, vp, vp]
        L.orc_blocktree_get.argtypes = [vp, vp]
        L.orc_aca.argtypes = [ci, ci, vp, vp, cd, ci, ci, ci, vp, vp, cd, ci, ci, vp, vp]
        L.orc_hmat_build.restype = vp
        L.orc_hmat_build.argtypes = [vp, vp, ci, cd, ci, cd, cd, cc, cc, ci, ci, ci, ci]
        L.orc_hmat_free.argtypes = [vp]
        L.orc_hmat_n_leaves.argtypes = [vp]
        L.orc_hmat_leaves.argtypes = [vp, vp]
        L.orc_hmat_leaf_data.argtypes = [vp, ci, vp, vp]
        L.orc_hmat_matvec.argtypes = [vp, vp, vp]
        L.orc_hmat_to_dense.argtypes = [vp, vp]
        L.orc_leaf_loop.argtypes = [ci, ci, vp, vp, vp, ci, vp, vp]
        L.orc_set_num_threads.argtypes = [ci]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _points(coords):
    """(d, N) array -> point-major contiguous copy, as the binding does with f_style|forcecast
    (src/htool/clustering/cluster_tree_builder.hpp:19-23)."""
    c = np.asfortranarray(np.asarray(coords, dtype=np.float64))
    return np.ascontiguousarray(c.T).copy(), c.shape[1], c.shape[0]


class Cluster:
    def __init__(self, coords, n_children=2, size_of_partition=1, partition=None, partition_is_local=False,
                 max_leaf=10, strategy=PCA_REGULAR, radii=None, weights=None):
        self.pts, self.N, self.d = _points(coords)
        part = None
        if partition is not None:
            part = np.asfortranarray(np.asarray(partition, dtype=np.int32))
            part = np.ascontiguousarray(part.T.ravel() if part.ndim == 2 else part).astype(np.int32)
        r = None if radii is None else np.ascontiguousarray(radii, dtype=np.float64)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        self.h = lib().orc_cluster_build(_ptr(self.pts), self.N, self.d, _ptr(r), _ptr(w), n_children, size_of_partition,
                                         _ptr(part), int(partition_is_local), max_leaf, strategy)
        n = lib().orc_cluster_n_nodes(self.h)
        self.perm = np.empty(self.N, dtype=np.int32)
        lib().orc_cluster_perm(self.h, _ptr(self.perm))
        self.inodes = np.empty((n, 7), dtype=np.int32)
        self.dnodes = np.empty((n, 4), dtype=np.float64)
        lib().orc_cluster_nodes(self.h, _ptr(self.inodes), _ptr(self.dnodes))

    def partition_node(self, p):
        return lib().orc_cluster_partition_node(self.h, p)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_cluster_free(self.h)
            self.h = None


def blocktree(tc, sc, eta, symmetry="N", uplo="N", min_t=0, min_s=0, target_partition=-1):
    na, nd = ctypes.c_int(), ctypes.c_int()
    lib().orc_blocktree(tc.h, sc.h, eta, symmetry.encode(), uplo.encode(), min_t, min_s, target_partition,
                        ctypes.byref(na), ctypes.byref(nd))
    adm = np.empty((na.value, 2), dtype=np.int32)
    dns = np.empty((nd.value, 2), dtype=np.int32)
    lib().orc_blocktree_get(_ptr(adm), _ptr(dns))
    return adm, dns


def aca(kind, tpts, spts, p0, rows, cols, eps, is_complex=False, reqrank=-1):
    """ACA of A[rows, cols] (user numbering).  tpts/spts point-major (N, d).  Returns (U m x r, V r x n) or None."""
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    M, N = len(rows), len(cols)
    cap = min(M, N)
    dt = np.complex128 if is_complex else np.float64
    U = np.empty((cap, M), dtype=dt)
    V = np.empty((cap, N), dtype=dt)
    r = lib().orc_aca(kind, tpts.shape[1], _ptr(tpts), _ptr(spts), p0, int(is_complex), M, N, _ptr(rows), _ptr(cols),
                      eps, reqrank, cap, _ptr(U), _ptr(V))
    if r < 0:
        return None
    return U[:r].T.copy(), V[:r].copy()


class HMatrix:
    def __init__(self, tc, sc, kind, p0=0.0, is_complex=False, eps=1e-3, eta=10.0, symmetry="N", uplo="N",
                 reqrank=-1, min_t=0, min_s=0, target_partition=-1, confirm=0):
        if reqrank < 0:
            reqrank = -1 - int(confirm)  # (confirmation steps of the stopping test ride on the reqrank argument, see hmat_oracle.cpp aca)
        self.tc, self.sc, self.is_complex = tc, sc, is_complex
        self.dtype = np.complex128 if is_complex else np.float64
        self.h = lib().orc_hmat_build(tc.h, sc.h, kind, p0, int(is_complex), eps, eta, symmetry.encode(), uplo.encode(),
                                      reqrank, min_t, min_s, target_partition)
        n = lib().orc_hmat_n_leaves(self.h)
        self.leaves = np.empty((n, 5), dtype=np.int32)
        lib().orc_hmat_leaves(self.h, _ptr(self.leaves))

    def leaf_data(self, i):
        t_off, m, s_off, n, r = self.leaves[i]
        if r < 0:
            D = np.empty((n, m), dtype=self.dtype)
            lib().orc_hmat_leaf_data(self.h, i, _ptr(D), None)
            return D.T, None
        U = np.empty((max(r, 1), m), dtype=self.dtype)
        V = np.empty((max(r, 1), n), dtype=self.dtype)
        lib().orc_hmat_leaf_data(self.h, i, _ptr(U), _ptr(V))
        return U[:r].T, V[:r]

    def matvec(self, x):
        x = np.ascontiguousarray(x, dtype=self.dtype)
        y = np.zeros(self.tc.N, dtype=self.dtype)
        lib().orc_hmat_matvec(self.h, _ptr(x), _ptr(y))
        return y

    def to_dense(self):
        out = np.zeros((self.sc.N, self.tc.N), dtype=self.dtype)
        lib().orc_hmat_to_dense(self.h, _ptr(out))
        return out.T

    def algorithmic_elements(self):
        L = self.leaves.astype(np.int64)
        dense = L[:, 4] < 0
        return int((L[dense, 1] * L[dense, 3]).sum() + (L[~dense, 4] * (L[~dense, 1] + L[~dense, 3])).sum())

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_hmat_free(self.h)
            self.h = None


def leaf_loop(leaves, offs, panels, Nt, xp, is_complex=False):
    """CPU leaf loop on externally supplied panels (cluster numbering): yp = sum_leaves leaf * xp."""
    dt = np.complex128 if is_complex else np.float64
    leaves = np.ascontiguousarray(leaves, dtype=np.int32)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    panels = np.ascontiguousarray(panels, dtype=dt)
    xp = np.ascontiguousarray(xp, dtype=dt)
    yp = np.zeros(Nt, dtype=dt)
    lib().orc_leaf_loop(int(is_complex), len(leaves), _ptr(leaves), _ptr(offs), _ptr(panels), Nt, _ptr(xp), _ptr(yp))
    return yp


def usable_cpus():
    """CPUs this process may really use: min(affinity mask, cgroup quota) -- a GPU box hands each
    job a CPU share smaller than the machine."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def num_threads():
    return lib().orc_num_threads()


# ----------------------------------------------------------------------------------------------
# Exact dense oracle in numpy: the same oracle the reference's tests use (generator.mat_vec,
# example/define_generators.py:29-43), vectorised.
# ----------------------------------------------------------------------------------------------
def kernel_block(kind, T, S, p0=0.0):
    """T (d, m), S (d, n) -> A (m, n)."""
    diff = T[:, :, None] - S[:, None, :]
    r = np.sqrt((diff * diff).sum(axis=0))
    if kind == K_INV_DELTA:
        return 1.0 / (p0 + r)
    with np.errstate(divide="ignore", invalid="ignore"):
        if kind == K_LAPLACE:
            return np.where(r > 0, 1.0 / (4 * np.pi * r), 0.0)
        return np.where(r > 0, np.exp(1j * p0 * r) / (4 * np.pi * r), 0.0)


def dense_matvec(kind, T, S, x, p0=0.0, rows=None, chunk=2048):
    """y = A x (or only y[rows]) without materialising A."""
    idx = np.arange(T.shape[1]) if rows is None else np.asarray(rows)
    out = np.zeros((len(idx),) + x.shape[1:], dtype=np.result_type(x.dtype, np.complex128 if kind == K_HELMHOLTZ else np.float64))
    for a in range(0, len(idx), chunk):
        out[a:a + chunk] = kernel_block(kind, T[:, idx[a:a + chunk]], S, p0) @ x
    return out


def points_in_sphere(n):
    """Same draw order as example/create_geometry.py:13-22 (u, theta, phi from the global numpy RNG)."""
    u = np.random.rand(n)
    theta = 2 * np.pi * np.random.rand(n)
    phi = np.arccos(2 * np.random.rand(n) - 1)
    r = np.cbrt(u)
    return np.array([r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)])


def points_in_disk(n):
    """example/create_geometry.py:4-10."""
    u = np.random.rand(n)
    theta = 2 * np.pi * np.random.rand(n)
    return np.array([np.sqrt(u) * np.cos(theta), np.sqrt(u) * np.sin(theta)])


def random_geometries(dimension, nb_rows, nb_cols):
    """example/create_geometry.py:25-37 (seed 0; source cloud shifted by +2 in x)."""
    np.random.seed(0)
    f = points_in_sphere if dimension == 3 else points_in_disk
    t, s = f(nb_rows), f(nb_cols)
    s[0, :] += 2
    return t, s


def leaf_sample_panels(hmatrix, indices, is_complex=False):
    """Download the panels of the given leaves of a (GPU-resident) Htool.HMatrix into the flat format of
    leaf_loop() with one bulk call.  Returns (leaves[k,5], offs[k,2], panels)."""
    L = np.asarray(hmatrix.leaves())
    ids = np.asarray(indices, dtype=np.int64)
    offs, panels = hmatrix.leaf_panels_bulk(ids)
    return np.ascontiguousarray(L[ids], dtype=np.int32).reshape(-1, 5), np.asarray(offs, dtype=np.int64), np.asarray(panels)
