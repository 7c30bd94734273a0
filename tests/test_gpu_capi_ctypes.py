"""The C ABI driven directly with ctypes -- no pybind11, no torch -- exactly as INTEGRATION.md section 4 shows:
cluster tree, native generator, build, product (host and multi-RHS), leaf table, statistics, error path."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class BuildParams(ctypes.Structure):
    _fields_ = [("epsilon", ctypes.c_double), ("eta", ctypes.c_double), ("symmetry", ctypes.c_char), ("uplo", ctypes.c_char),
                ("reqrank", ctypes.c_int), ("minimal_target_depth", ctypes.c_int), ("minimal_source_depth", ctypes.c_int),
                ("block_tree_consistency", ctypes.c_int), ("compress", ctypes.c_void_p), ("compress_ctx", ctypes.c_void_p),
                ("dense_blocks", ctypes.c_void_p), ("dense_blocks_ctx", ctypes.c_void_p), ("store_one_triangle", ctypes.c_int)]


def test_c_abi_end_to_end(built, oracle):
    O = oracle
    L = ctypes.CDLL(built[0])
    L.htool_last_error.restype = ctypes.c_char_p
    L.htool_hmatrix_leaf_count.restype = ctypes.c_int64
    L.htool_cluster_permutation.restype = ctypes.POINTER(ctypes.c_int)
    assert L.htool_device_count() >= 1
    N = 5000
    np.random.seed(0)
    points = O.points_in_sphere(N)
    pts = np.ascontiguousarray(points.T)  # point-major
    root, gen, H = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.htool_cluster_create(pts.ctypes, N, 3, None, None, 2, 1, None, 0, 50, 0, ctypes.byref(root)) == 0
    n = ctypes.c_int()
    perm = np.ctypeslib.as_array(L.htool_cluster_permutation(root, ctypes.byref(n)), shape=(N,)).copy()
    assert n.value == N and sorted(perm.tolist()) == list(range(N))
    assert np.array_equal(perm, O.Cluster(points, max_leaf=50).perm)
    assert L.htool_generator_create_native(1, 3, pts.ctypes, N, pts.ctypes, N, ctypes.c_double(0.0), ctypes.byref(gen)) == 0
    p = BuildParams()
    L.htool_build_params_default(ctypes.byref(p))
    p.epsilon, p.eta = 1e-5, 10.0
    assert L.htool_hmatrix_build(gen, root, root, ctypes.byref(p), -1, -1, ctypes.byref(H)) == 0, L.htool_last_error()
    assert (L.htool_hmatrix_nb_rows(H), L.htool_hmatrix_nb_cols(H)) == (N, N)
    x = np.random.rand(N)
    y = np.zeros(N)
    assert L.htool_hmatrix_matvec(H, ctypes.c_char(b"N"), None, x.ctypes, None, y.ctypes) == 0
    ye = O.dense_matvec(O.K_LAPLACE, points, points, x)
    assert np.linalg.norm(y - ye) / np.linalg.norm(ye) < 1e-5
    # y <- alpha H x + beta y
    alpha, beta = ctypes.c_double(2.0), ctypes.c_double(-1.0)
    y2 = y.copy()
    assert L.htool_hmatrix_matvec(H, ctypes.c_char(b"N"), ctypes.byref(alpha), x.ctypes, ctypes.byref(beta), y2.ctypes) == 0
    assert np.allclose(y2, y, rtol=1e-13, atol=0)
    # multi-RHS, column-major
    X = np.asfortranarray(np.random.rand(N, 3))
    Y = np.zeros((N, 3), order="F")
    assert L.htool_hmatrix_matmat(H, ctypes.c_char(b"N"), None, X.ctypes, 3, None, Y.ctypes) == 0
    assert np.linalg.norm(Y - O.dense_matvec(O.K_LAPLACE, points, points, X)) / np.linalg.norm(Y) < 1e-5
    # leaf table and statistics
    nl = L.htool_hmatrix_leaf_count(H)
    leaves = np.zeros((nl, 5), dtype=np.int32)
    L.htool_hmatrix_leaves(H, leaves.ctypes)
    assert (leaves[:, 1].astype(np.int64) * leaves[:, 3]).sum() == N * N
    st = (ctypes.c_int64 * 8)()
    L.htool_hmatrix_stats(H, st)
    d = leaves[:, 4] < 0
    assert st[0] == (leaves[d, 1].astype(np.int64) * leaves[d, 3]).sum() and st[2] == d.sum() and st[5] > 8 * (st[0] + st[1])
    times = (ctypes.c_double * 4)()
    assert L.htool_hmatrix_phase_times(H, times) == 0          # per-phase events are off by default
    assert L.htool_hmatrix_set_phase_timing(H, 1) == 0
    assert L.htool_hmatrix_matvec(H, ctypes.c_char(b"N"), None, x.ctypes, None, y2.ctypes) == 0
    assert L.htool_hmatrix_phase_times(H, times) >= 1 and times[3] > 0
    # errors: transposed products are not implemented; the message is retrievable
    assert L.htool_hmatrix_matvec(H, ctypes.c_char(b"T"), None, x.ctypes, None, y.ctypes) != 0
    assert b"trans='N'" in L.htool_last_error()
    L.htool_hmatrix_destroy(H)
    L.htool_generator_destroy(gen)
    L.htool_cluster_destroy(root)
