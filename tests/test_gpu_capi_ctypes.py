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
                ("dense_blocks", ctypes.c_void_p), ("dense_blocks_ctx", ctypes.c_void_p), ("compress_borrows", ctypes.c_int),
                ("store_one_triangle", ctypes.c_int)]


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
    # transposed product (the kernel is symmetric, the compressed operator only up to epsilon)
    yt = np.zeros(N)
    assert L.htool_hmatrix_matvec(H, ctypes.c_char(b"T"), None, x.ctypes, None, yt.ctypes) == 0, L.htool_last_error()
    assert np.linalg.norm(yt - O.dense_matvec(O.K_LAPLACE, points, points, x)) / np.linalg.norm(yt) < 1e-5
    # errors: the message is retrievable
    assert L.htool_hmatrix_matvec(H, ctypes.c_char(b"X"), None, x.ctypes, None, y.ctypes) != 0
    assert b"trans must be" in L.htool_last_error()
    L.htool_hmatrix_destroy(H)
    L.htool_generator_destroy(gen)
    L.htool_cluster_destroy(root)


class HtoolComm(ctypes.Structure):
    _fields_ = [("rank", ctypes.c_int), ("size", ctypes.c_int), ("ctx", ctypes.c_void_p), ("allgatherv", ctypes.c_void_p), ("rccl", ctypes.c_void_p)]


@pytest.mark.parametrize("padded", [False, True])
def test_c_abi_distributed_device_product_with_rccl_communicator(built, oracle, monkeypatch, padded):
    """The GPU-resident distributed product through the C ABI alone: RCCL bootstrap (unique id ->
    htool_comm_init_rccl), DefaultApproximationBuilder's decomposition (htool_distributed_create_default), then
    htool_distributed_matvec_device / _matmat_device on device buffers -- ncclAllGather of the x slices (zero-copy, or padded
    slices + the compaction kernel) followed by the cluster-numbered local product.  One rank here (RCCL needs one GPU per
    rank); the replicated-vector host API runs over the same communicator."""
    O = oracle
    monkeypatch.setenv("HTOOL_DIST_FORCE_PADDED", "1" if padded else "0")
    L = ctypes.CDLL(built[0])
    hip = ctypes.CDLL("libamdhip64.so.7")
    L.htool_last_error.restype = ctypes.c_char_p
    L.htool_cluster_permutation.restype = ctypes.POINTER(ctypes.c_int)
    L.htool_distributed_hmatrix.restype = ctypes.c_void_p
    N = 6000
    np.random.seed(1)
    points = O.points_in_sphere(N)
    pts = np.ascontiguousarray(points.T)
    uid = ctypes.create_string_buffer(128)
    assert L.htool_rccl_get_unique_id(uid) == 0, L.htool_last_error()
    comm = HtoolComm()
    assert L.htool_comm_init_rccl(uid, 0, 1, ctypes.byref(comm)) == 0, L.htool_last_error()
    assert (comm.rank, comm.size) == (0, 1) and comm.rccl and comm.allgatherv
    root, gen, dist = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.htool_cluster_create(pts.ctypes, N, 3, None, None, 2, 1, None, 0, 64, 0, ctypes.byref(root)) == 0
    n = ctypes.c_int()
    perm = np.ctypeslib.as_array(L.htool_cluster_permutation(root, ctypes.byref(n)), shape=(N,)).copy()
    assert L.htool_generator_create_native(1, 3, pts.ctypes, N, pts.ctypes, N, ctypes.c_double(0.0), ctypes.byref(gen)) == 0
    p = BuildParams()
    L.htool_build_params_default(ctypes.byref(p))
    p.epsilon, p.eta = 1e-5, 10.0
    assert L.htool_distributed_create_default(gen, root, root, ctypes.byref(p), ctypes.byref(comm), ctypes.byref(dist)) == 0, L.htool_last_error()

    def dev_buffer(nbytes):
        ptr = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(nbytes)) == 0
        return ptr

    mu = 3
    X = np.random.rand(mu, N)                                  # row c = right-hand side c, USER numbering
    Xc = np.ascontiguousarray(X[:, perm])                      # this rank's slices: cluster numbering (one rank: everything)
    dX, dY = dev_buffer(Xc.nbytes), dev_buffer(Xc.nbytes)
    assert hip.hipMemcpy(dX, Xc.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(Xc.nbytes), 1) == 0
    Yc = np.zeros_like(Xc)
    # single vector (stream NULL: the operator's own stream), then three columns in one exchange
    assert L.htool_distributed_matvec_device(dist, dX, dY, None) == 0, L.htool_last_error()
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemcpy(Yc.ctypes.data_as(ctypes.c_void_p), dY, ctypes.c_size_t(N * 8), 2) == 0
    y = np.zeros(N)
    y[perm] = Yc[0]
    ye = O.dense_matvec(O.K_LAPLACE, points, points, X[0])
    assert np.linalg.norm(y - ye) / np.linalg.norm(ye) < 1e-5
    assert L.htool_distributed_matmat_device(dist, dX, ctypes.c_int64(N), dY, ctypes.c_int64(N), mu, None) == 0, L.htool_last_error()
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemcpy(Yc.ctypes.data_as(ctypes.c_void_p), dY, ctypes.c_size_t(Xc.nbytes), 2) == 0
    assert np.array_equal(Yc[0][np.argsort(perm)], y)
    for c in range(mu):
        yc = np.zeros(N)
        yc[perm] = Yc[c]
        ye = O.dense_matvec(O.K_LAPLACE, points, points, X[c])
        assert np.linalg.norm(yc - ye) / np.linalg.norm(ye) < 1e-5
    # replicated-vector host API over the same communicator (distributed_operator.hpp:23-37)
    yh = np.zeros(N)
    assert L.htool_distributed_matvec(dist, X[0].ctypes, yh.ctypes) == 0, L.htool_last_error()
    assert np.array_equal(yh, y)
    rows, cols = ctypes.c_int(), ctypes.c_int()
    L.htool_distributed_shape(dist, ctypes.byref(rows), ctypes.byref(cols))
    assert (rows.value, cols.value) == (N, N)
    hip.hipFree(dX)
    hip.hipFree(dY)
    L.htool_distributed_destroy(dist)
    L.htool_comm_destroy_rccl(ctypes.byref(comm))
    L.htool_generator_destroy(gen)
    L.htool_cluster_destroy(root)
