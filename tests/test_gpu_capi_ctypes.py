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
                ("store_one_triangle", ctypes.c_int), ("aca_confirm_steps", ctypes.c_int), ("transposed_products", ctypes.c_int)]


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
    _fields_ = [("rank", ctypes.c_int), ("size", ctypes.c_int), ("ctx", ctypes.c_void_p), ("allgatherv", ctypes.c_void_p), ("rccl", ctypes.c_void_p),
                ("allgather_device", ctypes.c_void_p), ("reduce_scatter_device", ctypes.c_void_p)]


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
    assert (comm.rank, comm.size) == (0, 1) and comm.rccl and comm.allgatherv and comm.allgather_device
    root, gen, dist = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.htool_cluster_create(pts.ctypes, N, 3, None, None, 2, 1, None, 0, 64, 0, ctypes.byref(root)) == 0
    n = ctypes.c_int()
    perm = np.ctypeslib.as_array(L.htool_cluster_permutation(root, ctypes.byref(n)), shape=(N,)).copy()
    assert L.htool_generator_create_native(1, 3, pts.ctypes, N, pts.ctypes, N, ctypes.c_double(0.0), ctypes.byref(gen)) == 0
    p = BuildParams()
    L.htool_build_params_default(ctypes.byref(p))
    p.epsilon, p.eta = 1e-5, 10.0
    assert L.htool_distributed_create_default(gen, root, root, ctypes.byref(p), ctypes.byref(comm), ctypes.byref(dist)) == 0, L.htool_last_error()
    assert L.htool_distributed_exchange_kind(dist, 1) == (2 if padded else 1) and L.htool_distributed_exchange_kind(dist, 3) == 2

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
    # the transposed product with the same distribution (one rank: the reduce-scatter is ncclReduceScatter on one chunk)
    assert comm.reduce_scatter_device
    dZ = dev_buffer(Xc.nbytes)
    assert L.htool_distributed_matmat_device_trans(dist, ctypes.c_char(b"T"), dX, ctypes.c_int64(N), dZ, ctypes.c_int64(N), mu, None) == 0, L.htool_last_error()
    assert hip.hipDeviceSynchronize() == 0
    Zc = np.zeros_like(Xc)
    assert hip.hipMemcpy(Zc.ctypes.data_as(ctypes.c_void_p), dZ, ctypes.c_size_t(Xc.nbytes), 2) == 0
    for c in range(mu):
        zc = np.zeros(N)
        zc[perm] = Zc[c]
        ze = O.dense_matvec(O.K_LAPLACE, points, points, X[c])   # (the Laplace kernel matrix is symmetric)
        assert np.linalg.norm(zc - ze) / np.linalg.norm(ze) < 1e-5
    hip.hipFree(dZ)
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


def _hip():
    hip = ctypes.CDLL("libamdhip64.so.7")
    for f in (hip.hipMemcpy, hip.hipMemcpyAsync, hip.hipMalloc, hip.hipFree, hip.hipDeviceSynchronize, hip.hipMemset):
        f.restype = ctypes.c_int
    return hip


@pytest.mark.parametrize("cplx", [False, True])
def test_compaction_kernel_on_a_synthetic_gathered_buffer(built, cplx):
    """compact_slices_kernel alone (htool_debug_compact_slices): gathered[P=8][mu=2][pad] -> x_full[mu][N] with uneven counts,
    one empty slice, displacements NOT in rank order, NaN in every padding entry, an untouched gap in x_full."""
    L, hip = ctypes.CDLL(built[0]), _hip()
    L.htool_last_error.restype = ctypes.c_char_p
    P, mu = 8, 2
    counts = np.array([300, 257, 0, 511, 512, 1, 64, 130], dtype=np.int32)
    pad = int(counts.max())
    order = [3, 0, 7, 1, 2, 6, 4, 5]            # where the slices lie in x_full
    displs = np.zeros(P, dtype=np.int32)
    pos = 17                                     # leading gap
    for p in order:
        displs[p] = pos
        pos += counts[p] + (9 if p == 7 else 0)  # and a gap behind slice 7
    ldx = pos + 11
    dt = np.complex128 if cplx else np.float64
    rs = np.random.RandomState(5)
    gathered = np.full((P, mu, pad), np.nan, dtype=dt)
    expect = np.full((mu, ldx), -7.0, dtype=dt)
    for p in range(P):
        for c in range(mu):
            v = rs.rand(counts[p]) + (1j * rs.rand(counts[p]) if cplx else 0)
            gathered[p, c, : counts[p]] = v
            expect[c, displs[p]: displs[p] + counts[p]] = v
    d_g, d_x = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_g), ctypes.c_size_t(gathered.nbytes)) == 0 and hip.hipMalloc(ctypes.byref(d_x), ctypes.c_size_t(expect.nbytes)) == 0
    init = np.full((mu, ldx), -7.0, dtype=dt)
    assert hip.hipMemcpy(d_g, gathered.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(gathered.nbytes), 1) == 0
    assert hip.hipMemcpy(d_x, init.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(init.nbytes), 1) == 0
    assert L.htool_debug_compact_slices(d_g, d_x, counts.ctypes, displs.ctypes, P, pad, mu, ctypes.c_int64(ldx), int(cplx), None) == 0, L.htool_last_error()
    out = np.zeros_like(init)
    assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), d_x, ctypes.c_size_t(out.nbytes), 2) == 0
    assert np.array_equal(out, expect)
    # a slice that would leave x_full is refused before any launch
    bad = displs.copy()
    bad[4] = ldx - 5
    assert L.htool_debug_compact_slices(d_g, d_x, counts.ctypes, bad.ctypes, P, pad, mu, ctypes.c_int64(ldx), int(cplx), None) != 0
    assert b"out of range" in L.htool_last_error()
    hip.hipFree(d_g)
    hip.hipFree(d_x)


ALLGATHER_DEVICE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)


@pytest.mark.parametrize("n_points,rank,padded", [(6000, 3, False), (6000, 5, True), (6001, 0, False), (6001, 7, False)])
def test_c_abi_rank_r_of_eight_with_the_host_languages_own_device_allgather(built, oracle, monkeypatch, n_points, rank, padded):
    """ONE process plays rank r of an 8-rank run: htool_comm.allgather_device is a ctypes callback that delivers the other
    ranks' slices (prepared on the host, NaN in their padding) and copies this rank's from the send buffer the library
    hands over.  Exercises dist_state's counts / displacements for P = 8, the zero-copy layout (6000 = 8 x 750 points), the
    padded layout + compaction (6001 points, or forced), one and three columns; results against exact rows of the dense
    operator and, bitwise, against the same H-matrix applied to the whole vector."""
    O = oracle
    monkeypatch.setenv("HTOOL_DIST_FORCE_PADDED", "1" if padded else "0")
    L, hip = ctypes.CDLL(built[0]), _hip()
    L.htool_last_error.restype = ctypes.c_char_p
    L.htool_cluster_permutation.restype = ctypes.POINTER(ctypes.c_int)
    L.htool_distributed_hmatrix.restype = ctypes.c_void_p
    P, N = 8, n_points
    np.random.seed(2)
    points = O.points_in_sphere(N)
    pts = np.ascontiguousarray(points.T)
    root, gen, dist = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.htool_cluster_create(pts.ctypes, N, 3, None, None, 2, P, None, 0, 64, 0, ctypes.byref(root)) == 0, L.htool_last_error()
    n = ctypes.c_int()
    perm = np.ctypeslib.as_array(L.htool_cluster_permutation(root, ctypes.byref(n)), shape=(N,)).copy()
    assert L.htool_generator_create_native(1, 3, pts.ctypes, N, pts.ctypes, N, ctypes.c_double(0.0), ctypes.byref(gen)) == 0
    prm = BuildParams()
    L.htool_build_params_default(ctypes.byref(prm))
    prm.epsilon, prm.eta = 1e-5, 10.0
    state = {"calls": 0, "mu": 1, "slices": None, "bytes": []}

    def allgather_device(ctx, send_dev, recv_dev, nbytes, stream):
        state["calls"] += 1
        state["bytes"].append(nbytes)
        bufs = state["slices"]  # per rank: [mu][pad] (padded layout) -- for one column also what the zero-copy layout sends
        for p in range(P):
            dst = ctypes.c_void_p(recv_dev + p * nbytes)
            if p == rank:
                rc = hip.hipMemcpyAsync(dst, ctypes.c_void_p(send_dev), ctypes.c_size_t(nbytes), 3, ctypes.c_void_p(stream))
            else:
                assert bufs[p].nbytes >= nbytes
                rc = hip.hipMemcpyAsync(dst, bufs[p].ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nbytes), 1, ctypes.c_void_p(stream))
            if rc != 0:
                return 1
        return 0

    hook = ALLGATHER_DEVICE_FN(allgather_device)
    comm = HtoolComm(rank, P, None, None, None, ctypes.cast(hook, ctypes.c_void_p), None)
    assert L.htool_distributed_create_default(gen, root, root, ctypes.byref(prm), ctypes.byref(comm), ctypes.byref(dist)) == 0, L.htool_last_error()
    parts = []
    for p in range(P):
        o, sz = ctypes.c_int(), ctypes.c_int()
        assert L.htool_distributed_partition(dist, p, ctypes.byref(o), ctypes.byref(sz)) == 0
        parts.append((o.value, sz.value))
    assert sum(sz for _, sz in parts) == N and [o for o, _ in parts] == list(np.cumsum([0] + [sz for _, sz in parts[:-1]]))
    even = len({sz for _, sz in parts}) == 1
    assert even == (N % P == 0)
    pad = max(sz for _, sz in parts)
    off, mine = parts[rank]
    h = ctypes.c_void_p(L.htool_distributed_hmatrix(dist))

    def dev(arr):
        ptr = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(max(arr.nbytes, 8))) == 0
        assert hip.hipMemcpy(ptr, arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(arr.nbytes), 1) == 0
        return ptr

    for mu in (1, 3):
        want = 1 if (even and not padded and mu == 1) else 2
        assert L.htool_distributed_exchange_kind(dist, mu) == want
        X = np.random.rand(mu, N)                 # USER numbering
        Xc = np.ascontiguousarray(X[:, perm])     # cluster numbering
        slices = []
        for p in range(P):
            b = np.full((mu, pad), np.nan)
            b[:, : parts[p][1]] = Xc[:, parts[p][0]: parts[p][0] + parts[p][1]]
            slices.append(b)
        state.update(mu=mu, slices=slices)
        ldx, ldy = mine + 13, mine + 1
        x_loc = np.full((mu, ldx), np.nan)
        x_loc[:, :mine] = Xc[:, off: off + mine]
        d_x, d_y = dev(x_loc), dev(np.zeros((mu, ldy)))
        calls = state["calls"]
        if mu == 1:
            assert L.htool_distributed_matvec_device(dist, d_x, d_y, None) == 0, L.htool_last_error()
        else:
            assert L.htool_distributed_matmat_device(dist, d_x, ctypes.c_int64(ldx), d_y, ctypes.c_int64(ldy), mu, None) == 0, L.htool_last_error()
        assert hip.hipDeviceSynchronize() == 0
        assert state["calls"] == calls + 1 and state["bytes"][-1] == pad * mu * 8
        Y = np.zeros((mu, ldy))
        assert hip.hipMemcpy(Y.ctypes.data_as(ctypes.c_void_p), d_y, ctypes.c_size_t(Y.nbytes), 2) == 0
        assert not np.isnan(Y).any()
        rows = perm[off: off + mine]
        for c in range(mu):
            ye = O.dense_matvec(O.K_LAPLACE, points, points, X[c], rows=rows)
            assert np.linalg.norm(Y[c, :mine] - ye) / np.linalg.norm(ye) < 1e-5
        # bitwise: the same H-matrix on the whole cluster-numbered vector (numbering 1), no exchange
        d_xf, d_yr = dev(Xc), dev(np.zeros((mu, mine)))
        assert L.htool_hmatrix_matmat_device(h, d_xf, ctypes.c_int64(N), d_yr, ctypes.c_int64(mine), mu, 1, None) == 0, L.htool_last_error()
        assert hip.hipDeviceSynchronize() == 0
        Yr = np.zeros((mu, mine))
        assert hip.hipMemcpy(Yr.ctypes.data_as(ctypes.c_void_p), d_yr, ctypes.c_size_t(Yr.nbytes), 2) == 0
        assert np.array_equal(Yr, Y[:, :mine])
        for ptr in (d_x, d_y, d_xf, d_yr):
            hip.hipFree(ptr)
    # a failing hook is reported, not ignored
    state["slices"] = None
    bad = ALLGATHER_DEVICE_FN(lambda *a: 1)
    comm2 = HtoolComm(rank, P, None, None, None, ctypes.cast(bad, ctypes.c_void_p), None)
    dist2 = ctypes.c_void_p()
    assert L.htool_distributed_create_default(gen, root, root, ctypes.byref(prm), ctypes.byref(comm2), ctypes.byref(dist2)) == 0, L.htool_last_error()
    d_x, d_y = dev(np.zeros(pad)), dev(np.zeros(pad))
    assert L.htool_distributed_matvec_device(dist2, d_x, d_y, None) != 0
    assert b"allgather_device failed" in L.htool_last_error()
    hip.hipFree(d_x)
    hip.hipFree(d_y)
    L.htool_distributed_destroy(dist2)
    L.htool_distributed_destroy(dist)
    L.htool_generator_destroy(gen)
    L.htool_cluster_destroy(root)
