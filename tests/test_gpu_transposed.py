"""Transposed products y = H^T x / H^H x through the C ABI (`trans` of htool_hmatrix_matvec / matmat; the reference passes the
flag through from lu_solve and the local-operator hooks, src/htool/hmatrix/hmatrix.hpp:64-78,
src/htool/local_operator/virtual_local_to_local_operator.hpp).  Checked against the exact dense operator of the oracle, against
the H-matrix's own dense expansion (same panels, other kernels), by the adjoint identity, and for the bookkeeping around the lazy
tables (the direct product before and after, copies, row-split operators, several right-hand sides, repeatability)."""
import copy
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("leaf,eps", [(10, 1e-6), (64, 1e-4)])
def test_transposed_product_of_a_rectangular_operator(built, oracle, leaf, eps):
    import Htool
    from tests.helpers import NumpyGenerator, cluster_of

    O = oracle
    rng = np.random.RandomState(3)
    T, S = rng.random_sample((3, 1500)), rng.random_sample((3, 700)) + np.array([[0.3], [0.0], [0.0]])
    tcl, scl = cluster_of(T, leaf), cluster_of(S, leaf)
    gen = NumpyGenerator(T, S)
    H = Htool.HMatrixTreeBuilder(eps, 10.0, "N", "N").build(gen, tcl, scl)
    x_s, x_t = rng.random_sample(700), rng.random_sample(1500)
    y_before = H * x_s
    A = O.kernel_block(O.K_INV_DELTA, T, S, 0.1)
    D = np.asarray(H.to_dense_in_user_numbering())
    yt = H.transposed_mul(x_t)                         # first call: makes the tables
    assert yt.shape == (700,)
    assert _rel(yt, A.T @ x_t) < eps                   # against the exact operator, the reference's tolerance
    assert _rel(yt, D.T @ x_t) < 1e-12                 # against the same panels expanded by the direct kernels
    assert np.array_equal(yt, H.transposed_mul(x_t))   # fixed summation order
    assert np.array_equal(yt, H.transposed_mul(x_t, "C"))  # real operator: 'C' is 'T'
    # the direct product is untouched by the re-indexing (its tables were written again)
    assert np.array_equal(H * x_s, y_before)
    # adjoint identity
    assert abs(x_t @ (H * x_s) - yt @ x_s) < 1e-11 * abs(x_t @ y_before)
    # several right-hand sides: sweeps of 4, 2, 1 columns
    for mu in (2, 3, 5, 7):
        X = np.asfortranarray(rng.random_sample((1500, mu)))
        Y = np.asarray(H.transposed_mul(X))
        assert Y.shape == (700, mu)
        for c in range(mu):
            assert _rel(Y[:, c], H.transposed_mul(np.ascontiguousarray(X[:, c]))) < 1e-13
        assert _rel(Y, D.T @ X) < 1e-12
    # a deep copy carries the tables along; a copy made BEFORE them makes its own
    H2 = copy.deepcopy(H)
    assert np.array_equal(H2.transposed_mul(x_t), yt)
    assert np.array_equal(H2 * x_s, y_before)
    with pytest.raises(RuntimeError, match="Wrong size"):
        H.transposed_mul(x_s)


def test_transposed_product_native_square_and_multi_batch(built, oracle, monkeypatch):
    """Native generator (device ACA), 20 000 points, small arena: several pack batches, so several segments per tile."""
    import Htool
    from htool_python_amd.workloads import points_in_sphere
    from tests.helpers import cluster_of

    O = oracle
    monkeypatch.setenv("HTOOL_BUILD_ARENA_MB", "40")
    P = points_in_sphere(20000, seed=5)
    cl = cluster_of(P, 50)
    H = Htool.HMatrixTreeBuilder(1e-4, 10.0, "N", "N").build(Htool.NativeGenerator("laplace", P, P), cl, cl)
    assert H.stats()["n_low_rank"] > 1000
    rng = np.random.RandomState(0)
    x, w = rng.random_sample(20000), rng.random_sample(20000)
    y = H * x
    z = H.transposed_mul(w)
    rows = rng.choice(20000, 64, replace=False)
    # columns `rows` of A^T w = rows of A^T: exact dense rows of the (symmetric-kernel) operator with roles swapped
    exact = O.kernel_block(O.K_LAPLACE, P[:, rows], P, 0.0) @ w
    assert _rel(z[rows], exact) < 1e-4
    assert abs(w @ y - z @ x) < 1e-10 * abs(w @ y)
    assert np.array_equal(H * x, y)
    assert _rel(z, H * w) < 1e-3  # (symmetric kernel, one cluster tree: H^T = H up to the approximation)


def test_transposed_product_complex(built, oracle):
    import Htool
    from tests.helpers import ComplexNumpyGenerator, cluster_of

    O = oracle
    rng = np.random.RandomState(1)
    T, S = rng.random_sample((3, 900)), rng.random_sample((3, 1300))
    tcl, scl = cluster_of(T, 20), cluster_of(S, 20)
    gen = ComplexNumpyGenerator(T, S, 5.0)
    H = Htool.ComplexHMatrixTreeBuilder(1e-5, 10.0, "N", "N").build(gen, tcl, scl)
    A = O.kernel_block(O.K_HELMHOLTZ, T, S, 5.0)
    D = np.asarray(H.to_dense_in_user_numbering())
    x = rng.random_sample(900) + 1j * rng.random_sample(900)
    yT, yC = H.transposed_mul(x, "T"), H.transposed_mul(x, "C")
    assert _rel(yT, A.T @ x) < 1e-5 and _rel(yC, A.conj().T @ x) < 1e-5
    assert _rel(yT, D.T @ x) < 1e-12 and _rel(yC, D.conj().T @ x) < 1e-12
    X = np.asfortranarray(rng.random_sample((900, 3)) + 1j * rng.random_sample((900, 3)))
    assert _rel(np.asarray(H.transposed_mul(X, "C")), D.conj().T @ X) < 1e-12
    xs = rng.random_sample(1300) + 0j
    assert _rel(H * xs, D @ xs) < 1e-12


def test_transposed_product_of_one_triangle_storage(built, oracle):
    import Htool
    from tests.helpers import NumpyGenerator, cluster_of

    rng = np.random.RandomState(2)
    P = rng.random_sample((3, 1200))
    cl = cluster_of(P, 10)
    H = Htool.HMatrixTreeBuilder(1e-5, 10.0, "S", "L").build(NumpyGenerator(P, P), cl, cl)
    assert H.is_one_triangle()
    x = rng.random_sample(1200)
    assert np.array_equal(H.transposed_mul(x), H * x)  # H = H^T: the stored operator is the answer


def test_transposed_products_of_a_row_split_sum_to_the_whole(built, oracle):
    """rank p's block rows(p) x all columns: sum_p H_p^T x_p = A^T x (what a distributed transposed product would reduce)."""
    import Htool
    from tests.helpers import NumpyGenerator

    O = oracle
    rng = np.random.RandomState(4)
    P = rng.random_sample((3, 2000))
    cb = Htool.ClusterTreeBuilder()
    cb.set_maximal_leaf_size(10)
    cl = cb.create_cluster_tree(P, 2, size_of_partition=4)
    gen = NumpyGenerator(P, P)
    perm = np.asarray(cl.get_permutation())
    x = rng.random_sample(2000)
    total = np.zeros(2000)
    for p in range(4):
        Hp = Htool.HMatrixTreeBuilder(1e-6, 10.0, "N", "N").build(gen, cl, cl, p)
        part = cl.get_cluster_on_partition(p)
        rows = perm[part.get_offset():part.get_offset() + part.get_size()]
        # the block takes its rows in cluster order (as its direct product returns them) and answers in user numbering
        zp = Hp.transposed_mul(np.ascontiguousarray(x[rows]))
        assert zp.shape == (2000,)
        Ap = O.kernel_block(O.K_INV_DELTA, P[:, rows], P, 0.1)
        assert _rel(zp, Ap.T @ x[rows]) < 1e-6
        total += zp
    assert _rel(total, O.kernel_block(O.K_INV_DELTA, P, P, 0.1).T @ x) < 1e-6


def test_transposed_product_through_the_c_abi_with_scaling(built, oracle):
    """alpha / beta of htool_hmatrix_matvec with trans = 'T', driven with ctypes (no pybind, no torch)."""
    import Htool
    from tests.helpers import NumpyGenerator, cluster_of

    lib_path, _ = built
    L = ctypes.CDLL(lib_path)
    L.htool_last_error.restype = ctypes.c_char_p
    rng = np.random.RandomState(6)
    T, S = rng.random_sample((3, 600)), rng.random_sample((3, 350))
    H = Htool.HMatrixTreeBuilder(1e-6, 10.0, "N", "N").build(NumpyGenerator(T, S), cluster_of(T, 10), cluster_of(S, 10))
    D = np.asarray(H.to_dense_in_user_numbering())
    h = ctypes.c_void_p(H._handle)
    x, y = rng.random_sample(600), rng.random_sample(350)
    y0 = y.copy()
    alpha, beta = ctypes.c_double(-1.5), ctypes.c_double(0.25)
    assert L.htool_hmatrix_matvec(h, ctypes.c_char(b"T"), ctypes.byref(alpha), x.ctypes, ctypes.byref(beta), y.ctypes) == 0, L.htool_last_error()
    assert _rel(y, -1.5 * (D.T @ x) + 0.25 * y0) < 1e-12
    assert L.htool_hmatrix_matvec(h, ctypes.c_char(b"Q"), None, x.ctypes, None, y.ctypes) != 0
    assert b"trans must be" in L.htool_last_error()


def test_local_operator_hook_with_trans(built, oracle):
    """The H-matrix block behind a local-to-local operator answers the hook's trans argument
    (src/htool/local_operator/virtual_local_to_local_operator.hpp: add_vector_product(trans, ...))."""
    import Htool
    from htool_python_amd.distributed import _HMatrixLocalToLocal
    from tests.helpers import NumpyGenerator, cluster_of

    rng = np.random.RandomState(7)
    P = rng.random_sample((3, 500))
    cl = cluster_of(P, 10)
    H = Htool.HMatrixTreeBuilder(1e-6, 10.0, "N", "N").build(NumpyGenerator(P, P), cl, cl)
    D = np.asarray(H.to_dense_in_user_numbering())
    op = _HMatrixLocalToLocal(H, None, None)
    x = rng.random_sample(500)
    out = np.ones(500)
    op.local_add_vector_product("T", 2.0, x, 0.5, out)
    assert _rel(out, 2.0 * (D.T @ x) + 0.5) < 1e-12
    X = rng.random_sample((500, 3))
    out2 = np.zeros((500, 3))
    op.local_add_matrix_product_row_major("T", 1.0, X, 0.0, out2)
    assert _rel(out2, D.T @ X) < 1e-12


def test_transposed_product_after_recompression_and_after_reload(built, oracle, tmp_path, monkeypatch):
    """The tables of the transposed product are made from the batches as they are NOW: after a recompression in several chunks
    (every chunk becomes a batch of its own) and for an operator loaded from a checkpoint (one batch, no generator)."""
    import Htool
    from htool_python_amd.workloads import points_in_sphere
    from tests.helpers import cluster_of

    monkeypatch.setenv("HTOOL_RECOMPRESS_ARENA_MB", "20")
    P = points_in_sphere(12000, seed=9)
    cl = cluster_of(P, 40)
    H = Htool.HMatrixTreeBuilder(1e-6, 10.0, "N", "N").build(Htool.NativeGenerator("laplace", P, P), cl, cl)
    rng = np.random.RandomState(1)
    w, x = rng.random_sample(12000), rng.random_sample(12000)
    z0 = H.transposed_mul(w)                      # tables exist BEFORE the recompression, too
    assert Htool.recompression(H, 1e-4) > 0
    z1 = H.transposed_mul(w)
    y1 = H * x
    assert _rel(z1, z0) < 1e-3 and not np.array_equal(z1, z0)
    assert abs(w @ y1 - z1 @ x) < 1e-10 * abs(w @ y1)
    # the other order: recompressed first (chunks -> several batches), the tables made afterwards
    Hb = Htool.HMatrixTreeBuilder(1e-6, 10.0, "N", "N").build(Htool.NativeGenerator("laplace", P, P), cl, cl)
    assert Htool.recompression(Hb, 1e-4) > 0
    assert _rel(Hb.transposed_mul(w), z1) < 1e-12 and _rel(Hb * x, y1) < 1e-12
    del Hb
    path = str(tmp_path / "h.npz")
    Htool.save_hmatrix(path, H)
    H2 = Htool.load_hmatrix(path, cl)
    assert _rel(H2 * x, y1) < 1e-12
    z2 = H2.transposed_mul(w)
    assert _rel(z2, z1) < 1e-12
    assert abs(w @ (H2 * x) - z2 @ x) < 1e-10 * abs(w @ y1)


@pytest.mark.parametrize("case", ["rect_leaf10", "rect_leaf64", "native_multi_batch", "complex", "partition"])
def test_transposed_product_sixteen_wide(built, oracle, case, monkeypatch):
    """Y = H^T X / H^H X for more than eight right-hand sides: one sweep of the panels on the matrix cores per sixteen columns
    (operands transposed by MFMAs with selection matrices), against the sweeps of one column (<= 1e-13), the H-matrix's own
    dense expansion and the adjoint identity; rectangular, several pack batches, complex ('T' and 'C'), a row-partition share."""
    import Htool
    from tests.helpers import ComplexNumpyGenerator, NumpyGenerator, cluster_of

    O = oracle
    rng = np.random.RandomState(7)
    trans = ("T",)
    if case.startswith("rect"):
        leaf = int(case[9:])
        T, S = rng.random_sample((3, 1500)), rng.random_sample((3, 700)) + np.array([[0.3], [0.0], [0.0]])
        H = Htool.HMatrixTreeBuilder(1e-6, 10.0, "N", "N").build(NumpyGenerator(T, S), cluster_of(T, leaf), cluster_of(S, leaf))
        cplx = False
    elif case == "native_multi_batch":
        from htool_python_amd.workloads import points_in_sphere
        monkeypatch.setenv("HTOOL_BUILD_ARENA_MB", "40")
        T = S = points_in_sphere(20000, seed=5)
        cl = cluster_of(T, 50)
        H = Htool.HMatrixTreeBuilder(1e-4, 10.0, "N", "N").build(Htool.NativeGenerator("laplace", T, S), cl, cl)
        cplx = False
    elif case == "complex":
        T, S = rng.random_sample((3, 900)), rng.random_sample((3, 1300))
        H = Htool.ComplexHMatrixTreeBuilder(1e-5, 10.0, "N", "N").build(ComplexNumpyGenerator(T, S, 5.0), cluster_of(T, 20), cluster_of(S, 20))
        cplx = True
        trans = ("T", "C")
    else:  # the rows of one partition member against all columns
        T = S = O.points_in_sphere(6000)
        b = Htool.ClusterTreeBuilder()
        b.set_maximal_leaf_size(60)
        tcl = b.create_cluster_tree(T, 2, size_of_partition=3)
        H = Htool.HMatrixTreeBuilder(1e-5, 10.0, "N", "N").build(Htool.NativeGenerator("laplace", T, S), tcl, tcl, 1)
        cplx = False
    nt, ns = H.shape
    for tr in trans:
        for mu in (9, 16, 21):
            X = rng.random_sample((nt, mu))
            if cplx:
                X = X + 1j * rng.random_sample((nt, mu))
            X = np.asfortranarray(X)
            Y = np.asarray(H.transposed_mul(X, tr))
            assert Y.shape == (ns, mu)
            assert np.array_equal(np.asarray(H.transposed_mul(X, tr)), Y)
            for c in range(mu):
                yc = H.transposed_mul(np.ascontiguousarray(X[:, c]), tr)
                assert _rel(Y[:, c], yc) < 1e-13, (case, tr, mu, c)
    # adjoint identity with the direct 16-wide sweep: <X, H Z> = <H^T X, Z>
    Z = rng.random_sample((ns, 16)) + (1j * rng.random_sample((ns, 16)) if cplx else 0)
    Z = np.asfortranarray(Z)
    X = np.asfortranarray(X[:, :16])
    lhs = np.sum(X * (H @ Z))
    rhs = np.sum(np.asarray(H.transposed_mul(X, "T")) * Z)
    assert abs(lhs - rhs) < 1e-11 * abs(lhs)
