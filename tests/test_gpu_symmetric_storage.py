"""GPU parity tests of one-triangle symmetric storage (the default; `HMatrixTreeBuilder.set_symmetric_storage`):
symmetry 'S' keeps the UPLO triangle only, as the reference does, and the product uses every
stored off-diagonal leaf twice in one fused sweep.  Checked against the CPU restatement, which stores the same
triangle, and against the exact dense kernel.
"""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sym_build(pts, kind, p0, eps, eta, leaf, uplo, complex_=False, one_triangle=True, native=True):
    import Htool
    from tests.helpers import cluster_of, NumpyGenerator

    cl = cluster_of(pts, leaf)
    name = {0: "inv_delta", 1: "laplace", 2: "helmholtz"}[kind]
    if complex_:
        gen = Htool.ComplexNativeGenerator(name, pts, pts, p0)
        builder = Htool.ComplexHMatrixTreeBuilder(eps, eta, "S", uplo)
    else:
        gen = Htool.NativeGenerator(name, pts, pts, p0) if native else NumpyGenerator(pts, pts, kind, p0)
        builder = Htool.HMatrixTreeBuilder(eps, eta, "S", uplo)
    builder.set_symmetric_storage(one_triangle)
    return builder.build(gen, cl, cl), cl, gen


@pytest.mark.parametrize("n,leaf,eta,eps,kind,p0,uplo", [
    (3000, 10, 10.0, 1e-3, 0, 0.1, "L"),
    (3000, 10, 10.0, 1e-3, 0, 0.1, "U"),
    (9000, 100, 10.0, 1e-4, 1, 0.0, "L"),
    (7001, 300, 3.0, 1e-5, 1, 0.0, "U"),   # leaves above the tile size: clusters cut into several tiles
    (5000, 33, 100.0, 1e-3, 1, 0.0, "L"),
])
def test_one_triangle_matches_cpu_storage_and_product(built, oracle, n, leaf, eta, eps, kind, p0, uplo):
    O = oracle
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    H, cl, _ = _sym_build(pts, kind, p0, eps, eta, leaf, uplo)
    oc = O.Cluster(pts, max_leaf=leaf)
    OH = O.HMatrix(oc, oc, kind, p0, eps=eps, eta=eta, symmetry="S", uplo=uplo)
    mine = {tuple(l[:4]): int(l[4]) for l in np.asarray(H.leaves())}
    theirs = {tuple(l[:4]): int(l[4]) for l in OH.leaves}
    assert set(mine) == set(theirs)          # the same triangle is stored, leaf for leaf
    for (to, m, so, nn) in mine:              # nothing strictly in the other triangle
        assert (so < to + m) if uplo == "L" else (to < so + nn)
    diff = np.array([mine[k] - theirs[k] for k in mine])
    assert np.mean(diff != 0) < 0.01 and np.abs(diff).max() <= 1
    np.random.seed(1)
    x = np.random.rand(n)
    y, y_cpu = H * x, OH.matvec(x)
    y_exact = O.dense_matvec(kind, pts, pts, x, p0)
    e_gpu = np.linalg.norm(y - y_exact) / np.linalg.norm(y_exact)
    e_cpu = np.linalg.norm(y_cpu - y_exact) / np.linalg.norm(y_exact)
    assert e_gpu < eps
    assert e_gpu < 1.5 * e_cpu + 1e-14
    assert np.linalg.norm(y - y_cpu) / np.linalg.norm(y_cpu) < 2 * eps
    # bitwise reproducible (fixed summation order, no atomics)
    assert np.array_equal(H * x, y)
    # several right-hand sides: fused sweeps of 4, 2 and 1 columns (other reduction order than the single-column kernel)
    X = np.asfortranarray(np.random.rand(n, 7))
    Y = H @ X
    for j in range(7):
        yj = H * np.ascontiguousarray(X[:, j])
        assert np.linalg.norm(Y[:, j] - yj) / np.linalg.norm(yj) < 1e-13
    assert np.array_equal(H @ X, Y)


def test_one_triangle_equals_two_triangle_operator(built, oracle):
    """Both storages hold the same leaves on the stored triangle (bitwise) and give the same product to rounding."""
    O = oracle
    n, leaf, eps, eta = 6000, 50, 1e-4, 10.0
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    H1, cl, _ = _sym_build(pts, 1, 0.0, eps, eta, leaf, "L", one_triangle=True)
    H2, _, _ = _sym_build(pts, 1, 0.0, eps, eta, leaf, "L", one_triangle=False)
    L1, L2 = np.asarray(H1.leaves()), np.asarray(H2.leaves())
    full = {tuple(l[:4]): i for i, l in enumerate(L2)}
    assert len(L1) < len(L2)
    rng = np.random.RandomState(0)
    for i in rng.choice(len(L1), 40, replace=False):
        j = full[tuple(L1[i, :4])]
        assert L1[i, 4] == L2[j, 4]
        A1, B1 = H1.leaf_panels(int(i))
        A2, B2 = H2.leaf_panels(int(j))
        assert np.array_equal(np.asarray(A1), np.asarray(A2))
        if L1[i, 4] > 0:
            assert np.array_equal(np.asarray(B1), np.asarray(B2))
    x = np.random.rand(n)
    y1, y2 = H1 * x, H2 * x
    assert np.linalg.norm(y1 - y2) / np.linalg.norm(y2) < 1e-13
    # roughly half the bytes
    s1, s2 = H1.stats(), H2.stats()
    e1, e2 = s1["dense_elements"] + s1["low_rank_elements"], s2["dense_elements"] + s2["low_rank_elements"]
    assert 0.45 < e1 / e2 < 0.62
    assert s1["hbm_bytes"] < 0.7 * s2["hbm_bytes"]
    D = H1.to_dense_in_user_numbering()
    assert np.abs(D - D.T).max() <= 1e-13 * np.abs(D).max()
    K = O.kernel_block(1, pts, pts, 0.0)
    assert np.linalg.norm(D - K) / np.linalg.norm(K) < eps


def test_one_triangle_complex_symmetric(built, oracle):
    O = oracle
    n, leaf, eps, eta, kappa = 4000, 40, 1e-4, 10.0, 5.0
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    H, cl, _ = _sym_build(pts, 2, kappa, eps, eta, leaf, "L", complex_=True)
    x = np.random.rand(n) + 1j * np.random.rand(n)
    y = H * x
    y_exact = O.dense_matvec(2, pts, pts, x, kappa)
    assert np.linalg.norm(y - y_exact) / np.linalg.norm(y_exact) < eps
    oc = O.Cluster(pts, max_leaf=leaf)
    OH = O.HMatrix(oc, oc, 2, kappa, is_complex=True, eps=eps, eta=eta, symmetry="S", uplo="L")
    assert {tuple(l[:4]) for l in np.asarray(H.leaves())} == {tuple(l[:4]) for l in OH.leaves}
    assert np.linalg.norm(y - OH.matvec(x)) / np.linalg.norm(y) < 2 * eps
    X = np.asfortranarray(np.random.rand(n, 6) + 1j * np.random.rand(n, 6))
    Y = H @ X
    for j in range(6):
        yj = H * np.ascontiguousarray(X[:, j])
        assert np.linalg.norm(Y[:, j] - yj) / np.linalg.norm(yj) < 1e-13


def test_one_triangle_callback_generator_copy_and_recompression(built, oracle):
    """Host-ACA build (Python generator), deep copy and SVD recompression keep the one-triangle tables consistent."""
    import Htool

    O = oracle
    n, leaf, eps, eta = 2500, 20, 1e-6, 10.0
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    H, cl, gen = _sym_build(pts, 0, 0.1, eps, eta, leaf, "L", native=False)
    x = np.random.rand(n)
    y_exact = O.dense_matvec(0, pts, pts, x, 0.1)
    y = H * x
    assert np.linalg.norm(y - y_exact) / np.linalg.norm(y_exact) < eps
    H2 = copy.deepcopy(H)
    assert np.array_equal(H2 * x, y)
    del H
    assert np.array_equal(H2 * x, y)
    Htool.recompression(H2, 1e-3)
    y3 = H2 * x
    assert 1e-9 < np.linalg.norm(y3 - y_exact) / np.linalg.norm(y_exact) < 5e-3


def test_one_triangle_falls_back_when_not_eligible(built, oracle):
    """Non-symmetric builds ignore the flag."""
    import Htool
    from tests.helpers import cluster_of

    O = oracle
    np.random.seed(0)
    pts = O.points_in_sphere(1500)
    cl = cluster_of(pts, 20)
    b = Htool.HMatrixTreeBuilder(1e-3, 10.0, "N", "N")
    b.set_symmetric_storage(True)
    H = b.build(Htool.NativeGenerator("laplace", pts, pts, 0.0), cl, cl)
    x = np.random.rand(1500)
    y_exact = O.dense_matvec(1, pts, pts, x, 0.0)
    assert np.linalg.norm(H * x - y_exact) / np.linalg.norm(y_exact) < 1e-3


def test_one_triangle_hermitian(built, oracle):
    """'H': the second use of a stored leaf is its conjugate transpose.  Hermitian test kernel
    exp(i (theta_i - theta_j)) / (0.1 + |x_i - x_j|) through a Python generator (host ACA)."""
    import Htool
    from tests.helpers import cluster_of

    O = oracle
    n, leaf, eps, eta = 2000, 25, 1e-5, 10.0
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    theta = 3.0 * pts[0]

    class HermitianGenerator(Htool.ComplexVirtualGenerator):
        def build_submatrix(self, J, K, mat):
            mat[:, :] = np.exp(1j * (theta[J][:, None] - theta[K][None, :])) * O.kernel_block(0, pts[:, J], pts[:, K], 0.1)

    A = np.exp(1j * (theta[:, None] - theta[None, :])) * O.kernel_block(0, pts, pts, 0.1)
    assert np.abs(A - A.conj().T).max() == 0
    cl = cluster_of(pts, leaf)
    x = np.random.rand(n) + 1j * np.random.rand(n)
    results = {}
    for uplo in ("L", "U"):
        b = Htool.ComplexHMatrixTreeBuilder(eps, eta, "H", uplo)
        b.set_symmetric_storage(True)
        gen = HermitianGenerator()
        H = b.build(gen, cl, cl)
        L = np.asarray(H.leaves())
        assert all((l[2] < l[0] + l[1]) if uplo == "L" else (l[0] < l[2] + l[3]) for l in L)
        y = H * x
        assert np.linalg.norm(y - A @ x) / np.linalg.norm(A @ x) < eps
        X = np.asfortranarray(np.random.rand(n, 5) + 1j * np.random.rand(n, 5))
        assert np.linalg.norm(H @ X - A @ X) / np.linalg.norm(A @ X) < eps
        D = H.to_dense_in_user_numbering()
        assert np.linalg.norm(D - A) / np.linalg.norm(A) < eps
        assert np.abs(D - D.conj().T).max() < 1e-12 * np.abs(D).max()
        results[uplo] = y
    assert np.linalg.norm(results["L"] - results["U"]) / np.linalg.norm(results["L"]) < 2 * eps


@pytest.mark.parametrize("complex_", [False, True])
def test_one_triangle_native_build_recompression(built, oracle, complex_):
    """Device build (dense leaves and low-rank leaves live in different batches) followed by SVD recompression: the dense
    batch is left alone and must keep its slots of the coefficient workspace."""
    import Htool

    O = oracle
    n, leaf, eps, eta = 6000, 30, 1e-6, 10.0
    kind, p0 = (2, 4.0) if complex_ else (1, 0.0)
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    H, cl, _ = _sym_build(pts, kind, p0, eps, eta, leaf, "L", complex_=complex_)
    x = np.random.rand(n) + (1j * np.random.rand(n) if complex_ else 0)
    y_exact = O.dense_matvec(kind, pts, pts, x, p0)
    assert np.linalg.norm(H * x - y_exact) / np.linalg.norm(y_exact) < eps
    before = H.stats()["low_rank_elements"]
    Htool.recompression(H, 1e-3)
    assert H.stats()["low_rank_elements"] < before
    e = np.linalg.norm(H * x - y_exact) / np.linalg.norm(y_exact)
    assert 3e-6 < e < 5e-3   # the explicit tolerance (1e-3) is honoured: coarser than the build's 1e-6
    H2 = copy.deepcopy(H)
    assert np.array_equal(H2 * x, H * x)


def test_one_triangle_with_separately_built_identical_trees(built, oracle):
    """The reference's example builds the target and the source cluster separately from the same points
    (example/use_hmatrix.py:27-28) and asks for 'S','L': two structurally identical trees count as one."""
    import Htool
    from tests.helpers import cluster_of

    O = oracle
    n = 2500
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    tcl, scl = cluster_of(pts, 50), cluster_of(pts, 50)
    H = Htool.HMatrixTreeBuilder(1e-4, 10.0, "S", "L").build(Htool.NativeGenerator("inv_delta", pts, pts, 0.1), tcl, scl)
    assert H.is_one_triangle()
    x = np.random.rand(n)
    y_exact = O.dense_matvec(0, pts, pts, x, 0.1)
    assert np.linalg.norm(H * x - y_exact) / np.linalg.norm(y_exact) < 1e-4
    other = cluster_of(pts, 30)   # a different tree on the same points (one more level): not symmetric storage
    H2 = Htool.HMatrixTreeBuilder(1e-4, 10.0, "S", "L").build(Htool.NativeGenerator("inv_delta", pts, pts, 0.1), tcl, other)
    assert not H2.is_one_triangle()
    assert np.linalg.norm(H2 * x - y_exact) / np.linalg.norm(y_exact) < 1e-4


@pytest.mark.parametrize("case", ["leaf10_L", "leaf48_U", "leaf100_L", "leaf300_U", "multi_batch", "complex_S", "hermitian", "copy"])
def test_one_triangle_sixteen_wide_sweep(built, oracle, case, monkeypatch):
    """Sixteen right-hand sides per fused sweep on the matrix cores (the panels transposed by MFMAs with selection matrices
    for their second use): column by column against the single-vector products (<= 1e-13 relative), the sweeps of four on
    the vector units and the exact dense operator; row tiles of <= 32, <= 64, <= 128 rows and clusters cut into several
    tiles, several pack batches (segments per tile), complex symmetric and Hermitian storage, deep copies."""
    import Htool
    from tests.helpers import cluster_of

    O = oracle
    np.random.seed(5)
    if case == "hermitian":
        n, leaf, eps = 2000, 25, 1e-5
        pts = O.points_in_sphere(n)
        theta = 3.0 * pts[0]

        class HermitianGenerator(Htool.ComplexVirtualGenerator):
            def build_submatrix(self, J, K, mat):
                mat[:, :] = np.exp(1j * (theta[J][:, None] - theta[K][None, :])) * O.kernel_block(0, pts[:, J], pts[:, K], 0.1)

        A = np.exp(1j * (theta[:, None] - theta[None, :])) * O.kernel_block(0, pts, pts, 0.1)
        b = Htool.ComplexHMatrixTreeBuilder(eps, 10.0, "H", "U")
        b.set_symmetric_storage(True)
        cl = cluster_of(pts, leaf)
        gen = HermitianGenerator()
        H = b.build(gen, cl, cl)
        exact = lambda X: A @ X
        cplx = True
    elif case == "complex_S":
        n, leaf, eps = 4000, 40, 1e-4
        pts = O.points_in_sphere(n)
        H, cl, _ = _sym_build(pts, 2, 5.0, eps, 10.0, leaf, "L", complex_=True)
        exact = lambda X: O.dense_matvec(2, pts, pts, X, 5.0)
        cplx = True
    else:
        leaf = {"leaf10_L": 10, "leaf48_U": 48, "leaf300_U": 300}.get(case, 100)
        n = {10: 3000, 48: 4000, 300: 7001}.get(leaf, 9000)
        eps = 1e-5
        if case == "multi_batch":
            n = 20000
            monkeypatch.setenv("HTOOL_BUILD_ARENA_MB", "40")
        pts = O.points_in_sphere(n)
        H, cl, _ = _sym_build(pts, 1, 0.0, eps, 10.0, leaf, case[-1] if case[-1] in "LU" else "L")
        exact = lambda X: O.dense_matvec(1, pts, pts, X, 0.0)
        cplx = False
    assert H.is_one_triangle()
    if case == "copy":
        H0 = H
        X0 = np.asfortranarray(np.random.rand(n, 16))
        Y0 = H0 @ X0
        H = copy.deepcopy(H0)
        del H0
        assert np.array_equal(H @ X0, Y0)
    for mu in (9, 16, 37):
        X = np.random.rand(n, mu)
        if cplx:
            X = X + 1j * np.random.rand(n, mu)
        X = np.asfortranarray(X)
        Y = H @ X
        assert np.array_equal(H @ X, Y)  # fixed summation order
        for c in range(mu):
            yc = H * np.ascontiguousarray(X[:, c])
            assert np.linalg.norm(Y[:, c] - yc) <= 1e-13 * np.linalg.norm(yc), (case, mu, c)
        Ye = exact(X)
        assert np.linalg.norm(Y - Ye) / np.linalg.norm(Ye) < eps
    # the same columns through the sweeps of four (vector units): equal to rounding, not bitwise (other order of the sums)
    monkeypatch.setenv("HTOOL_MULTI_RHS_KERNEL", "valu")
    H2 = copy.deepcopy(H)
    Yv = H2 @ X
    assert np.linalg.norm(Yv - Y) <= 1e-13 * np.linalg.norm(Y)
    assert not np.array_equal(Yv, Y)
