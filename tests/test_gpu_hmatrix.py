"""GPU parity tests of the H-matrix product (hot-path rows a9-a12 of SURVEY.md section 8).

Mirrors the assertions of the reference's tests/test_hmatrix.py:67,83-85,94-96 (same geometry seed,
sizes, eta, epsilon, leaf size; our own harness) and adds the oracle checks: HIP product vs CPU leaf
loop on identical panels (<= 1e-12), vs the exact dense kernel (< epsilon).
"""
import copy
import logging

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("symmetry", ["N", "S"])
@pytest.mark.parametrize("custom_svd", [True, False])
def test_hmatrix_reference_case(built, oracle, symmetry, custom_svd):
    import Htool
    from tests.helpers import CustomSVD, NumpyGenerator, cluster_of

    logging.basicConfig(level=logging.INFO)
    O = oracle
    nb_rows = nb_cols = 500
    target_points, source_points = O.random_geometries(3, nb_rows, nb_cols)  # seed 0, source shifted by +2
    eta, epsilon = 100, 1e-3
    target_cluster = cluster_of(target_points, 10)
    if symmetry == "N":
        source_cluster = cluster_of(source_points, 10)
        generator = NumpyGenerator(target_points, source_points)
    else:
        source_cluster = target_cluster
        generator = NumpyGenerator(target_points, target_points)
    builder = Htool.HMatrixTreeBuilder(epsilon, eta, "N", "N")
    lr = None
    if custom_svd:
        lr = CustomSVD(generator, False)
        builder.set_low_rank_generator(lr)
    hmatrix = builder.build(generator, target_cluster, source_cluster)
    assert hmatrix.shape == (nb_rows, nb_cols)
    copy_hmatrix = copy.deepcopy(hmatrix)
    _ = hmatrix.to_dense()
    dense_user = hmatrix.to_dense_in_user_numbering()

    np.random.seed(0)
    x = np.random.rand(nb_cols)
    y = hmatrix * x
    y_exact = generator.mat_vec(x)
    y_dense = dense_user.dot(x)
    y_copy = copy_hmatrix * x
    assert np.linalg.norm(y - y_exact) / np.linalg.norm(y_exact) < epsilon
    assert np.linalg.norm(y - y_dense) / np.linalg.norm(y_dense) < 1e-10
    assert np.linalg.norm(y - y_copy) < 1e-10

    np.random.seed(0)
    X = np.random.rand(nb_cols, 2)
    Y = hmatrix @ X
    Y_exact = generator.mat_mat(X)
    assert np.linalg.norm(Y - Y_exact) / np.linalg.norm(Y_exact) < epsilon
    assert np.linalg.norm(Y - dense_user @ X) / np.linalg.norm(dense_user @ X) < 1e-10
    assert np.linalg.norm(Y - copy_hmatrix @ X) < 1e-10

    if symmetry != "N":
        # H-LU / H-Cholesky solves (tests/test_hmatrix.py:98-128): the hierarchical factorisation on the device (round 4)
        copy_hmatrix.lu_factorization()
        assert copy_hmatrix.factorization_info()["kind"] == "hierarchical"
        x_ref = np.ones(nb_cols)
        x_lu = copy_hmatrix.lu_solve("N", hmatrix * x_ref)
        assert np.linalg.norm(x_lu - x_ref) / np.linalg.norm(x_ref) < epsilon
        x_ref2 = np.ones((nb_cols, 2))
        x_lu2 = copy_hmatrix.lu_solve("N", hmatrix @ x_ref2)
        assert np.linalg.norm(x_lu2 - x_ref2) / np.linalg.norm(x_ref2) < epsilon
        copy_hmatrix = copy.deepcopy(hmatrix)
        copy_hmatrix.cholesky_factorization("L")
        x_ch = copy_hmatrix.cholesky_solve("L", hmatrix * x_ref)
        assert np.linalg.norm(x_ch - x_ref) / np.linalg.norm(x_ref) < epsilon
        x_ch2 = copy_hmatrix.cholesky_solve("L", hmatrix @ x_ref2)
        assert np.linalg.norm(x_ch2 - x_ref2) / np.linalg.norm(x_ref2) < epsilon

    # densified in cluster numbering == permuted user-numbered one
    pt, ps = np.asarray(target_cluster.get_permutation()), np.asarray(source_cluster.get_permutation())
    assert np.allclose(hmatrix.to_dense(), dense_user[np.ix_(pt, ps)], rtol=0, atol=1e-13)

    print(hmatrix.get_tree_parameters())
    print(hmatrix.get_local_information())
    if lr is not None:
        lr.clear_data()

    with pytest.raises(RuntimeError, match="Wrong size for HMatrix-vector product"):
        hmatrix * np.zeros(nb_cols + 1)
    with pytest.raises(RuntimeError, match="Wrong dimension for HMatrix-vector product"):
        hmatrix * np.zeros((nb_cols, 1))


@pytest.mark.parametrize("n,leaf,eta,epsilon", [(3000, 10, 10.0, 1e-3), (6000, 64, 10.0, 1e-4), (5000, 100, 3.0, 1e-6)])
def test_product_vs_cpu_leaf_loop_and_dense(built, oracle, n, leaf, eta, epsilon):
    """HIP product vs (i) the CPU leaf loop on the SAME panels, (ii) the exact dense operator."""
    import Htool
    from tests.helpers import NumpyGenerator, cluster_of, cpu_leaf_loop

    O = oracle
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    cl = cluster_of(pts, leaf)
    gen = NumpyGenerator(pts, pts, O.K_LAPLACE, 0.0)
    H = Htool.HMatrixTreeBuilder(epsilon, eta, "N", "N").build(gen, cl, cl)
    np.random.seed(0)
    x = np.random.rand(n)
    y = H * x
    y_cpu = cpu_leaf_loop(H, x)
    assert np.linalg.norm(y - y_cpu) / np.linalg.norm(y_cpu) < 1e-12
    y_exact = gen.mat_vec(x)
    assert np.linalg.norm(y - y_exact) / np.linalg.norm(y_exact) < epsilon
    # bitwise reproducible (fixed summation order, no atomics)
    assert np.array_equal(y, H * x)
    # linearity (size-independent property)
    z = np.random.rand(n)
    assert np.linalg.norm(H * (2.0 * x + z) - (2.0 * y + H * z)) / np.linalg.norm(y) < 1e-12
    # leaves tile the matrix exactly once
    L = np.asarray(H.leaves()).astype(np.int64)
    assert (L[:, 1] * L[:, 3]).sum() == n * n


def test_rectangular_and_host_aca_ranks_match_oracle(built, oracle):
    """400 x 200 operator (tests/test_distributed_operator.py:25 shape); host-driven ACA (callback generator): every
    sampled low-rank leaf against the independent numpy formulation (explicit-residual ACA, SVD epsilon-rank, Frobenius
    error of U V against the exact block -- oracle/independent.py), then leaf-by-leaf rank equality with the C++ oracle."""
    import Htool
    from tests.helpers import NumpyGenerator, cluster_of, independent_leaf_checks

    O = oracle
    np.random.seed(0)
    T, S = np.random.random((3, 400)), np.random.random((3, 200))
    tcl, scl = cluster_of(T, 10), cluster_of(S, 10)
    gen = NumpyGenerator(T, S)
    for epsilon in (1e-3, 1e-6):
        H = Htool.HMatrixTreeBuilder(epsilon, 10.0, "N", "N").build(gen, tcl, scl)
        assert H.shape == (400, 200)
        x = np.random.rand(200)
        y = H * x
        ye = gen.mat_vec(x)
        assert np.linalg.norm(y - ye) / np.linalg.norm(ye) < epsilon
        independent_leaf_checks(H, T, S, O.K_INV_DELTA, 0.1, epsilon, n_sample=200, transpose_rule=True, min_leaves=20 if epsilon == 1e-3 else 0)
        otc, osc = O.Cluster(T, max_leaf=10), O.Cluster(S, max_leaf=10)
        assert np.array_equal(otc.perm, np.asarray(tcl.get_permutation()))
        assert np.array_equal(osc.perm, np.asarray(scl.get_permutation()))
        OH = O.HMatrix(otc, osc, O.K_INV_DELTA, 0.1, eps=epsilon, eta=10.0)
        mine = {tuple(l[:4]): l[4] for l in np.asarray(H.leaves())}
        theirs = {tuple(l[:4]): l[4] for l in OH.leaves}
        assert mine == theirs
        assert np.linalg.norm(y - OH.matvec(x)) / np.linalg.norm(ye) < 1e-12


def test_complex_product(built, oracle):
    import Htool
    from tests.helpers import ComplexNumpyGenerator, cluster_of, cpu_leaf_loop

    O = oracle
    n = 3000
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    cl = cluster_of(pts, 32)
    gen = ComplexNumpyGenerator(pts, pts, 5.0)
    H = Htool.ComplexHMatrixTreeBuilder(1e-4, 10.0, "N", "N").build(gen, cl, cl)
    x = np.random.rand(n) + 1j * np.random.rand(n)
    y = H * x
    y_cpu = cpu_leaf_loop(H, x, True)
    assert np.linalg.norm(y - y_cpu) / np.linalg.norm(y_cpu) < 1e-12
    ye = gen.mat_vec(x)
    assert np.linalg.norm(y - ye) / np.linalg.norm(ye) < 1e-4


def test_custom_dense_blocks_generator(built, oracle):
    import Htool
    from tests.helpers import CustomDenseBlocksGenerator, CustomSVD, NumpyGenerator, cluster_of

    np.random.seed(0)
    T, S = np.random.random((2, 400)), np.random.random((2, 400))
    tcl, scl = cluster_of(T, 10), cluster_of(S, 10)
    gen = NumpyGenerator(T, S)
    b = Htool.HMatrixTreeBuilder(1e-6, 10.0, "N", "N")
    b.set_dense_blocks_generator(CustomDenseBlocksGenerator(gen, tcl, scl))
    b.set_low_rank_generator(CustomSVD(gen))
    H = b.build(gen, tcl, scl)
    x = np.random.rand(400)
    ye = gen.mat_vec(x)
    assert np.linalg.norm(H * x - ye) / np.linalg.norm(ye) < 1e-6


@pytest.mark.parametrize("complex_", [False, True])
def test_multi_rhs_sweep_equals_column_products(built, oracle, complex_):
    """H @ X (src/htool/hmatrix/hmatrix.hpp:119-138) multiplies up to 8 right-hand sides per VALU sweep of the panels --
    every column then equals the single-vector product bit for bit (same summation order) -- and, with more than 8 columns,
    16 per sweep on the matrix cores (v_mfma_f64_16x16x4_f64, real and complex: other summation order, <= 1e-13)."""
    import Htool
    from tests.helpers import cluster_of

    O = oracle
    n = 5000
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    cl = cluster_of(pts, 40)
    if complex_:
        H = Htool.ComplexHMatrixTreeBuilder(1e-4, 10.0, "N", "N").build(Htool.ComplexNativeGenerator("helmholtz", pts, pts, 6.0), cl, cl)
    else:
        H = Htool.HMatrixTreeBuilder(1e-4, 10.0, "N", "N").build(Htool.NativeGenerator("laplace", pts, pts), cl, cl)
    for mu in (1, 2, 3, 5, 8, 11, 16, 21):
        X = np.random.rand(n, mu) + (1j * np.random.rand(n, mu) if complex_ else 0)
        X = np.asfortranarray(X)
        Y = H @ X
        assert Y.shape == (n, mu) and Y.flags.f_contiguous
        mfma = mu > 8
        for c in range(mu):
            yc = H * np.ascontiguousarray(X[:, c])
            if mfma and c < (mu // 16) * 16 + (16 if mu % 16 > 8 else 0):
                assert np.linalg.norm(Y[:, c] - yc) <= 1e-13 * np.linalg.norm(yc)
            else:
                assert np.array_equal(Y[:, c], yc)
        if mfma:
            assert np.array_equal(Y, H @ X)  # bitwise repeatable
            # a column's result does not depend on its neighbours or on their number
            X2 = np.asfortranarray(np.random.rand(n, 12) + (1j * np.random.rand(n, 12) if complex_ else 0))
            X2[:, 3] = X[:, 3]
            assert np.array_equal((H @ X2)[:, 3], Y[:, 3])
    Ye = O.dense_matvec(O.K_HELMHOLTZ if complex_ else O.K_LAPLACE, pts, pts, X, 6.0 if complex_ else 0.0)
    assert np.linalg.norm(Y - Ye) / np.linalg.norm(Ye) < 1e-4
    with pytest.raises(RuntimeError, match="Wrong size for HMatrix-matrix product"):
        H @ np.zeros((n + 1, 2), order="F")


@pytest.mark.parametrize("case", ["leaf10", "leaf48", "leaf100", "rect", "partition", "copy", "recompressed"])
def test_sixteen_wide_mfma_sweep(built, oracle, case):
    """The 16-wide sweep on the matrix cores against the single-vector products (<= 1e-13 relative, column by column) and the
    exact dense operator, over the shapes its row-group / column-share logic distinguishes: row tiles of <= 32, <= 64 and
    <= 128 rows, rectangular operators, operators built on one partition (cluster-numbered device path), deep copies and
    recompressed operators (the 16-wide workspace is rebuilt)."""
    import copy

    import torch

    import Htool
    from tests.helpers import cluster_of

    O = oracle
    np.random.seed(3)
    leaf = {"leaf10": 10, "leaf48": 48}.get(case, 100)
    n = 6000 if leaf == 100 else 3000
    T = O.points_in_sphere(n)
    S = O.points_in_sphere(2500) + np.array([[0.4], [0.0], [0.0]]) if case == "rect" else T
    world = 3 if case == "partition" else 1
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(leaf)
    tcl = b.create_cluster_tree(T, 2, size_of_partition=world)
    scl = tcl if S is T else cluster_of(S, leaf)
    gen = Htool.NativeGenerator("laplace", T, S)
    H = Htool.HMatrixTreeBuilder(1e-5, 10.0, "N", "N").build(gen, tcl, scl, 1 if case == "partition" else -1)
    if case == "copy":
        H0 = H
        X0 = np.asfortranarray(np.random.rand(S.shape[1], 16))
        Y0 = H0 @ X0
        H = copy.deepcopy(H0)
        del H0
        assert np.array_equal(H @ X0, Y0)
    if case == "recompressed":
        X0 = np.asfortranarray(np.random.rand(S.shape[1], 16))
        Y0 = H @ X0
        assert Htool.recompression(H) > 0
        Y1 = H @ X0
        assert np.linalg.norm(Y1 - Y0) / np.linalg.norm(Y0) < 1e-4 and not np.array_equal(Y1, Y0)
    ns = S.shape[1]
    for mu in (9, 16, 35):
        X = np.asfortranarray(np.random.rand(ns, mu))
        Y = H @ X
        for c in range(mu):
            yc = H * np.ascontiguousarray(X[:, c])
            assert np.linalg.norm(Y[:, c] - yc) <= 1e-13 * np.linalg.norm(yc), (case, mu, c)
        if case != "partition":
            Ye = O.dense_matvec(O.K_LAPLACE, T, S, X)
            assert np.linalg.norm(Y - Ye) / np.linalg.norm(Ye) < 1e-5
    if case == "partition":  # the multi-GPU path: cluster-numbered x in, the local row slice out, on device buffers
        sub = tcl.get_cluster_on_partition(1)
        perm = np.asarray(tcl.get_permutation())
        Xd = torch.from_numpy(np.ascontiguousarray(X[perm].T)).cuda()  # mu x n, cluster numbering
        Yd = torch.zeros(mu, sub.get_size(), dtype=torch.float64, device="cuda")
        H.matmat_device(Xd.data_ptr(), ns, Yd.data_ptr(), sub.get_size(), mu, 1, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(Yd.cpu().numpy().T, Y)
        rows = perm[sub.get_offset():sub.get_offset() + sub.get_size()]
        Ye = O.dense_matvec(O.K_LAPLACE, T, S, X, rows=rows)
        assert np.linalg.norm(Y - Ye) / np.linalg.norm(Ye) < 1e-5


@pytest.mark.parametrize("case", ["leaf10", "leaf40", "leaf100", "rect", "partition"])
def test_sixteen_wide_mfma_sweep_complex(built, oracle, case, monkeypatch):
    """The same for complex128 operators (Helmholtz): a 16-byte load is one (re, im) row, four MFMAs per load into a real and
    an imaginary accumulator tile.  Row tiles of <= 16, <= 32 and <= 64 rows, rectangular, one partition; against the
    single-vector products (1e-13), the 8-wide VALU sweeps and the exact dense operator."""
    import torch

    import Htool
    from tests.helpers import cluster_of

    O = oracle
    np.random.seed(5)
    rng = np.random.RandomState(5)
    leaf = {"leaf10": 10, "leaf40": 40}.get(case, 100)
    n = 5000 if leaf == 100 else 2500
    T = O.points_in_sphere(n)
    S = O.points_in_sphere(2100) + np.array([[0.4], [0.0], [0.0]]) if case == "rect" else T
    world = 3 if case == "partition" else 1
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(leaf)
    tcl = b.create_cluster_tree(T, 2, size_of_partition=world)
    scl = tcl if S is T else cluster_of(S, leaf)
    gen = Htool.ComplexNativeGenerator("helmholtz", T, S, 6.0)
    H = Htool.ComplexHMatrixTreeBuilder(1e-5, 10.0, "N", "N").build(gen, tcl, scl, 1 if case == "partition" else -1)
    ns = S.shape[1]
    for mu in (9, 16, 21):
        X = np.asfortranarray(rng.random_sample((ns, mu)) + 1j * rng.random_sample((ns, mu)))
        Y = np.asarray(H @ X)
        assert np.array_equal(Y, np.asarray(H @ X))  # fixed summation order
        for c in range(mu):
            yc = H * np.ascontiguousarray(X[:, c])
            assert np.linalg.norm(Y[:, c] - yc) <= 1e-13 * np.linalg.norm(yc), (case, mu, c)
        if case != "partition":
            Ye = O.dense_matvec(O.K_HELMHOLTZ, T, S, X, 6.0)
            assert np.linalg.norm(Y - Ye) / np.linalg.norm(Ye) < 1e-5
    # the VALU sweeps (8 columns at a time) give the same columns
    monkeypatch.setenv("HTOOL_MULTI_RHS_KERNEL", "valu")
    H2 = Htool.ComplexHMatrixTreeBuilder(1e-5, 10.0, "N", "N").build(gen, tcl, scl, 1 if case == "partition" else -1)
    Yv = np.asarray(H2 @ X)
    assert np.linalg.norm(Yv - Y) <= 1e-13 * np.linalg.norm(Y)
    monkeypatch.delenv("HTOOL_MULTI_RHS_KERNEL")
    if case == "partition":  # the multi-GPU path: cluster-numbered x in, the local row slice out, on device buffers
        sub = tcl.get_cluster_on_partition(1)
        perm = np.asarray(tcl.get_permutation())
        Xd = torch.from_numpy(np.ascontiguousarray(X[perm].T)).cuda()  # mu x n, cluster numbering
        Yd = torch.zeros(mu, sub.get_size(), dtype=torch.complex128, device="cuda")
        H.matmat_device(Xd.data_ptr(), ns, Yd.data_ptr(), sub.get_size(), mu, 1, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(Yd.cpu().numpy().T, Y)
        rows = perm[sub.get_offset():sub.get_offset() + sub.get_size()]
        Ye = O.dense_matvec(O.K_HELMHOLTZ, T, S, X, 6.0, rows=rows)
        assert np.linalg.norm(Y - Ye) / np.linalg.norm(Ye) < 1e-5


@pytest.mark.parametrize("complex_", [False, True])
def test_gmres_on_device(built, oracle, complex_):
    """GMRES with the HIP product as operator: converges to the solution of the dense system
    (tests/test_ddm_solver.py:659-660 bar: residual < tol, |x - x_ref| / |x_ref| < 10 eps)."""
    import torch

    import Htool
    from htool_python_amd.krylov import gmres
    from htool_python_amd.solver import DeviceOperator
    from tests.helpers import cluster_of

    O = oracle
    n, eps = 3000, 1e-8
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    cl = cluster_of(pts, 32)
    if complex_:
        kind, p0 = O.K_HELMHOLTZ, 3.0
        H = Htool.ComplexHMatrixTreeBuilder(eps, 10.0, "N", "N").build(Htool.ComplexNativeGenerator("helmholtz", pts, pts, p0), cl, cl)
    else:
        kind, p0 = O.K_INV_DELTA, 0.1
        H = Htool.HMatrixTreeBuilder(eps, 10.0, "N", "N").build(Htool.NativeGenerator("inv_delta", pts, pts, p0), cl, cl)
    shift = 50.0  # second-kind-like, well conditioned: (shift I + A) x = b
    A = O.kernel_block(kind, pts, pts, p0) + shift * np.eye(n)
    x_ref = np.random.rand(n) + (1j * np.random.rand(n) if complex_ else 0)
    b = A @ x_ref
    perm = np.asarray(cl.get_permutation())
    op = DeviceOperator(H, shift=shift)
    xl, info = gmres(op.apply, torch.from_numpy(b[perm]).cuda(), tol=1e-10, restart=30, max_it=300)
    x = np.empty_like(b)
    x[perm] = xl.cpu().numpy()
    assert info["converged"] and info["residuals"][-1] <= 1e-10
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-7
    assert np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref) < 1e-6
    # residual history is non-increasing within a restart cycle
    r = np.array(info["residuals"][:30])
    assert np.all(r[1:] <= r[:-1] * (1 + 1e-12))


@pytest.mark.parametrize("n_t,n_s,dim,children,leaf,eta", [
    (1, 1, 3, 2, 10, 10.0),        # a single entry
    (7, 5, 2, 2, 10, 10.0),        # smaller than a leaf: one dense block
    (37, 211, 3, 2, 3, 1.0),       # odd sizes, tiny leaves (tiles of 3..5 rows)
    (513, 129, 1, 2, 1, 10.0),     # 1-D points, leaves of a single point
    (600, 600, 2, 3, 7, 10.0),     # ternary tree
    (1000, 333, 3, 9, 5, 0.5),     # 9 children, strict admissibility
    (300, 2500, 3, 2, 200, 10.0),  # leaves larger than a tile (cut in pieces)
])
def test_edge_shapes(built, oracle, n_t, n_s, dim, children, leaf, eta):
    """Ragged / degenerate inputs: the product must still equal the exact dense operator to epsilon and the
    CPU leaf loop on the same panels to rounding, for both build paths."""
    import Htool
    from tests.helpers import NumpyGenerator, cpu_leaf_loop

    O = oracle
    rng = np.random.RandomState(n_t + 7 * n_s)
    T, S = rng.rand(dim, n_t), rng.rand(dim, n_s) + 0.25
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(leaf)
    tcl, scl = b.create_cluster_tree(T, children), b.create_cluster_tree(S, children)
    eps = 1e-6
    x = rng.rand(n_s)
    ye = O.dense_matvec(O.K_INV_DELTA, T, S, x, 0.1)
    for native in (True, False):
        gen = Htool.NativeGenerator("inv_delta", T, S, 0.1) if native else NumpyGenerator(T, S)
        H = Htool.HMatrixTreeBuilder(eps, eta, "N", "N").build(gen, tcl, scl)
        assert H.shape == (n_t, n_s)
        y = H * x
        assert y.shape == (n_t,)
        assert np.linalg.norm(y - ye) <= eps * np.linalg.norm(ye) + 1e-300
        yc = cpu_leaf_loop(H, x)
        assert np.linalg.norm(y - yc) <= 1e-12 * np.linalg.norm(yc) + 1e-300
        L = np.asarray(H.leaves()).astype(np.int64)
        assert (L[:, 1] * L[:, 3]).sum() == n_t * n_s
        X = np.asfortranarray(rng.rand(n_s, 3))
        Y = H @ X
        for c in range(3):
            assert np.array_equal(Y[:, c], H * np.ascontiguousarray(X[:, c]))
        D = H.to_dense_in_user_numbering()
        assert np.linalg.norm(D @ x - y) <= 1e-10 * np.linalg.norm(y) + 1e-300


def test_local_blocks_partition_by_partition(built, oracle):
    """Sub-operators restricted to (target partition p) x (source partition q) -- what block_diagonal_hmatrix and
    DefaultLocalApproximationBuilder are made of (src/htool/distributed_operator/utility.hpp:31,34-41): summing the
    P x P local products reproduces the full product; each equals the exact dense sub-block to epsilon."""
    import Htool

    O = oracle
    n, P, eps = 3000, 3, 1e-5
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(20)
    cl = b.create_cluster_tree(pts, 2, size_of_partition=P)
    perm = np.asarray(cl.get_permutation())
    gen = Htool.NativeGenerator("laplace", pts, pts)
    builder = Htool.HMatrixTreeBuilder(eps, 10.0, "N", "N")
    x = np.random.rand(n)
    xp = x[perm]
    yp = np.zeros(n)
    for p in range(P):
        tp = cl.get_cluster_on_partition(p)
        for q in range(P):
            sq = cl.get_cluster_on_partition(q)
            H = builder.build_local(gen, cl, cl, p, q)
            assert H.shape == (tp.get_size(), sq.get_size())
            xs = xp[sq.get_offset():sq.get_offset() + sq.get_size()]
            ys = H * xs  # local slices in cluster order on both sides
            A = O.kernel_block(O.K_LAPLACE, pts[:, perm[tp.get_offset():tp.get_offset() + tp.get_size()]], pts[:, perm[sq.get_offset():sq.get_offset() + sq.get_size()]])
            assert np.linalg.norm(ys - A @ xs) <= eps * np.linalg.norm(A @ xs)
            D = H.to_dense()
            assert np.linalg.norm(D - A) <= 10 * eps * np.linalg.norm(A)
            yp[tp.get_offset():tp.get_offset() + tp.get_size()] += ys
    y = np.zeros(n)
    y[perm] = yp
    ye = O.dense_matvec(O.K_LAPLACE, pts, pts, x)
    assert np.linalg.norm(y - ye) / np.linalg.norm(ye) < eps


def test_degenerate_inputs_sweep(built):
    """243 builds on degenerate inputs (1, 2, 7 points; identical, duplicated, collinear points; 2-D; leaf size 1 and
    leaf size > N; 'N' / 'S','L' / 'S','U'; the three native kernels): every one must build, multiply without NaN and
    match the exact dense operator (tools/edge_sweep.py)."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "edge_sweep.py")
    spec = importlib.util.spec_from_file_location("edge_sweep", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(verbose=False) == []


@pytest.mark.parametrize("symmetry,uplo", [("N", "N"), ("S", "L")])
def test_block_jacobi_preconditioned_gmres(built, oracle, symmetry, uplo):
    """`solver.facto_one_level()` (example/use_ddm_solver.py:66-68): block-Jacobi from the dense diagonal leaves as a
    right preconditioner; the solution matches the dense solve and the iteration count does not grow."""
    import mpi4py
    import torch

    import Htool
    from htool_python_amd.krylov import gmres
    from htool_python_amd.solver import BlockJacobi, DeviceOperator
    from tests.helpers import cluster_of

    O = oracle
    n, eps = 4000, 1e-8
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    cl = cluster_of(pts, 40, size_of_partition=1)
    gen = Htool.NativeGenerator("inv_delta", pts, pts, 0.1)
    approx = Htool.DefaultApproximationBuilder(gen, cl, cl, Htool.HMatrixTreeBuilder(eps, 10.0, symmetry, uplo), mpi4py.MPI.COMM_WORLD)
    A = O.kernel_block(0, pts, pts, 0.1)
    x_ref = np.random.rand(n)
    b = A @ x_ref
    # raw Krylov loop, with and without the preconditioner
    perm = np.asarray(cl.get_permutation())
    op = DeviceOperator(approx.hmatrix)
    bl = torch.from_numpy(b[perm]).cuda()
    _, plain = gmres(op.apply, bl, tol=1e-9, restart=60, max_it=600)
    M = BlockJacobi(approx.hmatrix, 0, n)
    v = torch.rand(n, dtype=torch.float64, device="cuda")
    Dv = np.zeros(n)  # M v = blockdiag(A_cluster) v on the host
    Ac = A[np.ix_(perm, perm)]
    for l in np.asarray(approx.hmatrix.leaves()):
        if l[4] < 0 and l[0] == l[2]:
            s = slice(l[0], l[0] + l[1])
            Dv[s] = Ac[s, s] @ M(v).cpu().numpy()[s]
    assert np.linalg.norm(Dv - v.cpu().numpy()) / np.linalg.norm(v.cpu().numpy()) < 1e-10   # M^-1 inverts the diagonal blocks
    xl, prec = gmres(op.apply, bl, tol=1e-9, restart=60, max_it=600, precond=M)
    assert prec["converged"] and prec["iterations"] <= plain["iterations"]
    x = np.empty(n)
    x[perm] = xl.cpu().numpy()
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-6
    # the reference's call sequence
    solver = Htool.DDMSolverBuilder(approx.distributed_operator, approx.block_diagonal_hmatrix).solver
    solver.set_hpddm_args("-hpddm_compute_residual l2 -hpddm_tol 1e-9 -hpddm_max_it 600 -hpddm_gmres_restart 60")
    solver.facto_one_level()
    xs = np.zeros(n)
    solver.solve(xs, b)
    info = solver.get_information()
    # round 3: with block_diagonal_hmatrix at hand (one rank: the operator itself) facto_one_level() inverts that whole block -- a dense
    # device factorisation standing in for the reference's H-LU -- so the preconditioned system is solved at once
    assert "dense device LU" in info["Preconditioner"] and int(info["Nb_it"]) <= 3
    assert np.linalg.norm(A @ xs - b) / np.linalg.norm(b) < 1e-6
    # without the block: block-Jacobi on the dense diagonal leaves, the iteration count of the raw loop above
    solver = Htool.DDMSolverBuilder(approx.distributed_operator).solver
    solver.set_hpddm_args("-hpddm_tol 1e-9 -hpddm_max_it 600 -hpddm_gmres_restart 60")
    solver.facto_one_level()
    xs = np.zeros(n)
    solver.solve(xs, b)
    info = solver.get_information()
    assert "block-jacobi" in info["Preconditioner"] and int(info["Nb_it"]) == prec["iterations"]
    assert np.linalg.norm(A @ xs - b) / np.linalg.norm(b) < 1e-6


@pytest.mark.parametrize("case", ["native_sym", "callback_complex", "partition", "recompressed"])
def test_save_and_load_hmatrix(built, oracle, tmp_path, case):
    """Checkpoint / resume: save leaves + panels to .npz, rebuild on the same cluster trees without any
    generator call; same leaf table, same product to rounding, wrong trees are refused."""
    import Htool
    from tests.helpers import ComplexNumpyGenerator, cluster_of

    O = oracle
    n = 3000
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    part = -1
    if case == "native_sym":
        cl = cluster_of(pts, 25, size_of_partition=1)
        H = Htool.HMatrixTreeBuilder(1e-5, 10.0, "S", "L").build(Htool.NativeGenerator("inv_delta", pts, pts, 0.1), cl, cl)
        assert H.is_one_triangle()
    elif case == "callback_complex":
        cl = cluster_of(pts[:, :1200], 20, size_of_partition=1)
        gen = ComplexNumpyGenerator(pts[:, :1200], pts[:, :1200], 4.0)
        H = Htool.ComplexHMatrixTreeBuilder(1e-4, 10.0, "N", "N").build(gen, cl, cl)
    elif case == "partition":
        cl = cluster_of(pts, 30, size_of_partition=3)
        part = 1
        H = Htool.HMatrixTreeBuilder(1e-4, 10.0, "N", "N").build(Htool.NativeGenerator("laplace", pts, pts, 0.0), cl, cl, part)
    else:
        cl = cluster_of(pts, 16, size_of_partition=1)
        H = Htool.HMatrixTreeBuilder(1e-6, 10.0, "N", "N").build(Htool.NativeGenerator("laplace", pts, pts, 0.0), cl, cl)
        Htool.recompression(H, 1e-3)
    path = str(tmp_path / "operator.npz")
    Htool.save_hmatrix(path, H)
    H2 = Htool.load_hmatrix(path, cl, target_partition_number=part)
    assert type(H2) is type(H) and H2.shape == H.shape and H2.is_one_triangle() == H.is_one_triangle()
    L1 = {tuple(l) for l in np.asarray(H.leaves()).tolist()}
    L2 = {tuple(l) for l in np.asarray(H2.leaves()).tolist()}
    assert L1 == L2
    ncol = H.shape[1]
    x = np.random.rand(ncol) + (1j * np.random.rand(ncol) if case == "callback_complex" else 0)
    y1, y2 = H * x, H2 * x
    assert np.linalg.norm(y1 - y2) / np.linalg.norm(y1) < 1e-13
    assert H2.get_tree_parameters()["Epsilon"] == H.get_tree_parameters()["Epsilon"]
    X = np.asfortranarray(np.stack([x, 2 * x], axis=1))
    assert np.linalg.norm(H2 @ X - H @ X) / np.linalg.norm(H @ X) < 1e-13
    # a different tree is refused
    other = cluster_of(pts[:, ::-1].copy() if case != "callback_complex" else pts[:, 1200:2400], 25, size_of_partition=3 if case == "partition" else 1)
    with pytest.raises(RuntimeError):
        Htool.load_hmatrix(path, other, target_partition_number=part)


def test_randomised_stress_short(built):
    """25 s of tools/fuzz.py with a fixed seed: random geometry / leaf size / eta / eps / kernel / storage / partition /
    arena size / recompression cases against sampled exact rows (longer runs: `python tools/fuzz.py 600 <seed>`)."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz.py")
    spec = importlib.util.spec_from_file_location("fuzz", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        assert mod.main(25.0, 3) == 0
    finally:
        os.environ.pop("HTOOL_BUILD_ARENA_MB", None)


@pytest.mark.parametrize("case", ["binary_leaf12", "ternary_leaf40", "leaf300_multi_tile_leaves", "complex_leaf20", "one_triangle", "rect_partition"])
def test_grouped_phase_a_equals_per_tile_scheme(built, oracle, monkeypatch, case):
    """Phase A streams GROUPS of source tiles (a leaf inside a group gets its t = V x from one workgroup, leaves above one
    partial per group).  HTOOL_PHASE_A_GROUP=1 builds the same operator with one tile per group (one partial per tile, the
    round-1 scheme): same panels, other summation order -- the products agree to rounding, each is bitwise repeatable, and
    both meet the dense operator."""
    import Htool

    O = oracle
    np.random.seed(5)
    cplx = case == "complex_leaf20"
    n = 5000
    leaf, children = {"binary_leaf12": (12, 2), "ternary_leaf40": (40, 3), "leaf300_multi_tile_leaves": (300, 2), "complex_leaf20": (20, 2),
                      "one_triangle": (25, 2), "rect_partition": (30, 2)}[case]
    T = O.points_in_sphere(n)
    S = O.points_in_sphere(3100) + np.array([[0.3], [0.1], [0.0]]) if case == "rect_partition" else T
    world = 3 if case == "rect_partition" else 1
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(leaf)
    tcl = b.create_cluster_tree(T, children, size_of_partition=world)
    scl = tcl if S is T else b.create_cluster_tree(S, children)
    sym = ("S", "L") if case == "one_triangle" else ("N", "N")
    ys = []
    for group in ("512", "1", "100000"):
        monkeypatch.setenv("HTOOL_PHASE_A_GROUP", group)
        if cplx:
            H = Htool.ComplexHMatrixTreeBuilder(1e-5, 10.0, *sym).build(Htool.ComplexNativeGenerator("helmholtz", T, S, 5.0), tcl, scl)
        else:
            H = Htool.HMatrixTreeBuilder(1e-5, 10.0, *sym).build(Htool.NativeGenerator("laplace", T, S), tcl, scl, 1 if world > 1 else -1)
        x = np.random.RandomState(1).rand(S.shape[1]) + (1j * np.random.RandomState(2).rand(S.shape[1]) if cplx else 0)
        y = H * x
        assert np.array_equal(y, H * x)
        X = np.asfortranarray(np.stack([x, 2 * x, -x, x, x, 0.5 * x, x, x, x, 3 * x, x], axis=1))  # 11 columns: the 16-wide sweep (real), 8 + 2 + 1 (complex)
        Y = H @ X
        assert np.linalg.norm(Y[:, 0] - y) <= 1e-13 * np.linalg.norm(y) and np.linalg.norm(Y[:, 9] - 3 * y) <= 1e-13 * np.linalg.norm(y)
        ys.append(y)
        del H
    assert np.linalg.norm(ys[0] - ys[1]) <= 1e-13 * np.linalg.norm(ys[1]) and np.linalg.norm(ys[2] - ys[1]) <= 1e-13 * np.linalg.norm(ys[1])
    if world == 1:
        ye = O.dense_matvec(O.K_HELMHOLTZ if cplx else O.K_LAPLACE, T, S, x, 5.0 if cplx else 0.0)
        assert np.linalg.norm(ys[0] - ye) / np.linalg.norm(ye) < 1e-5


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("n,restart", [(5000, 12), (70001, 40), (300, 150)])
def test_gmres_device_kernels_match_the_textbook_history(built, cplx, n, restart):
    """The GPU path of krylov.gmres -- library GEMVs for the two Gram-Schmidt passes and ONE launch of this library for the tail
    (htool_krylov_finish_step: second projection taken out, norm, scaling, coefficient row) -- on a dense operator living on the device: residual history equal to a textbook GMRES
    (numpy, modified Gram-Schmidt + explicit least squares) to 1e-11 of the first residual, true residual equal to the last
    estimate, lockstep block of right-hand sides equal to the single solves, repeatable.  Sizes: several workgroups per vector
    with a ragged last one (70 001 = 136 x 512 + 369), one workgroup (300) with a long basis."""
    import torch

    from htool_python_amd import krylov
    from htool_python_amd.krylov import gmres
    from tests.test_krylov_cpu import _reference_gmres_history

    rng = np.random.RandomState(3)
    if n <= 5000:
        A = rng.rand(n, n) + (1j * rng.rand(n, n) if cplx else 0) + n * 0.06 * np.eye(n)
        At = torch.from_numpy(A).cuda()
        apply = lambda v: At @ v  # noqa: E731
        Anp = A
    else:  # a banded operator applied without forming it: diagonal + two shifted copies
        d0 = 2.0 + rng.rand(n) + (0.3j * rng.rand(n) if cplx else 0)
        d0t = torch.from_numpy(d0).cuda()
        apply = lambda v: d0t * v + 0.4 * torch.roll(v, 1) + 0.3 * torch.roll(v, -7)  # noqa: E731
        Anp = None
    b = rng.rand(n) + (1j * rng.rand(n) if cplx else 0)
    bt = torch.from_numpy(b).cuda()
    iters = 2 * restart + 5 if n > 300 else 200
    s0, r0 = krylov.HOST_SYNCS, krylov.REDUCE_CALLS
    x, info = gmres(apply, bt, tol=0.0 if n > 300 else 1e-12, restart=restart, max_it=iters, reduce=lambda t: t)
    if n > 300:
        assert info["iterations"] == iters and krylov.REDUCE_CALLS - r0 == 2 * iters + info["restarts"] + 1 and krylov.HOST_SYNCS - s0 == iters + info["restarts"] + 1
    got = np.array(info["residuals"])
    if Anp is not None:
        ref = _reference_gmres_history(Anp, b, restart, len(got))
        assert np.max(np.abs(got - ref)) <= 1e-11 * ref[0] + 1e-15, np.max(np.abs(got - ref))
    true_res = float(torch.linalg.norm(bt - apply(x)) / torch.linalg.norm(bt))
    assert abs(true_res - got[-1]) <= 1e-9 + 1e-6 * got[-1]
    x2, info2 = gmres(apply, bt, tol=0.0 if n > 300 else 1e-12, restart=restart, max_it=iters, reduce=lambda t: t)
    assert torch.equal(x, x2) and info2["residuals"] == info["residuals"]      # no atomics anywhere: repeatable
    if n == 5000:  # a block of right-hand sides in lockstep = the single solves
        B = torch.stack([bt, 2.0 * bt, torch.zeros_like(bt), torch.from_numpy(rng.rand(n) + (1j * rng.rand(n) if cplx else 0)).cuda()])
        X, bi = gmres(lambda Z: Z @ At.t(), B, tol=1e-10, restart=restart, max_it=400)
        for c in (0, 1, 3):
            xc, ic = gmres(apply, B[c], tol=1e-10, restart=restart, max_it=400)
            assert ic["iterations"] == bi["iterations_per_column"][c]
            assert float(torch.linalg.norm(X[c] - xc)) <= 1e-9 * float(torch.linalg.norm(xc))
        assert bi["iterations_per_column"][2] == 0 and float(torch.linalg.norm(X[2])) == 0.0
