"""GPU parity tests of the device build (hot-path rows a4/a6 of SURVEY.md section 8): device ACA + device
dense evaluation for native generators, against the CPU restatement (oracle/) on the same inputs and
against the exact dense kernel.  Leaf-level quantities are "parity unpinned" in the reference
(SURVEY.md 8c); the bar here is the north star's: leaf ranks / errors match the CPU ACA.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(points_t, points_s, kind, p0, eps, eta, leaf, complex_=False):
    import Htool
    from tests.helpers import cluster_of

    tcl = cluster_of(points_t, leaf)
    scl = tcl if points_s is points_t else cluster_of(points_s, leaf)
    name = {0: "inv_delta", 1: "laplace", 2: "helmholtz"}[kind]
    if complex_:
        gen = Htool.ComplexNativeGenerator(name, points_t, points_s, p0)
        H = Htool.ComplexHMatrixTreeBuilder(eps, eta, "N", "N").build(gen, tcl, scl)
    else:
        gen = Htool.NativeGenerator(name, points_t, points_s, p0)
        H = Htool.HMatrixTreeBuilder(eps, eta, "N", "N").build(gen, tcl, scl)
    return H, tcl, scl


@pytest.mark.parametrize("n,leaf,eta,eps,kind,p0", [
    (2000, 10, 10.0, 1e-3, 0, 0.1),
    (8000, 64, 10.0, 1e-4, 1, 0.0),
    (6000, 100, 2.0, 1e-6, 1, 0.0),
])
def test_device_aca_matches_cpu_aca(built, oracle, n, leaf, eta, eps, kind, p0):
    O = oracle
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    H, tcl, scl = _build(pts, pts, kind, p0, eps, eta, leaf)
    oc = O.Cluster(pts, max_leaf=leaf)
    assert np.array_equal(oc.perm, np.asarray(tcl.get_permutation()))
    OH = O.HMatrix(oc, oc, kind, p0, eps=eps, eta=eta)
    mine = {tuple(l[:4]): int(l[4]) for l in np.asarray(H.leaves())}
    theirs = {tuple(l[:4]): int(l[4]) for l in OH.leaves}
    # same block structure
    assert set(mine) == set(theirs)
    diff = np.array([mine[k] - theirs[k] for k in mine])
    # same ranks: identical pivots give identical ranks; the only freedom is the summation order of the
    # norms in the stopping test, which may move a borderline leaf by one step
    assert np.mean(diff != 0) < 0.01, f"{np.mean(diff != 0):.4f} of the leaves differ in rank"
    assert np.abs(diff).max() <= 1
    np.random.seed(1)
    x = np.random.rand(n)
    y, y_cpu = H * x, OH.matvec(x)
    y_exact = O.dense_matvec(kind, pts, pts, x, p0)
    e_gpu = np.linalg.norm(y - y_exact) / np.linalg.norm(y_exact)
    e_cpu = np.linalg.norm(y_cpu - y_exact) / np.linalg.norm(y_exact)
    assert e_gpu < eps
    assert e_gpu < 1.5 * e_cpu + 1e-14
    assert np.linalg.norm(y - y_cpu) / np.linalg.norm(y_cpu) < 2 * eps
    # independent of the C++ oracle: >= 200 sampled admissible leaves against the explicit-residual numpy ACA, the SVD
    # epsilon-rank and the exact block (oracle/independent.py)
    from tests.helpers import independent_leaf_checks

    stats = independent_leaf_checks(H, pts, pts, kind, p0, eps, n_sample=250, max_block=800, min_leaves=20)
    assert stats["leaves"] >= 200 or leaf == 100  # (the eta = 2 tree has fewer admissible leaves than that)


def test_device_panels_equal_cpu_panels(built, oracle):
    """Leaf by leaf: U V^T of the device ACA equals the CPU ACA's to rounding (same pivots)."""
    O = oracle
    n, leaf, eps, eta = 3000, 32, 1e-5, 10.0
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    H, tcl, _ = _build(pts, pts, 1, 0.0, eps, eta, leaf)
    oc = O.Cluster(pts, max_leaf=leaf)
    OH = O.HMatrix(oc, oc, O.K_LAPLACE, eps=eps, eta=eta)
    theirs = {tuple(l[:4]): i for i, l in enumerate(OH.leaves)}
    L = np.asarray(H.leaves())
    rng = np.random.RandomState(0)
    lr = [i for i in range(len(L)) if L[i, 4] > 0]
    dn = [i for i in range(len(L)) if L[i, 4] < 0]
    for i in list(rng.choice(lr, 40, replace=False)) + list(rng.choice(dn, 10, replace=False)):
        A, B = H.leaf_panels(int(i))
        Ao, Bo = OH.leaf_data(theirs[tuple(L[i, :4])])
        if L[i, 4] < 0:
            assert np.array_equal(np.asarray(A), Ao)  # same kernel arithmetic on CPU and GPU: bitwise
        elif L[i, 4] == OH.leaves[theirs[tuple(L[i, :4])], 4]:
            blk, blk_o = np.asarray(A) @ np.asarray(B), Ao @ Bo
            assert np.linalg.norm(blk - blk_o) <= 1e-12 * np.linalg.norm(blk_o)


def test_native_rectangular_and_2d(built, oracle):
    O = oracle
    np.random.seed(0)
    T, S = np.random.random((2, 1500)), np.random.random((2, 700))
    S[0] += 0.5
    H, _, _ = _build(T, S, 0, 0.1, 1e-6, 10.0, 10)
    assert H.shape == (1500, 700)
    x = np.random.rand(700)
    ye = O.dense_matvec(0, T, S, x, 0.1)
    assert np.linalg.norm(H * x - ye) / np.linalg.norm(ye) < 1e-6


def test_native_helmholtz_complex(built, oracle):
    O = oracle
    n = 6000
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    H, _, _ = _build(pts, pts, 2, 8.0, 1e-4, 10.0, 50, complex_=True)
    x = np.random.rand(n) + 1j * np.random.rand(n)
    y = H * x
    ye = O.dense_matvec(2, pts, pts, x, 8.0)
    assert np.linalg.norm(y - ye) / np.linalg.norm(ye) < 1e-4
    oc = O.Cluster(pts, max_leaf=50)
    OH = O.HMatrix(oc, oc, 2, 8.0, is_complex=True, eps=1e-4, eta=10.0)
    mine = {tuple(l[:4]): int(l[4]) for l in np.asarray(H.leaves())}
    theirs = {tuple(l[:4]): int(l[4]) for l in OH.leaves}
    assert set(mine) == set(theirs)
    diff = np.array([mine[k] - theirs[k] for k in mine])
    assert np.mean(diff != 0) < 0.02 and np.abs(diff).max() <= 1


def test_reqrank(built, oracle):
    import Htool
    from tests.helpers import cluster_of

    O = oracle
    np.random.seed(0)
    pts = O.points_in_sphere(2000)
    cl = cluster_of(pts, 32)
    gen = Htool.NativeGenerator("laplace", pts, pts)
    H = Htool.HMatrixTreeBuilder(1e-3, 10.0, "N", "N", reqrank=5).build(gen, cl, cl)
    L = np.asarray(H.leaves())
    assert set(L[L[:, 4] >= 0, 4]) == {5}


@pytest.mark.parametrize("n,eps,kind,p0,leaf", [
    (100_000, 1e-4, 1, 0.0, 100),     # BASELINE config C2
    (1_000_000, 1e-3, 1, 0.0, 100),   # BASELINE config C4's operator on one GPU (the bench workload)
    (1_000_000, 1e-3, 2, 10.0, 100),  # BASELINE config C3: 1 M-point Helmholtz, kappa = 10, complex128 (207 GB of panels)
], ids=["C2-100k-laplace", "C4op-1M-laplace", "C3-1M-helmholtz-c128"])
def test_full_size_configs_by_properties(built, oracle, n, eps, kind, p0, leaf):
    """At BASELINE.json's full sizes the oracle is too slow to run end to end; check size-independent
    properties instead: exact rows sampled from the dense operator, linearity, bitwise reproducibility,
    leaves tiling the matrix, sampled leaves against the exact kernel block, and the CPU leaf loop on a sampled
    subset of the device's own panels (isolated leaf by leaf, helpers.single_leaf_product_checks).
    Reference bar: tests/test_hmatrix.py:83 (relative error of the product below epsilon)."""
    import Htool
    from htool_python_amd.workloads import points_in_sphere
    from tests.helpers import cluster_of, single_leaf_product_checks

    O = oracle
    complex_ = kind == 2
    dtype = np.complex128 if complex_ else np.float64
    pts = points_in_sphere(n, seed=0)
    cl = cluster_of(pts, leaf)
    name = {1: "laplace", 2: "helmholtz"}[kind]
    if complex_:
        H = Htool.ComplexHMatrixTreeBuilder(eps, 10.0, "N", "N").build(Htool.ComplexNativeGenerator(name, pts, pts, p0), cl, cl)
    else:
        H = Htool.HMatrixTreeBuilder(eps, 10.0, "N", "N").build(Htool.NativeGenerator(name, pts, pts, p0), cl, cl)
    assert H.shape == (n, n)
    rng = np.random.RandomState(0)
    x, z = rng.rand(n).astype(dtype), rng.rand(n).astype(dtype)
    if complex_:
        x, z = x + 1j * rng.rand(n), z - 0.5j * rng.rand(n)
    y = H * x
    rows = rng.choice(n, 200, replace=False)
    ye = O.dense_matvec(kind, pts, pts, x, p0, rows=rows)
    assert np.linalg.norm(y[rows] - ye) / np.linalg.norm(ye) < eps
    assert np.array_equal(y, H * x)
    yz = H * z
    assert np.linalg.norm(H * (x - 3.0 * z) - (y - 3.0 * yz)) / np.linalg.norm(y) < 1e-12
    L = np.asarray(H.leaves()).astype(np.int64)
    assert (L[:, 1] * L[:, 3]).sum() == n * n
    # sampled low-rank leaves against the exact kernel block
    small = np.flatnonzero((L[:, 4] > 0) & (L[:, 1] <= 256) & (L[:, 3] <= 256))
    perm = np.asarray(cl.get_permutation())
    for i in rng.choice(small, 20, replace=False):
        t_off, m, s_off, nn, r = L[i]
        U, V = H.leaf_panels(int(i))
        blk = np.asarray(U) @ np.asarray(V)
        exact = O.kernel_block(kind, pts[:, perm[t_off:t_off + m]], pts[:, perm[s_off:s_off + nn]], p0)
        # partial-pivot ACA stops on a heuristic estimate: a single leaf may miss eps by a small factor
        assert np.linalg.norm(blk - exact) <= 10 * eps * np.linalg.norm(exact)
    # CPU leaf loop on a sampled subset of the device's own panels
    assert single_leaf_product_checks(H, cl, cl, dtype, n_sources=3, n_targets=12) >= 18
    # independent of the C++ oracle: 200 sampled admissible leaves against the explicit-residual numpy ACA, the SVD
    # epsilon-rank and the exact block (oracle/independent.py)
    from tests.helpers import independent_leaf_checks

    stats = independent_leaf_checks(H, pts, pts, kind, p0, eps, n_sample=200, max_block=500, seed=1)
    assert stats["leaves"] == 200
    # transposed product at full size (tables made on first use; no row tile is cut in slices at 1 M points): the adjoint
    # identity  w . (H x) = (H^T w) . x  and sampled exact entries (the kernels are symmetric: row j of A is column j)
    if True:  # (round 3: also for the complex 1 M-point operator, VERDICT round 2 weak spot 7)
        zt = H.transposed_mul(z, "T")
        lhs, rhs = np.sum(z * y), np.sum(zt * x)  # (no conjugation: the transpose, not the adjoint)
        assert abs(lhs - rhs) < 1e-11 * abs(lhs)
        ze = O.dense_matvec(kind, pts, pts, z, p0, rows=rows)
        assert np.linalg.norm(zt[rows] - ze) / np.linalg.norm(ze) < eps
        assert np.array_equal(H * x, y)
    del H
    Htool.release_workspace()


@pytest.mark.parametrize("world,ranks", [(8, (0, 3, 7)), (4, (2,)), (2, (1,))], ids=["8-way", "4-way", "2-way"])
def test_full_size_c4_row_split_per_rank_builds(built, oracle, world, ranks):
    """BASELINE config C4: the 1 M-point Laplace operator split by rows over 2 / 4 / 8 ranks (size_of_partition = world,
    DefaultApproximationBuilder's decomposition: rank p builds rows(partition p) x all columns,
    src/htool/distributed_operator/utility.hpp:26).  Some ranks of every split are built one after the other on this GPU and each
    is checked on its own rows: exact sampled rows, the cluster-numbered device path matvec_device(numbering=1) the
    multi-GPU loop uses, bitwise repeatability, tiling of its row block, single-leaf CPU products."""
    import torch

    import Htool
    from htool_python_amd.workloads import points_in_sphere
    from tests.helpers import single_leaf_product_checks

    O = oracle
    n, eps = 1_000_000, 1e-3
    pts = points_in_sphere(n, seed=0)
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(100)
    cl = b.create_cluster_tree(pts, 2, size_of_partition=world)
    perm = np.asarray(cl.get_permutation())
    sizes = [cl.get_cluster_on_partition(p).get_size() for p in range(world)]
    assert sum(sizes) == n and max(sizes) - min(sizes) <= world
    gen = Htool.NativeGenerator("laplace", pts, pts)
    rng = np.random.RandomState(0)
    x = rng.rand(n)
    x_cluster = torch.from_numpy(x[perm]).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    for p in ranks:
        sub = cl.get_cluster_on_partition(p)
        off, size = sub.get_offset(), sub.get_size()
        H = Htool.HMatrixTreeBuilder(eps, 10.0, "N", "N").build(gen, cl, cl, p)
        assert H.shape == (size, n)
        L = np.asarray(H.leaves()).astype(np.int64)
        assert (L[:, 1] * L[:, 3]).sum() == size * n
        assert L[:, 0].min() == off and (L[:, 0] + L[:, 1]).max() == off + size
        y_local = H * x  # host API of a partition-built operator: its rows in cluster order
        assert y_local.shape == (size,)
        idx = rng.choice(size, 100, replace=False)
        ye = O.dense_matvec(O.K_LAPLACE, pts, pts, x, 0.0, rows=perm[off + idx])
        assert np.linalg.norm(y_local[idx] - ye) / np.linalg.norm(ye) < eps
        # the path of the multi-GPU loop: whole permuted x in, local row slice out, both in cluster numbering
        y_dev = torch.zeros(size, dtype=torch.float64, device="cuda")
        H.matvec_device(x_cluster.data_ptr(), y_dev.data_ptr(), 1, stream)
        torch.cuda.synchronize()
        assert np.array_equal(y_dev.cpu().numpy(), y_local)
        H.matvec_device(x_cluster.data_ptr(), y_dev.data_ptr(), 1, stream)
        torch.cuda.synchronize()
        assert np.array_equal(y_dev.cpu().numpy(), y_local)
        assert single_leaf_product_checks(H, cl, cl, np.float64, n_sources=2, n_targets=10, row_window=(off, size)) >= 10
        del H
    Htool.release_workspace()


def test_full_size_c5_gmres_500k(built, oracle):
    """BASELINE config C5: 500 000-point Laplace H-matrix as the operator of GMRES, 50 iterations on device-resident
    vectors (use_ddm_solver.py-style: b = A x_ref, solve, compare).  The system is (shift I + H) with the shift a fixed
    fraction of the operator norm, chosen so that the 50 iterations are needed (residual 1e-6 is not reached early).
    Reference bar (tests/test_ddm_solver.py:659-660): convergence error below the tolerance, solution error below 10 eps;
    on top the TRUE residual is recomputed from exact rows of the dense operator."""
    import torch

    import Htool
    from htool_python_amd.krylov import gmres
    from htool_python_amd.solver import DeviceOperator
    from htool_python_amd.workloads import gmres_shift, points_in_sphere
    from tests.helpers import cluster_of

    O = oracle
    n, eps = 500_000, 1e-3
    pts = points_in_sphere(n, seed=0)
    cl = cluster_of(pts, 100)
    H = Htool.HMatrixTreeBuilder(eps, 10.0, "N", "N").build(Htool.NativeGenerator("laplace", pts, pts), cl, cl)
    plain = DeviceOperator(H, None, 0, None, 0.0)
    shift, norm_est = gmres_shift(plain.apply, n, torch.float64)
    op = DeviceOperator(H, None, 0, None, shift)
    g = torch.Generator(device="cpu").manual_seed(7)
    x_ref = torch.rand(n, dtype=torch.float64, generator=g).cuda()
    bvec = op.apply(x_ref)
    xs, info = gmres(op.apply, bvec, tol=1e-6, restart=50, max_it=50)
    res = info["residuals"]
    assert info["iterations"] <= 50 and res[-1] <= 1e-6, (info["iterations"], res[-1], shift, norm_est)
    assert res[14] > 1e-6, "the system is too easy: GMRES converged within 15 iterations"
    # convergence error against the operator itself, and the solution error (reference bar)
    r = bvec - op.apply(xs)
    assert float(torch.linalg.norm(r) / torch.linalg.norm(bvec)) < 2e-6
    assert float(torch.linalg.norm(xs - x_ref) / torch.linalg.norm(x_ref)) < 10 * eps
    # true residual from exact rows of the dense operator: limited by the H-matrix accuracy, not by the solver
    perm = np.asarray(cl.get_permutation())
    x_user = np.empty(n)
    x_user[perm] = xs.cpu().numpy()
    b_user = np.empty(n)
    b_user[perm] = bvec.cpu().numpy()
    rows = np.random.RandomState(0).choice(n, 256, replace=False)
    ax = O.dense_matvec(O.K_LAPLACE, pts, pts, x_user, 0.0, rows=rows) + shift * x_user[rows]
    assert np.linalg.norm(b_user[rows] - ax) / np.linalg.norm(b_user[rows]) < 2 * eps
    del H, op, plain
    Htool.release_workspace()


@pytest.mark.parametrize("native", [True, False])
def test_symmetric_operator_is_exactly_symmetric(built, oracle, native):
    """"sym" role rule of the ACA (leaves below the diagonal are compressed through their transpose): for a
    symmetric kernel on one cluster tree the leaves (t,s) and (s,t) carry exactly transposed factors, so the
    H-matrix is symmetric bit for bit, and the engine (storing both triangles here) agrees to rounding with
    the oracle's one-triangle storage ('S','L' and 'S','U') that applies stored leaves transposed."""
    import Htool
    from tests.helpers import NumpyGenerator, cluster_of

    O = oracle
    n = 1500
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    cl = cluster_of(pts, 16)
    gen = Htool.NativeGenerator("inv_delta", pts, pts, 0.1) if native else NumpyGenerator(pts, pts)
    for sym, uplo in (("N", "N"), ("S", "L"), ("S", "U")):
        builder = Htool.HMatrixTreeBuilder(1e-5, 10.0, sym, uplo)
        builder.set_symmetric_storage(False)  # both triangles (the default keeps the UPLO triangle only)
        H = builder.build(gen, cl, cl)
        L = np.asarray(H.leaves())
        index = {tuple(l[:4]): i for i, l in enumerate(L)}
        pairs = [(i, index[(l[2], l[3], l[0], l[1])]) for i, l in enumerate(L) if l[0] > l[2]]
        assert len(pairs) > 50
        rng = np.random.RandomState(0)
        for i, j in [pairs[q] for q in rng.choice(len(pairs), 40, replace=False)]:
            assert L[i, 4] == L[j, 4]
            A1, B1 = H.leaf_panels(int(i))
            A2, B2 = H.leaf_panels(int(j))
            if L[i, 4] < 0:
                assert np.array_equal(np.asarray(A1), np.asarray(A2).T)  # dense leaves: same entries
            else:
                assert np.array_equal(np.asarray(A1), np.asarray(B2).T) and np.array_equal(np.asarray(B1), np.asarray(A2).T)
        # the densified operator is symmetric up to the summation order of the product kernel
        D = H.to_dense_in_user_numbering()
        assert np.abs(D - D.T).max() <= 1e-14 * np.abs(D).max()
    oc = O.Cluster(pts, max_leaf=16)
    x = np.random.rand(n)
    y = H * x
    for uplo in ("L", "U"):
        OH = O.HMatrix(oc, oc, O.K_INV_DELTA, 0.1, eps=1e-5, eta=10.0, symmetry="S", uplo=uplo)
        assert np.linalg.norm(y - OH.matvec(x)) / np.linalg.norm(y) < 1e-12


@pytest.mark.parametrize("eps,eta,leaf", [(1e-12, 10.0, 16), (1e-9, 100.0, 8)])
def test_high_accuracy_exercises_capacity_growth_and_resplit(built, oracle, eps, eta, leaf):
    """Tight tolerances push ranks past the arena's initial capacity (retry with twice the room) and make many
    admissible leaves not worth compressing (re-split on the host): structure and ranks must still
    match the CPU restatement and the product the exact operator."""
    O = oracle
    n = 2500
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    H, tcl, _ = _build(pts, pts, 1, 0.0, eps, eta, leaf)
    oc = O.Cluster(pts, max_leaf=leaf)
    OH = O.HMatrix(oc, oc, O.K_LAPLACE, eps=eps, eta=eta)
    mine = {tuple(l[:4]): int(l[4]) for l in np.asarray(H.leaves())}
    theirs = {tuple(l[:4]): int(l[4]) for l in OH.leaves}
    assert set(mine) == set(theirs)
    diff = np.array([mine[k] - theirs[k] for k in mine])
    assert np.mean(diff != 0) < 0.02 and np.abs(diff).max() <= 1
    assert max(mine.values()) > 40  # beyond the initial capacity of the temporary arena
    x = np.random.rand(n)
    ye = O.dense_matvec(O.K_LAPLACE, pts, pts, x)
    assert np.linalg.norm(H * x - ye) / np.linalg.norm(ye) < max(10 * eps, 1e-13)


@pytest.mark.parametrize("kind,p0,symmetry,uplo", [(0, 0.1, "S", "L"), (0, 0.1, "N", "N"), (1, 0.0, "S", "L"), (1, 0.0, "N", "N")])
def test_config_c1_use_hmatrix_10k(built, oracle, kind, p0, symmetry, uplo):
    """BASELINE config C1: example/use_hmatrix.py scaled to 10 000 points (eta=10, eps=1e-3, leaf 50, 'S','L' as in
    use_hmatrix.py:42 and 'N'), both kernels of SURVEY 8d: full parity -- block structure and ranks equal the CPU
    restatement's, product equal to the oracle's product and to the exact dense operator."""
    import Htool
    from htool_python_amd.workloads import points_in_sphere
    from tests.helpers import cluster_of

    O = oracle
    n, eps, eta, leaf = 10_000, 1e-3, 10.0, 50
    pts = points_in_sphere(n, seed=0)
    cl = cluster_of(pts, leaf)
    name = {0: "inv_delta", 1: "laplace"}[kind]
    H = Htool.HMatrixTreeBuilder(eps, eta, symmetry, uplo).build(Htool.NativeGenerator(name, pts, pts, p0), cl, cl)
    np.random.seed(0)
    x = np.random.rand(n)  # use_hmatrix.py:48-49
    y = H * x
    y_exact = O.dense_matvec(kind, pts, pts, x, p0)
    assert np.linalg.norm(y - y_exact) / np.linalg.norm(y_exact) < eps
    oc = O.Cluster(pts, max_leaf=leaf)
    OH = O.HMatrix(oc, oc, kind, p0, eps=eps, eta=eta, symmetry=symmetry, uplo=uplo)
    assert np.linalg.norm(y - OH.matvec(x)) / np.linalg.norm(y_exact) < 1e-5  # ranks may differ by one step on < 1 % of the leaves
    mine = {tuple(l[:4]): int(l[4]) for l in np.asarray(H.leaves())}
    theirs = {tuple(l[:4]): int(l[4]) for l in OH.leaves}
    assert set(mine) == set(theirs)  # 'S','L': the same triangle is stored, leaf for leaf
    diff = np.array([mine[k] - theirs[k] for k in theirs])
    assert np.mean(diff != 0) < 0.01 and np.abs(diff).max() <= 1
    X = np.random.rand(n, 2)
    Y = H @ X
    Ye = O.dense_matvec(kind, pts, pts, X, p0)
    assert np.linalg.norm(Y - Ye) / np.linalg.norm(Ye) < eps


@pytest.mark.parametrize("native", [True, False])
@pytest.mark.parametrize("eps,leaf", [(1e-3, 50), (1e-5, 16)])
def test_recompression(built, oracle, native, eps, leaf):
    """Htool.recompression(hmatrix) (src/htool/hmatrix/hmatrix.hpp:96-99): SVD recompression of the low-rank leaves
    on the device.  Ranks shrink to (about) the numpy-SVD truncation of the SAME panels under the rule of
    example/advanced/define_custom_low_rank_generator.py:16-24, the product stays within epsilon of the exact
    operator, dense leaves are untouched, and the re-packed operator still equals the CPU leaf loop on its panels."""
    import copy

    import Htool
    from tests.helpers import NumpyGenerator, cluster_of, cpu_leaf_loop

    O = oracle
    n = 4000
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    cl = cluster_of(pts, leaf)
    gen = Htool.NativeGenerator("laplace", pts, pts) if native else NumpyGenerator(pts, pts, O.K_LAPLACE, 0.0)
    H = Htool.HMatrixTreeBuilder(eps, 10.0, "N", "N").build(gen, cl, cl)
    H0 = copy.deepcopy(H)
    L0 = np.asarray(H0.leaves())
    reduced = Htool.recompression(H)
    L1 = np.asarray(H.leaves())
    assert np.array_equal(L0[:, :4], L1[:, :4]) and np.all(L1[:, 4] <= L0[:, 4])
    assert np.array_equal(L1[L0[:, 4] < 0, 4], L0[L0[:, 4] < 0, 4])  # dense leaves stay dense
    assert reduced == int((L1[:, 4] < L0[:, 4]).sum()) and reduced > 0.3 * (L0[:, 4] > 1).sum()
    x = np.random.rand(n)
    ye = O.dense_matvec(O.K_LAPLACE, pts, pts, x)
    y = H * x
    assert np.linalg.norm(y - ye) / np.linalg.norm(ye) < eps
    assert np.linalg.norm(y - cpu_leaf_loop(H, x)) / np.linalg.norm(y) < 1e-12
    # leaf by leaf against numpy's SVD truncation of the ORIGINAL panels
    rng = np.random.RandomState(1)
    lr = np.flatnonzero(L0[:, 4] >= 2)
    same = 0
    sel = rng.choice(lr, 60, replace=False)
    for i in sel:
        U0, V0 = (np.asarray(a) for a in H0.leaf_panels(int(i)))
        U1, V1 = (np.asarray(a) for a in H.leaf_panels(int(i)))
        A0 = U0 @ V0
        s = np.linalg.svd(A0, compute_uv=False)[: U0.shape[1]]
        tot = (s ** 2).sum()
        r_ref = len(s)
        while r_ref > 1 and (s[r_ref - 1:] ** 2).sum() <= eps * eps * tot:
            r_ref -= 1
        same += int(U1.shape[1] == r_ref)
        assert abs(U1.shape[1] - r_ref) <= 1
        assert np.linalg.norm(U1 @ V1 - A0) <= 1.05 * eps * np.linalg.norm(A0)
    assert same >= 0.9 * len(sel)
    # the untouched copy still multiplies like before; a second pass changes (almost) nothing
    assert np.linalg.norm(H0 * x - ye) / np.linalg.norm(ye) < eps
    # a second pass truncates the already truncated leaves once more (each pass may discard up to eps^2 of a
    # leaf's energy): fewer leaves change, and the product stays within 2 eps
    again = Htool.openmp_recompression(H)
    assert again < reduced
    assert np.linalg.norm(H * x - ye) / np.linalg.norm(ye) < 2 * eps
    if native:
        X = np.asfortranarray(np.random.rand(n, 3))
        Y = H @ X
        for c in range(3):
            assert np.array_equal(Y[:, c], H * np.ascontiguousarray(X[:, c]))


def test_recompression_complex(built, oracle):
    """Same as test_recompression for a complex operator (Helmholtz): ranks against numpy's SVD truncation of the
    original panels, leaf accuracy, product accuracy, re-packed panels consistent with the CPU leaf loop."""
    import copy

    import Htool
    from tests.helpers import cluster_of, cpu_leaf_loop

    O = oracle
    n, eps, kappa = 4000, 1e-4, 6.0
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    cl = cluster_of(pts, 24)
    H = Htool.ComplexHMatrixTreeBuilder(eps, 10.0, "N", "N").build(Htool.ComplexNativeGenerator("helmholtz", pts, pts, kappa), cl, cl)
    H0 = copy.deepcopy(H)
    L0 = np.asarray(H0.leaves())
    reduced = Htool.recompression(H)
    L1 = np.asarray(H.leaves())
    assert reduced == int((L1[:, 4] < L0[:, 4]).sum()) and reduced > 0.3 * (L0[:, 4] > 1).sum()
    x = np.random.rand(n) + 1j * np.random.rand(n)
    ye = O.dense_matvec(O.K_HELMHOLTZ, pts, pts, x, kappa)
    y = H * x
    assert np.linalg.norm(y - ye) / np.linalg.norm(ye) < eps
    assert np.linalg.norm(y - cpu_leaf_loop(H, x, True)) / np.linalg.norm(y) < 1e-12
    rng = np.random.RandomState(1)
    sel = rng.choice(np.flatnonzero(L0[:, 4] >= 2), 50, replace=False)
    same = 0
    for i in sel:
        U0, V0 = (np.asarray(a) for a in H0.leaf_panels(int(i)))
        U1, V1 = (np.asarray(a) for a in H.leaf_panels(int(i)))
        A0 = U0 @ V0
        s = np.linalg.svd(A0, compute_uv=False)[: U0.shape[1]]
        tot = (s ** 2).sum()
        r_ref = len(s)
        while r_ref > 1 and (s[r_ref - 1:] ** 2).sum() <= eps * eps * tot:
            r_ref -= 1
        same += int(U1.shape[1] == r_ref)
        assert abs(U1.shape[1] - r_ref) <= 1
        assert np.linalg.norm(U1 @ V1 - A0) <= 1.05 * eps * np.linalg.norm(A0)
    assert same >= 0.9 * len(sel)


@pytest.mark.parametrize("arena_mb", [2, 12, 40])
def test_multi_round_build_with_held_factors(built, oracle, monkeypatch, arena_mb):
    """A small temporary arena forces several ACA rounds: finished factors are compacted to the front of the arena and
    packed in few batches (2 MB: leaves larger than the staging buffer take the direct path).  The operator must be
    the one a single-round build gives, bit for bit at leaf level."""
    O = oracle
    n, leaf, eps, eta = 12000, 40, 1e-5, 10.0
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    H1, _, _ = _build(pts, pts, 1, 0.0, eps, eta, leaf)
    monkeypatch.setenv("HTOOL_BUILD_ARENA_MB", str(arena_mb))
    H2, _, _ = _build(pts, pts, 1, 0.0, eps, eta, leaf)
    monkeypatch.delenv("HTOOL_BUILD_ARENA_MB")
    L1, L2 = np.asarray(H1.leaves()), np.asarray(H2.leaves())
    k1 = {tuple(l[:4]): (i, int(l[4])) for i, l in enumerate(L1)}
    k2 = {tuple(l[:4]): (i, int(l[4])) for i, l in enumerate(L2)}
    assert {k: v[1] for k, v in k1.items()} == {k: v[1] for k, v in k2.items()}
    rng = np.random.RandomState(0)
    keys = list(k1)
    for j in rng.choice(len(keys), 60, replace=False):
        A1, B1 = H1.leaf_panels(k1[keys[j]][0])
        A2, B2 = H2.leaf_panels(k2[keys[j]][0])
        assert np.array_equal(np.asarray(A1), np.asarray(A2))
        if k1[keys[j]][1] > 0:
            assert np.array_equal(np.asarray(B1), np.asarray(B2))
    x = np.random.rand(n)
    y1, y2 = H1 * x, H2 * x
    assert np.linalg.norm(y1 - y2) / np.linalg.norm(y1) < 1e-13
    y_exact = O.dense_matvec(1, pts, pts, x, 0.0)
    assert np.linalg.norm(y2 - y_exact) / np.linalg.norm(y_exact) < eps


def test_recompression_in_chunks(built, oracle, monkeypatch):
    """A small recompression arena splits a batch into several chunks (each becomes a batch of its own): same ranks and
    same operator as the single-chunk pass."""
    import Htool

    O = oracle
    n, leaf, eps, eta = 8000, 32, 1e-6, 10.0
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    H1, _, _ = _build(pts, pts, 1, 0.0, eps, eta, leaf)
    H2, _, _ = _build(pts, pts, 1, 0.0, eps, eta, leaf)
    Htool.recompression(H1, 1e-3)
    monkeypatch.setenv("HTOOL_RECOMPRESS_ARENA_MB", "3")
    Htool.recompression(H2, 1e-3)
    monkeypatch.delenv("HTOOL_RECOMPRESS_ARENA_MB")
    r1 = {tuple(l[:4]): int(l[4]) for l in np.asarray(H1.leaves())}
    r2 = {tuple(l[:4]): int(l[4]) for l in np.asarray(H2.leaves())}
    assert r1 == r2
    x = np.random.rand(n)
    y1, y2 = H1 * x, H2 * x
    assert np.linalg.norm(y1 - y2) / np.linalg.norm(y1) < 1e-13
    y_exact = O.dense_matvec(1, pts, pts, x, 0.0)
    assert 1e-8 < np.linalg.norm(y2 - y_exact) / np.linalg.norm(y_exact) < 5e-3
    assert np.array_equal((H2 @ np.asfortranarray(x[:, None]))[:, 0], y2)


def test_workspace_cache_is_reused_and_released(built, oracle):
    """Builds keep their temporary arena in a process-wide cache (a large hipFree would make the next allocation wait
    for the driver to scrub it); `Htool.release_workspace()` hands it back."""
    import Htool

    O = oracle
    np.random.seed(0)
    pts = O.points_in_sphere(5000)
    Htool.release_workspace()
    H1, _, _ = _build(pts, pts, 1, 0.0, 1e-4, 10.0, 32)
    x = np.random.rand(5000)
    y1 = H1 * x
    H2, _, _ = _build(pts, pts, 1, 0.0, 1e-4, 10.0, 32)   # second build runs in the cached arena
    assert np.array_equal(H2 * x, y1)
    Htool.recompression(H2, 1e-2)
    freed = Htool.release_workspace()
    assert freed > 0
    assert Htool.release_workspace() == 0
    H3, _, _ = _build(pts, pts, 1, 0.0, 1e-4, 10.0, 32)
    assert np.array_equal(H3 * x, y1)


def test_no_device_memory_leak(built, oracle):
    """build / multiply / multi-RHS / deep copy / recompress / destroy, both storages, in a loop: once the workspace
    cache is released the free device memory is back where it was (tools/leak_check.py does the same at 300 000 points)."""
    import copy
    import gc

    import torch

    import Htool
    from tests.helpers import cluster_of

    O = oracle
    n = 20000
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    cl = cluster_of(pts, 50)
    x = np.random.rand(n)

    def free_bytes():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    marks = []
    for _ in range(3):
        for sym, uplo in (("N", "N"), ("S", "L")):
            H = Htool.HMatrixTreeBuilder(1e-4, 10.0, sym, uplo).build(Htool.NativeGenerator("laplace", pts, pts, 0.0), cl, cl)
            H * x
            H @ np.asfortranarray(np.random.rand(n, 3))
            H2 = copy.deepcopy(H)
            Htool.recompression(H2, 1e-2)
            H2 * x
            del H, H2
            gc.collect()
        Htool.release_workspace()
        marks.append(free_bytes())
    assert abs(marks[-1] - marks[0]) < 32 << 20, marks


def test_more_work_items_than_one_launch_can_hold(built, oracle):
    """Millions of tiny leaves: the pack kernels have one workgroup per (leaf, tile) pair, and 26 million of them exceed
    the 2^32 work-items a single launch can address (the excess was silently dropped before the launches were sliced).
    Found by tools/fuzz.py (seed 44, case 538)."""
    import Htool
    from tools.fuzz import exact_rows

    rng = np.random.RandomState(5)

    def ball(m):
        p = rng.randn(3, m)
        p /= np.linalg.norm(p, axis=0)
        return np.asfortranarray(p * rng.rand(m) ** (1.0 / 3))

    n, ns, eps = 91668, 112393, 1.55e-3
    pt, ps = ball(n), np.asfortranarray(ball(ns) + 0.3)
    cb = Htool.ClusterTreeBuilder()
    cb.set_maximal_leaf_size(5)
    cb.set_partitioning_strategy(Htool.BoundingBoxRegular())
    ct, cs = cb.create_cluster_tree(pt, 4, size_of_partition=1), cb.create_cluster_tree(ps, 4, size_of_partition=1)
    H = Htool.HMatrixTreeBuilder(eps, 0.7, "N", "N").build(Htool.NativeGenerator("laplace", pt, ps, 0.0), ct, cs)
    st = H.stats()
    assert st["n_dense"] + st["n_low_rank"] > 8_000_000
    x = rng.rand(ns)
    y = H * x
    rows = rng.choice(n, 500, replace=False)
    ye = exact_rows("laplace", pt, ps, x, 0.0, rows)
    assert np.linalg.norm(y[rows] - ye) / np.linalg.norm(ye) < eps
    Htool.release_workspace()


@pytest.mark.parametrize("case", ["laplace", "inv_delta_eta2", "helmholtz", "high_accuracy", "reqrank", "rectangular"])
def test_lockstep_aca_of_the_largest_leaves(built, oracle, monkeypatch, case):
    """The lockstep ACA (csrc/device_aca_steps.inc: every phase of a pivot step one grid-wide launch over all leaves of the
    class) is meant for leaves of more than 8192 rows or columns; HTOOL_ACA_STEP_MIN sends ordinary leaves down that path so
    that it is compared, leaf by leaf, with the CPU ACA (ranks equal on >= 99 %, +-1 otherwise), with the build that uses the
    workgroup kernel (same ranks but for borderline leaves, factors equal to 1e-12 where the ranks agree) and with the exact
    operator.  `high_accuracy` makes leaves exceed their capacity and fail (retry / re-split), `reqrank` fixes the step count,
    chunks of 1024: leaves of 1200-2400 rows have two or three chunks per phase."""
    import Htool
    from tests.helpers import cluster_of

    O = oracle
    np.random.seed(0)
    cplx = case == "helmholtz"
    n, leaf, eta, eps, kind, p0, reqrank = {"laplace": (20000, 100, 10.0, 1e-4, 1, 0.0, -1), "inv_delta_eta2": (9000, 64, 2.0, 1e-6, 0, 0.1, -1),
                                            "helmholtz": (12000, 64, 10.0, 1e-4, 2, 6.0, -1), "high_accuracy": (6000, 16, 10.0, 1e-11, 1, 0.0, -1),
                                            "reqrank": (8000, 64, 10.0, 1e-3, 1, 0.0, 6), "rectangular": (9000, 50, 10.0, 1e-5, 0, 0.1, -1)}[case]
    pts = O.points_in_sphere(n)
    pts_s = O.points_in_sphere(n // 2 + 77) + 0.3 if case == "rectangular" else pts
    name = {0: "inv_delta", 1: "laplace", 2: "helmholtz"}[kind]

    def build():
        tcl = cluster_of(pts, leaf)
        scl = tcl if pts_s is pts else cluster_of(pts_s, leaf)
        if cplx:
            return Htool.ComplexHMatrixTreeBuilder(eps, eta, "N", "N", reqrank=reqrank).build(Htool.ComplexNativeGenerator(name, pts, pts_s, p0), tcl, scl), tcl, scl
        return Htool.HMatrixTreeBuilder(eps, eta, "N", "N", reqrank=reqrank).build(Htool.NativeGenerator(name, pts, pts_s, p0), tcl, scl), tcl, scl

    monkeypatch.setenv("HTOOL_ACA_STEP_MIN", "0")      # never
    H0, _, _ = build()
    smin = 120 if case == "inv_delta_eta2" else 300    # (eta = 2: few large admissible leaves)
    monkeypatch.setenv("HTOOL_ACA_STEP_MIN", str(smin))  # every leaf of more than that many rows or columns
    H1, tcl, scl = build()
    L0, L1 = np.asarray(H0.leaves()), np.asarray(H1.leaves())
    a, b = {tuple(l[:4]): int(l[4]) for l in L0}, {tuple(l[:4]): int(l[4]) for l in L1}
    assert set(a) == set(b)
    big = [k for k in a if max(k[1], k[3]) > smin and a[k] >= 0]
    assert len(big) >= 8, len(big)                        # the path was really taken
    d = np.array([a[k] - b[k] for k in big])
    assert np.mean(d != 0) <= 0.02 and np.abs(d).max() <= 1, (np.mean(d != 0), np.abs(d).max())
    if reqrank >= 0:
        assert {b[k] for k in big} == {reqrank}
    # factors: where the ranks agree, U V of the two builds agree (same pivots, other summation order)
    idx1 = {tuple(l[:4]): i for i, l in enumerate(L1)}
    idx0 = {tuple(l[:4]): i for i, l in enumerate(L0)}
    rng = np.random.RandomState(1)
    same = [k for k in big if a[k] == b[k] and a[k] > 0]
    for k in [same[i] for i in rng.choice(len(same), size=min(25, len(same)), replace=False)]:
        U0, V0 = H0.leaf_panels(idx0[k])
        U1, V1 = H1.leaf_panels(idx1[k])
        B0, B1 = np.asarray(U0) @ np.asarray(V0), np.asarray(U1) @ np.asarray(V1)
        assert np.linalg.norm(B0 - B1) <= 1e-9 * np.linalg.norm(B0), (k, np.linalg.norm(B0 - B1) / np.linalg.norm(B0))
    # against the CPU ACA (oracle) and the exact operator
    if case in ("laplace", "helmholtz", "inv_delta_eta2"):
        oc = O.Cluster(pts, max_leaf=leaf)
        OH = O.HMatrix(oc, oc, kind, p0, is_complex=cplx, eps=eps, eta=eta)
        theirs = {tuple(l[:4]): int(l[4]) for l in OH.leaves}
        assert set(theirs) == set(b)
        dd = np.array([b[k] - theirs[k] for k in big])
        assert np.mean(dd != 0) < 0.02 and np.abs(dd).max() <= 1
    x = rng.rand(pts_s.shape[1]) + (1j * rng.rand(pts_s.shape[1]) if cplx else 0)
    rows = np.arange(0, n, 7)
    ye = O.dense_matvec(kind, pts, pts_s, x, p0, rows=rows)
    if reqrank < 0:
        assert np.linalg.norm((H1 * x)[rows] - ye) / np.linalg.norm(ye) < eps
    assert np.linalg.norm(H1 * x - H0 * x) / np.linalg.norm(H0 * x) < (1e-2 if reqrank >= 0 else 2 * eps)


def test_aca_confirmation_steps(built, oracle):
    """htool_build_params.aca_confirm_steps / HMatrixTreeBuilder.set_aca_confirmation_steps (an extension, default 0 = the
    reference's stopping rule).  VERDICT round 2, weak spot 5a: on nearly collinear clouds (`shape = sheet` of tools/fuzz.py) the
    partially pivoted ACA passes its stopping test far too early -- shown with the independent explicit-residual ACA in
    profiles/r03_fuzz_sheet_case_61_322.txt -- and one confirming step repairs it.
    (1) sheet cloud: error of the product 40-60 epsilon without, < epsilon with one confirming step; ranks as the CPU oracle's.
    (2) ball (the reference's geometry): leaves whose rank is unchanged keep their factors BIT FOR BIT (the confirming terms are
        dropped); the others gained terms because a confirming step failed; storage grows by a few per cent, the error shrinks.
    (3) the callback-generator path (host ACA) follows the same rule; out-of-range values are refused."""
    import Htool
    from tests.helpers import NumpyGenerator, cluster_of

    O = oracle

    def build(pt, ps, kind, p0, eps, eta, leaf, confirm, strategy=None):
        cb = Htool.ClusterTreeBuilder()
        cb.set_maximal_leaf_size(leaf)
        if strategy is not None:
            cb.set_partitioning_strategy(strategy)
        tcl = cb.create_cluster_tree(pt, 2)
        scl = tcl if ps is pt else cb.create_cluster_tree(ps, 2)
        b = Htool.HMatrixTreeBuilder(eps, eta, "N", "N")
        b.set_aca_confirmation_steps(confirm)
        return b.build(Htool.NativeGenerator({0: "inv_delta", 1: "laplace"}[kind], pt, ps, p0), tcl, scl)

    def ranks(H):
        return {tuple(l[:4]): int(l[4]) for l in np.asarray(H.leaves() if hasattr(H, "leaves") and callable(H.leaves) else H.leaves)}

    # (1) sheet
    rng = np.random.RandomState(4)
    n = 40000
    pt = np.asfortranarray(rng.rand(2, n)); pt[-1] *= 1e-3
    ns = int(n * 0.55)
    ps = np.asfortranarray(rng.rand(2, ns)); ps[-1] *= 1e-3
    x = rng.rand(ns)
    rows = np.arange(0, n, 97)
    eps = 4e-6
    ye = O.dense_matvec(0, pt, ps, x, 0.1, rows=rows)
    errs = []
    for confirm in (0, 1):
        H = build(pt, ps, 0, 0.1, eps, 10.0, 33, confirm)
        errs.append(np.linalg.norm((H * x)[rows] - ye) / np.linalg.norm(ye) / eps)
        oc, ocs = O.Cluster(pt, max_leaf=33), O.Cluster(ps, max_leaf=33)
        OH = O.HMatrix(oc, ocs, 0, 0.1, eps=eps, eta=10.0, confirm=confirm)
        a, b = ranks(H), {tuple(l[:4]): int(l[4]) for l in OH.leaves}
        assert set(a) == set(b)
        d = np.array([a[k] - b[k] for k in a])
        assert np.mean(d != 0) < 0.02 and np.abs(d).max() <= 2, (confirm, np.mean(d != 0), np.abs(d).max())
    assert errs[0] > 10 and errs[1] < 1.0, errs
    # (2) ball
    np.random.seed(0)
    pts = O.points_in_sphere(20000)
    H0, H1 = build(pts, pts, 1, 0.0, 1e-4, 10.0, 64, 0), build(pts, pts, 1, 0.0, 1e-4, 10.0, 64, 1)
    L0, L1 = np.asarray(H0.leaves()), np.asarray(H1.leaves())
    assert np.array_equal(L0[:, :4], L1[:, :4])
    adm = L0[:, 4] >= 0
    grew = L1[adm, 4] - L0[adm, 4]
    assert grew.min() >= 0 and 0.02 < np.mean(grew > 0) < 0.4          # a confirming step failed on some leaves: they went on
    stored = lambda L: float((L[adm, 4] * (L[adm, 1] + L[adm, 3])).sum())  # noqa: E731
    assert 1.0 < stored(L1) / stored(L0) < 1.1
    same = np.flatnonzero(adm & (L0[:, 4] == L1[:, 4]) & (L0[:, 4] > 0))
    for i in np.random.RandomState(2).choice(same, size=40, replace=False):
        (U0, V0), (U1, V1) = H0.leaf_panels(int(i)), H1.leaf_panels(int(i))
        assert np.array_equal(np.asarray(U0), np.asarray(U1)) and np.array_equal(np.asarray(V0), np.asarray(V1))
    xb = np.random.rand(20000)
    rb = np.arange(0, 20000, 53)
    yb = O.dense_matvec(1, pts, pts, xb, 0.0, rows=rb)
    e0, e1 = (np.linalg.norm((H * xb)[rb] - yb) / np.linalg.norm(yb) for H in (H0, H1))
    assert e1 <= e0 * 1.05 and e0 < 1e-4
    # (3) callback generator: the host ACA applies the same rule (ranks of the CPU oracle with the same setting)
    np.random.seed(1)
    ph = O.points_in_sphere(1500)
    cl = cluster_of(ph, 20)
    for confirm in (0, 2):
        b = Htool.HMatrixTreeBuilder(1e-5, 10.0, "N", "N")
        b.set_aca_confirmation_steps(confirm)
        Hh = b.build(NumpyGenerator(ph, ph), cl, cl)
        OH = O.HMatrix(O.Cluster(ph, max_leaf=20), O.Cluster(ph, max_leaf=20), O.K_INV_DELTA, 0.1, eps=1e-5, eta=10.0, confirm=confirm)
        assert ranks(Hh) == {tuple(l[:4]): int(l[4]) for l in OH.leaves}
    with pytest.raises(RuntimeError):
        Htool.HMatrixTreeBuilder(1e-3, 10.0, "N", "N").set_aca_confirmation_steps(9)
