"""GPU tests of the boundary rows that are not products or builds proper (b):
LowRankMatrix / recompression(hmatrix, fn), borrowed compressor factors, builds on partition Cluster objects, minimal
depths, failed deep copies, concurrent products from several host threads, damaged checkpoints."""
import copy
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _operator(n=3000, leaf=32, eps=1e-4, world=1, kind="laplace"):
    import Htool
    from oracle import oracle as O

    np.random.seed(0)
    pts = O.points_in_sphere(n)
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(leaf)
    cl = b.create_cluster_tree(pts, 2, size_of_partition=world)
    gen = Htool.NativeGenerator(kind, pts, pts)
    return pts, cl, gen, Htool.HMatrixTreeBuilder(eps, 10.0, "N", "N")


def test_low_rank_matrix_and_recompression_with_function(built, oracle):
    """Htool.recompression(hmatrix, fn) / openmp_recompression(hmatrix, fn) (src/htool/hmatrix/hmatrix.hpp:96,98) hand every
    low-rank leaf to fn as a LowRankMatrix with nb_rows / nb_cols / rank (lrmat.hpp:15-17)."""
    import Htool

    pts, cl, gen, builder = _operator()
    H = builder.build(gen, cl, cl)
    L = np.asarray(H.leaves())
    x = np.random.rand(len(pts[0]))
    y0 = H * x
    seen = []
    n = Htool.recompression(H, lambda lr: seen.append((lr.nb_rows(), lr.nb_cols(), lr.rank())))
    lowrank = sorted((int(l[1]), int(l[3]), int(l[4])) for l in L if l[4] >= 0)
    assert n == len(lowrank) and sorted(seen) == lowrank
    assert isinstance(seen, list) and np.array_equal(H * x, y0)  # a visit: nothing was recompressed
    seen2 = []
    Htool.openmp_recompression(H, lambda lr: seen2.append(lr.rank()))
    assert sorted(seen2) == sorted(r for _, _, r in lowrank)
    # the built-in rule (no function) does recompress
    assert Htool.recompression(H) > 0
    assert np.linalg.norm(H * x - y0) / np.linalg.norm(y0) < 1e-4


@pytest.mark.parametrize("allow_copy", [True, False])
def test_custom_compressor_copy_or_borrow(built, oracle, allow_copy):
    """VirtualLowRankGenerator(allow_copy) (virtual_low_rank_generator.hpp:33-42): with allow_copy=False the factors are
    borrowed from the Python arrays until the build has shipped them to HBM (htool_build_params.compress_borrows); the
    operator is the same either way, and clear_data() afterwards is harmless."""
    import Htool
    from tests.helpers import CustomSVD, NumpyGenerator, cluster_of

    O = oracle
    np.random.seed(0)
    T, S = np.random.random((3, 500)), np.random.random((3, 400)) + np.array([[1.2], [0.0], [0.0]])
    tcl, scl = cluster_of(T, 10), cluster_of(S, 10)
    gen = NumpyGenerator(T, S)
    lr = CustomSVD(gen, allow_copy)
    H = Htool.HMatrixTreeBuilder(1e-4, 100.0, "N", "N", low_rank_strategy=lr).build(gen, tcl, scl)
    x = np.random.rand(400)
    ye = gen.mat_vec(x)
    assert np.linalg.norm(H * x - ye) / np.linalg.norm(ye) < 1e-4
    assert (np.asarray(H.leaves())[:, 4] > 0).sum() > 20
    lr.clear_data()
    assert np.linalg.norm(H * x - ye) / np.linalg.norm(ye) < 1e-4
    Href = Htool.HMatrixTreeBuilder(1e-4, 100.0, "N", "N", low_rank_strategy=CustomSVD(gen, True)).build(gen, tcl, scl)
    assert np.array_equal(H * x, Href * x)


def test_build_on_partition_cluster_objects(built, oracle):
    """The reference builds on the Cluster object it is given: a partition sub-cluster as target means that partition's rows."""
    pts, cl, gen, builder = _operator(world=3)
    x = np.random.rand(len(pts[0]))
    for p in range(3):
        sub = cl.get_cluster_on_partition(p)
        H = builder.build(gen, sub, cl)
        Href = builder.build(gen, cl, cl, p)
        assert H.shape == Href.shape == (sub.get_size(), len(x))
        assert np.array_equal(H * x, Href * x)
    with pytest.raises(RuntimeError, match="disagree"):
        builder.build(gen, cl.get_cluster_on_partition(1), cl, 2)
    # a (partition x partition) block through the sub-cluster objects = build_local
    Hb = builder.build(gen, cl.get_cluster_on_partition(1), cl.get_cluster_on_partition(1))
    Hl = builder.build_local(gen, cl, cl, 1, 1)
    xs = np.random.rand(cl.get_cluster_on_partition(1).get_size())
    assert Hb.shape == Hl.shape and np.array_equal(Hb * xs, Hl * xs)


def test_minimal_depths_on_the_device_build(built, oracle):
    import Htool

    pts, cl, gen, _ = _operator(n=4000, leaf=20)
    n = 4000
    b = Htool.HMatrixTreeBuilder(1e-4, 100.0, "N", "N")
    b.set_minimal_target_depth(4)
    b.set_minimal_source_depth(5)
    b.set_block_tree_consistency(False)  # accepted, documented as not changing the leaves
    H = b.build(gen, cl, cl)
    ints, _ = cl._nodes()
    depth = {(r[0], r[1]): r[2] for r in ints}
    L = np.asarray(H.leaves())
    for t_off, m, s_off, nn, r in L[L[:, 4] >= 0]:
        assert depth[(t_off, m)] >= 4 and depth[(s_off, nn)] >= 5
    x = np.random.rand(n)
    ye = oracle.dense_matvec(oracle.K_LAPLACE, pts, pts, x)
    assert np.linalg.norm(H * x - ye) / np.linalg.norm(ye) < 1e-4
    H0 = Htool.HMatrixTreeBuilder(1e-4, 100.0, "N", "N").build(gen, cl, cl)
    assert len(H0.leaves()) < len(L)


def test_failed_deepcopy_leaves_the_source_intact(built, oracle, monkeypatch):
    """ADVICE (device_clone): a deep copy that runs out of device memory half-way must not free or alias the source's buffers."""
    pts, cl, gen, builder = _operator()
    H = builder.build(gen, cl, cl)
    x = np.random.rand(len(pts[0]))
    y0 = H * x
    for allowed in (0, 3, 9, 20):
        monkeypatch.setenv("HTOOL_TEST_FAIL_ALLOC_AFTER", str(allowed))
        with pytest.raises(RuntimeError):
            copy.deepcopy(H)
        monkeypatch.delenv("HTOOL_TEST_FAIL_ALLOC_AFTER")
        assert np.array_equal(H * x, y0)
    C = copy.deepcopy(H)
    assert np.array_equal(C * x, y0)
    del H
    assert np.array_equal(C * x, y0)


def test_products_from_several_host_threads(built, oracle):
    """`H * x` releases the GIL; products of ONE handle share its workspace and are serialised inside the library."""
    pts, cl, gen, builder = _operator(n=6000, leaf=50)
    H = builder.build(gen, cl, cl)
    n = 6000
    rng = np.random.RandomState(0)
    xs = [rng.rand(n) for _ in range(6)]
    Xs = [np.asfortranarray(rng.rand(n, 3)) for _ in range(3)]
    expect = [H * x for x in xs] + [H @ X for X in Xs]
    got = [None] * len(expect)

    def work(i):
        for _ in range(20):
            got[i] = H * xs[i] if i < len(xs) else H @ Xs[i - len(xs)]

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(expect))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for a, b in zip(got, expect):
        assert np.array_equal(a, b)


def test_damaged_checkpoints_are_refused(built, oracle, tmp_path):
    """ADVICE (build_from_leaves): a leaf table with a missing or repeated leaf, or a one-triangle file loaded where one-triangle
    storage is not possible, must fail instead of giving a wrong operator."""
    import Htool

    pts, cl, gen, builder = _operator(n=2000, leaf=16)
    H = builder.build(gen, cl, cl)
    path = str(tmp_path / "h.npz")
    Htool.save_hmatrix(path, H)
    x = np.random.rand(2000)
    assert np.linalg.norm(Htool.load_hmatrix(path, cl) * x - H * x) <= 1e-12 * np.linalg.norm(H * x)  # (same panels, other column order)
    f = dict(np.load(path))
    for name, edit in (("missing", lambda L: L[:-1]), ("repeated", lambda L: np.vstack([L[:-1], L[:1]]))):
        g = dict(f)
        g["leaves"] = edit(f["leaves"])
        g["offsets"] = f["offsets"][:-1] if name == "missing" else np.vstack([f["offsets"][:-1], f["offsets"][:1]])
        g["n_leaves"] = len(g["leaves"])
        p2 = str(tmp_path / (name + ".npz"))
        np.savez(p2, **g)
        with pytest.raises(RuntimeError, match="leaf table|leaves cover"):
            Htool.load_hmatrix(p2, cl)
    g = dict(f)
    g["n_leaves"] = len(f["leaves"]) + 5
    np.savez(str(tmp_path / "trunc.npz"), **g)
    with pytest.raises(RuntimeError, match="truncated"):
        Htool.load_hmatrix(str(tmp_path / "trunc.npz"), cl)
    # one triangle of a symmetric operator, loaded where one-triangle storage is not possible: refused
    Hs = Htool.HMatrixTreeBuilder(1e-4, 10.0, "S", "L").build(gen, cl, cl)
    assert Hs.is_one_triangle()
    ps = str(tmp_path / "sym.npz")
    Htool.save_hmatrix(ps, Hs)
    assert np.linalg.norm(Htool.load_hmatrix(ps, cl) * x - Hs * x) <= 1e-12 * np.linalg.norm(Hs * x)
    with pytest.raises(RuntimeError, match="ONE triangle"):
        Htool.load_hmatrix(ps, cl, target_partition_number=0)  # (a build restricted to a partition cannot store one triangle)
    b2 = Htool.ClusterTreeBuilder()
    b2.set_maximal_leaf_size(16)
    other = b2.create_cluster_tree(pts[:, ::-1].copy(), 2)
    with pytest.raises(RuntimeError, match="permutations differ"):
        Htool.load_hmatrix(ps, cl, other)


def test_graph_replay_then_other_buffers(built, oracle):
    """the hipGraph of a repeated product (same buffers, same caller-named stream) is replayed; a product on OTHER
    buffers issued right behind replays that may still be in flight (call pattern A, A, A, A, B, A, A, C ...) must neither disturb
    them nor be disturbed: every result equals the eager product of the same operator, bit for bit."""
    import torch

    import Htool

    O = oracle
    n = 20000
    np.random.seed(4)
    pts = O.points_in_sphere(n)
    cb = Htool.ClusterTreeBuilder()
    cb.set_maximal_leaf_size(64)
    cl = cb.create_cluster_tree(pts, 2)
    H = Htool.HMatrixTreeBuilder(1e-4, 10.0, "N", "N").build(Htool.NativeGenerator("laplace", pts, pts), cl, cl)
    torch.cuda.set_stream(torch.cuda.Stream())
    st = torch.cuda.current_stream().cuda_stream
    xs = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    ys = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    # eager references on the operator's own stream (stream 0: never captured)
    refs = []
    for x in xs:
        r = torch.zeros(n, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        H.matvec_device(x.data_ptr(), r.data_ptr(), 0, 0)
        torch.cuda.synchronize()
        refs.append(r.clone())
    for _round in range(3):
        for which in (0, 0, 0, 0, 1, 0, 0, 2, 2, 2, 1):  # second identical call captures, later ones replay; then the key changes at once
            H.matvec_device(xs[which].data_ptr(), ys[which].data_ptr(), 0, st)
        # several right-hand sides right behind replays of the single-vector graph (another workspace size: the graph is dropped)
        X = torch.stack(xs)
        Y = torch.zeros(3, n, dtype=torch.float64, device="cuda")
        H.matmat_device(X.data_ptr(), n, Y.data_ptr(), n, 3, 0, st)
        H.matvec_device(xs[0].data_ptr(), ys[0].data_ptr(), 0, st)
        torch.cuda.synchronize()
        for i in range(3):
            assert torch.equal(ys[i], refs[i]) and torch.equal(Y[i], refs[i])
            ys[i].zero_()
