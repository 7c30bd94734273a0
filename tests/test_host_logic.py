"""CPU tests of the product's host logic (no GPU, no compute calls): cluster tree, block-tree work
queues and tile partition through the C ABI / the Htool shim, against the oracle's restatement, plus
the reference's cluster-tree invariants (tests/test_cluster.py:8-34)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol(built):
    lib_path, _ = built
    lib = ctypes.CDLL(lib_path)
    header = open(os.path.join(ROOT, "include", "htool_mi355x.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(htool_[a-z0-9_]+)\s*\(", header))
    names -= {"htool_log_sink", "htool_copy_submatrix_fn", "htool_compress_fn", "htool_dense_blocks_fn"}
    assert len(names) > 40
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    lib.htool_last_error.restype = ctypes.c_char_p
    assert lib.htool_device_count() >= 0


def test_no_gpu_means_loud_failure(built):
    """The product path has no CPU fallback: without a device, build fails with a clear error."""
    import Htool

    if Htool.device_count() > 0:
        pytest.skip("a GPU is present")
    pts = np.random.RandomState(0).rand(3, 200)
    b = Htool.ClusterTreeBuilder()
    cl = b.create_cluster_tree(pts, 2)
    gen = Htool.NativeGenerator("laplace", pts, pts)
    with pytest.raises(RuntimeError, match="no HIP device"):
        Htool.HMatrixTreeBuilder(1e-3, 10.0, "N", "N").build(gen, cl, cl)


def test_logger_routes_to_python_logging(built, caplog):
    """tests/test_logger.py:6-8 + src/htool/misc/logger.hpp:13-33."""
    import logging

    import Htool

    with caplog.at_level(logging.DEBUG, logger="Htool"):
        Htool.test_logger()
    got = [(r.levelname, r.getMessage()) for r in caplog.records if r.name == "Htool"]
    assert got == [("CRITICAL", "Critical message"), ("ERROR", "Error message"), ("WARNING", "Warning message"),
                   ("DEBUG", "Debug message"), ("INFO", "Info message")]


def _make_cluster(points, children, partition_type, world, max_leaf=10, strategy=None):
    import Htool

    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(max_leaf)
    if strategy is not None:
        b.set_partitioning_strategy(strategy)
    n = points.shape[1]
    if partition_type == "None":
        return b.create_cluster_tree(points, children, size_of_partition=world), None, False
    local = n // world
    part = np.zeros((2, world), dtype=int)
    for i in range(world):
        part[0, i] = i * local
        part[1, i] = local if i < world - 1 else n - (world - 1) * local
    if partition_type == "Local":
        return b.create_cluster_tree_from_local_partition(points, children, world, part), part, True
    glob = np.zeros(n)  # float labels, force-cast by the binding (tests/conftest.py:173-178)
    for i in range(world):
        glob[part[0, i]:part[0, i] + part[1, i]] = i
    return b.create_cluster_tree_from_global_partition(points, children, world, glob), glob.astype(np.int32), False


@pytest.mark.parametrize("world", [1, 2, 3, 4])
@pytest.mark.parametrize("dimension,partition_type,children", [
    (2, "None", 2), (3, "None", 2), (2, "Local", 2), (3, "Local", 2), (2, "Global", 2), (3, "Global", 2),
    (2, "None", 3), (2, "None", 9), (2, "None", 10),
])
def test_cluster_tree_invariants_and_oracle_equality(built, oracle, dimension, partition_type, children, world):
    """Reference invariants (tests/test_cluster.py:33-34) for every 'rank count' it is run with
    (mpirun -np 1..4), and node-for-node equality with the oracle's tree."""
    O = oracle
    rng = np.random.RandomState(0)
    n = 500
    pts = rng.rand(dimension, n)
    if partition_type != "None":
        local = n // world
        for i in range(world):
            pts[0, i * local:(i + 1) * local if i < world - 1 else n] = i  # tests/conftest.py:111-129
    cl, part, is_local = _make_cluster(pts, children, partition_type, world)
    perm = np.asarray(cl.get_permutation())
    assert sorted(perm.tolist()) == list(range(n))
    total = sum(cl.get_cluster_on_partition(p).get_size() for p in range(world))
    assert total == len(perm) == len(np.asarray(cl.get_cluster_on_partition(0).get_permutation()))
    offs = [cl.get_cluster_on_partition(p).get_offset() for p in range(world)]
    assert offs == sorted(offs) and offs[0] == 0
    assert cl.get_maximal_leaf_size() == 10
    if partition_type != "None":
        for p in range(world):
            sub = cl.get_cluster_on_partition(p)
            idx = perm[sub.get_offset():sub.get_offset() + sub.get_size()]
            assert np.all(pts[0, idx] == p)
    # independent property check (numpy eigh split direction, centre / radius from their definitions; oracle/independent.py)
    from oracle import independent as I

    ints, dbl = cl._nodes()
    I.check_cluster_tree(ints, dbl, perm, pts, children, 10, 0, given_partition=partition_type != "None")
    # and node-for-node equality with the C++ oracle (a second implementation of the same recurrence)
    oc = O.Cluster(pts, n_children=children, size_of_partition=world, partition=part, partition_is_local=is_local, max_leaf=10)
    assert np.array_equal(oc.perm, perm)
    ints, dbl = cl._nodes()
    mine = {(r[0], r[1]): (r[2], r[5], r[6]) for r in ints}
    theirs = {(r[0], r[1]): (r[2], r[5], r[6]) for r in oc.inodes}
    assert mine == theirs
    geo_m = {(r[0], r[1]): tuple(d) for r, d in zip(ints, dbl)}
    geo_o = {(r[0], r[1]): tuple(d) for r, d in zip(oc.inodes, oc.dnodes)}
    for k in geo_m:
        assert np.allclose(geo_m[k], geo_o[k], rtol=1e-14, atol=1e-15)
    # leaves respect the minimum cluster size; parents are the union of their children
    for r in ints:
        if r[5] > 0:
            ch = ints[r[4]:r[4] + r[5]]
            assert ch[0, 0] == r[0] and ch[:, 1].sum() == r[1] and np.all(ch[:, 1] >= 10)


@pytest.mark.parametrize("strategy_name", ["PCARegular", "PCAGeometric", "BoundingBoxRegular", "BoundingBoxGeometric"])
def test_partitioning_strategies(built, oracle, strategy_name):
    import Htool

    O = oracle
    pts = np.random.RandomState(1).rand(3, 800) * np.array([[4.0], [1.0], [0.5]])
    cl, _, _ = _make_cluster(pts, 2, "None", 2, strategy=getattr(Htool, strategy_name)())
    sid = {"PCARegular": 0, "PCAGeometric": 1, "BoundingBoxRegular": 2, "BoundingBoxGeometric": 3}[strategy_name]
    oc = O.Cluster(pts, size_of_partition=2, strategy=sid)
    assert np.array_equal(oc.perm, np.asarray(cl.get_permutation()))
    assert sorted(np.asarray(cl.get_permutation()).tolist()) == list(range(800))
    from oracle import independent as I

    ints, dbl = cl._nodes()
    assert I.check_cluster_tree(ints, dbl, np.asarray(cl.get_permutation()), pts, 2, 10, sid) >= 10


def test_radii_and_weights(built, oracle):
    import Htool

    rng = np.random.RandomState(2)
    pts, radii, weights = rng.rand(3, 300), rng.rand(300) * 0.1, rng.rand(300) + 0.5
    b = Htool.ClusterTreeBuilder()
    cl = b.create_cluster_tree(pts, 2, radii=radii, weights=weights)
    oc = oracle.Cluster(pts, size_of_partition=2, radii=radii, weights=weights)
    assert np.array_equal(oc.perm, np.asarray(cl.get_permutation()))
    ints, dbl = cl._nodes()
    assert np.isclose(dbl[0, 3], oc.dnodes[0, 3])
    from oracle import independent as I

    assert I.check_cluster_tree(ints, dbl, np.asarray(cl.get_permutation()), pts, 2, 10, 0, radii, weights) >= 5


def test_wrong_partition_format_raises(built):
    import Htool

    pts = np.random.RandomState(0).rand(2, 100)
    b = Htool.ClusterTreeBuilder()
    with pytest.raises(RuntimeError, match="Wrong format for partition"):
        b.create_cluster_tree_from_global_partition(pts, 2, 2, np.zeros(99))
    with pytest.raises(RuntimeError, match="Wrong format for partition"):
        b.create_cluster_tree_from_local_partition(pts, 2, 2, np.zeros((3, 2)))


@pytest.mark.parametrize("eta", [0.5, 10.0, 100.0])
@pytest.mark.parametrize("case", ["square", "rect", "partition"])
def test_block_tree_queues_equal_oracle_and_tile_the_matrix(built, oracle, eta, case):
    """The two work queues cover every matrix entry exactly once, every admissible
    block satisfies the admissibility inequality, and both queues equal the oracle's."""
    import Htool

    O = oracle
    rng = np.random.RandomState(3)
    T = rng.rand(3, 700)
    S = T if case != "rect" else rng.rand(3, 450) + 0.3
    world = 2 if case == "partition" else 1
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(10)
    tcl = b.create_cluster_tree(T, 2, size_of_partition=world)
    scl = tcl if S is T else b.create_cluster_tree(S, 2, size_of_partition=world)
    otc = O.Cluster(T, size_of_partition=world, max_leaf=10)
    osc = otc if S is T else O.Cluster(S, size_of_partition=world, max_leaf=10)
    for p in ([-1] if world == 1 else [0, 1]):
        adm, dns = Htool.block_tree_queues(tcl, scl, eta, target_partition_number=p)
        oadm, odns = O.blocktree(otc, osc, eta, target_partition=p)
        key = lambda tree, ids: sorted((tree.inodes[i, 0], tree.inodes[i, 1]) for i in ids)  # noqa: E731
        to_set = lambda q: sorted(map(tuple, np.asarray(q)))  # noqa: E731
        o_adm = sorted((otc.inodes[t, 0], otc.inodes[t, 1], osc.inodes[s, 0], osc.inodes[s, 1]) for t, s in oadm)
        o_dns = sorted((otc.inodes[t, 0], otc.inodes[t, 1], osc.inodes[s, 0], osc.inodes[s, 1]) for t, s in odns)
        assert to_set(adm) == o_adm and to_set(dns) == o_dns
        sub = tcl.get_cluster_on_partition(p) if p >= 0 else tcl
        r0, nr = sub.get_offset(), sub.get_size()
        cover = np.zeros((nr, S.shape[1]), dtype=np.int32)
        for t_off, m, s_off, n in list(np.asarray(adm)) + list(np.asarray(dns)):
            cover[t_off - r0:t_off - r0 + m, s_off:s_off + n] += 1
        assert cover.min() == 1 and cover.max() == 1
        # admissibility of every low-rank candidate, from the node geometry
        ints, dbl = tcl._nodes()
        sints, sdbl = scl._nodes()
        gt = {(r[0], r[1]): d for r, d in zip(ints, dbl)}
        gs = {(r[0], r[1]): d for r, d in zip(sints, sdbl)}
        for t_off, m, s_off, n in np.asarray(adm):
            a, c = gt[(t_off, m)], gs[(s_off, n)]
            dist = np.linalg.norm(a[:3] - c[:3]) - a[3] - c[3]
            assert 2 * min(a[3], c[3]) < eta * max(0.0, dist)


@pytest.mark.parametrize("tile_max", [16, 64, 128])
def test_tiles_partition_rows_and_align_with_clusters(built, tile_max):
    """Row/column tiles (pieces of cluster leaves, <= tile_max) partition the index range, and every
    cluster node is a union of whole tiles -- the property the tile-major panel layout relies on."""
    import Htool

    pts = np.random.RandomState(4).rand(3, 3000)
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(40)
    cl = b.create_cluster_tree(pts, 2, size_of_partition=3)
    tiles = np.asarray(Htool.cluster_tiles(cl, -1, tile_max))
    assert tiles[0, 0] == 0 and np.all(tiles[1:, 0] == tiles[:-1, 0] + tiles[:-1, 1]) and tiles[-1].sum() == 3000
    assert tiles[:, 1].max() <= tile_max and tiles[:, 1].min() >= 1
    starts = set(tiles[:, 0].tolist()) | {3000}
    ints, _ = cl._nodes()
    for r in ints:
        assert r[0] in starts and r[0] + r[1] in starts
    sub = np.asarray(Htool.cluster_tiles(cl, 1, tile_max))
    p1 = cl.get_cluster_on_partition(1)
    assert sub[0, 0] == p1.get_offset() and sub[:, 1].sum() == p1.get_size()


@pytest.mark.parametrize("dimension", [2, 3])
def test_cluster_plot_runs(built, dimension):
    """tests/test_cluster.py:37-42: Htool.plot(ax, cluster, points, depth) for the root and for a partition."""
    import matplotlib

    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    import Htool

    pts = np.random.RandomState(5).rand(dimension, 500)
    b = Htool.ClusterTreeBuilder()
    cl = b.create_cluster_tree(pts, 2, size_of_partition=2)
    local = cl.get_cluster_on_partition(1)
    fig = plt.figure()
    kw = {"projection": "3d"} if dimension == 3 else {}
    axes = [fig.add_subplot(2, 2, i + 1, **kw) for i in range(4)]
    Htool.plot(axes[0], cl, pts, 1)
    Htool.plot(axes[1], cl, pts, 2)
    Htool.plot(axes[2], local, pts, 1)
    Htool.plot(axes[3], local, pts, 2)
    # depth-1 colouring of the root = the two partitions
    assert len(np.unique(np.asarray(axes[0].collections[0].get_facecolors()), axis=0)) == 2
    # the partition sub-tree only draws its own points
    n_local = axes[2].collections[0].get_offsets().shape[0] if dimension == 2 else len(axes[2].collections[0]._offsets3d[0])
    assert n_local == local.get_size()
    plt.close(fig)
    if dimension == 3:  # the reference's own test draws 3-D points on plain 2-D axes (tests/test_cluster.py:38-42)
        _, ax = plt.subplots(2, 2)
        Htool.plot(ax[0, 0], cl, pts, 1)
        Htool.plot(ax[1, 1], local, pts, 2)
        assert ax[0, 0].collections[0].get_offsets().shape[0] == 500
        plt.close("all")


def test_mpi4py_standin_world_of_one(built):
    import mpi4py

    comm = mpi4py.MPI.COMM_WORLD
    assert comm.size == comm.Get_size() == 1 and comm.rank == comm.Get_rank() == 0
    assert comm.allreduce(7, op=mpi4py.MPI.SUM) == 7
    send = np.arange(5, dtype=np.uint8)
    recv = np.zeros(5, dtype=np.uint8)
    comm._htool_allgatherv(send, recv, [5], [0])
    assert np.array_equal(send, recv)
    comm.Barrier()
    assert comm.bcast("x") == "x"


def test_block_tree_queues_one_triangle(oracle):
    import Htool
    from tests.helpers import cluster_of

    O = oracle
    np.random.seed(0)
    pts = O.points_in_sphere(4000)
    cl = cluster_of(pts, 25)
    oc = O.Cluster(pts, max_leaf=25)
    a, d = Htool.block_tree_queues(cl, cl, 10.0, symmetry="S", UPLO="L", one_triangle=True)
    oa, od = O.blocktree(oc, oc, 10.0, "S", "L")
    assert (len(a), len(d)) == (len(oa), len(od))
    a1, d1 = Htool.block_tree_queues(cl, cl, 10.0, symmetry="S", UPLO="L")
    assert (len(a1), len(d1)) == (len(a), len(d))   # ... which is the default, as in the reference
    a2, d2 = Htool.block_tree_queues(cl, cl, 10.0, symmetry="S", UPLO="L", one_triangle=False)
    assert len(a2) > len(a) and len(d2) > len(d)    # both triangles on request


def test_one_triangle_eligibility_of_separately_built_trees(built):
    """Two cluster trees built alike on the same points count as one tree for symmetric storage (the reference's
    example builds target and source clusters separately); a different tree on the same points does not."""
    import Htool

    pts = np.random.RandomState(3).rand(3, 3000)

    def tree(leaf):
        b = Htool.ClusterTreeBuilder()
        b.set_maximal_leaf_size(leaf)
        return b.create_cluster_tree(pts, 2, size_of_partition=1)

    t1, t2, other = tree(20), tree(20), tree(45)
    a_same, d_same = Htool.block_tree_queues(t1, t1, 10.0, symmetry="S", UPLO="L")
    a_twin, d_twin = Htool.block_tree_queues(t1, t2, 10.0, symmetry="S", UPLO="L")
    assert np.array_equal(a_same, a_twin) and np.array_equal(d_same, d_twin)
    a_full, d_full = Htool.block_tree_queues(t1, t1, 10.0, symmetry="S", UPLO="L", one_triangle=False)
    assert len(a_full) > len(a_same) and len(d_full) > len(d_same)
    a_other, d_other = Htool.block_tree_queues(t1, other, 10.0, symmetry="S", UPLO="L")
    a_plain, d_plain = Htool.block_tree_queues(t1, other, 10.0)
    assert np.array_equal(a_other, a_plain) and np.array_equal(d_other, d_plain)   # not eligible: every block is kept


def test_read_cluster_from_round_trip(built, tmp_path):
    """Htool.read_cluster_from (src/htool/clustering/utility.hpp:10; tests/conftest.py:446-449): a cluster tree written by
    save_cluster_to and read back is the same tree -- permutation, node table, partitions -- and usable as such."""
    import Htool

    pts = np.random.RandomState(4).rand(3, 700)
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(15)
    cl = b.create_cluster_tree(pts, 2, size_of_partition=3)
    props, tree = str(tmp_path / "c_cluster_tree_properties.csv"), str(tmp_path / "c_cluster_tree.csv")
    Htool.save_cluster_to(cl, props, tree)
    back = Htool.read_cluster_from(props, tree)
    assert np.array_equal(np.asarray(back.get_permutation()), np.asarray(cl.get_permutation()))
    (i0, d0), (i1, d1) = cl._nodes(), back._nodes()
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    assert back.get_size() == 700 and back.get_maximal_leaf_size() == 15
    for p in range(3):
        a, c = cl.get_cluster_on_partition(p), back.get_cluster_on_partition(p)
        assert (a.get_offset(), a.get_size()) == (c.get_offset(), c.get_size())
    # same block tree on the tree that was read
    q0, q1 = Htool.block_tree_queues(cl, cl, 10.0), Htool.block_tree_queues(back, back, 10.0)
    assert np.array_equal(q0[0], q1[0]) and np.array_equal(q0[1], q1[1])
    # damaged files are refused
    lines = open(tree).read().splitlines()
    open(tree, "w").write("\n".join(lines[:-3]) + "\n")
    with pytest.raises(RuntimeError):
        Htool.read_cluster_from(props, tree)
    rows = [ln.split(",") for ln in lines]
    rows[1][1] = str(int(rows[1][1]) + 1)  # a child that no longer tiles its parent
    open(tree, "w").write("\n".join(",".join(r) for r in rows) + "\n")
    with pytest.raises(RuntimeError, match="tile"):
        Htool.read_cluster_from(props, tree)
    # a file of another origin (e.g. upstream's save_cluster_tree output) is refused with a message that says what is supported
    open(props, "w").write("dimension,3\nsomething,else\n")
    with pytest.raises(RuntimeError, match="upstream htool"):
        Htool.read_cluster_from(props, tree)
    # nonsense parameters do not get past the table constructor
    open(tree, "w").write("\n".join(lines) + "\n")
    Htool.save_cluster_to(cl, props, str(tmp_path / "unused.csv"))
    text = open(props).read()
    for bad in (text.replace("maximal_leaf_size: 15", "maximal_leaf_size: 0"), text.replace("number_of_children: 2", "number_of_children: 1")):
        open(props, "w").write(bad)
        with pytest.raises(RuntimeError, match="maximal_leaf_size must be"):
            Htool.read_cluster_from(props, tree)


@pytest.mark.parametrize("min_t,min_s", [(0, 0), (4, 0), (0, 5), (6, 6)])
def test_minimal_depths_of_the_block_tree(built, min_t, min_s):
    """set_minimal_target_depth / set_minimal_source_depth (hmatrix_tree_builder.hpp:39-40): no block shallower than the
    minimal depths is taken as admissible; the queues still tile the matrix."""
    import Htool

    pts = np.random.RandomState(6).rand(3, 900)
    b = Htool.ClusterTreeBuilder()
    cl = b.create_cluster_tree(pts, 2)
    ints, _ = cl._nodes()
    depth = {(r[0], r[1]): r[2] for r in ints}
    adm, dns = Htool.block_tree_queues(cl, cl, 100.0, min_target_depth=min_t, min_source_depth=min_s)
    assert len(adm) > 0
    for t_off, m, s_off, n in np.asarray(adm):
        assert depth[(t_off, m)] >= min_t and depth[(s_off, n)] >= min_s
    a0, _ = Htool.block_tree_queues(cl, cl, 100.0)
    if min_t or min_s:  # with eta = 100 the unconstrained tree has admissible blocks above these depths
        assert min(min(depth[(t, m)], depth[(s, n)]) for t, m, s, n in np.asarray(a0)) < max(min_t, min_s)
    cover = np.zeros((900, 900), dtype=np.int8)
    for t_off, m, s_off, n in list(np.asarray(adm)) + list(np.asarray(dns)):
        cover[t_off:t_off + m, s_off:s_off + n] += 1
    assert cover.min() == 1 and cover.max() == 1


@pytest.mark.parametrize("strategy", [0, 1, 2, 3])
def test_large_cluster_tree_is_the_same_for_every_thread_count_and_equals_the_oracle(built, oracle, strategy):
    """Nodes of more than 4096 points sum their means and covariances in blocks (csrc/cluster.cpp: SUM_BLOCK) so that the top
    levels of the tree can use all threads: the permutation and the node table must not depend on the thread count, and equal
    the oracle's (which forms the same blocked sums serially).  Also the independent property check on the big tree."""
    import Htool
    from oracle import independent as I

    rng = np.random.RandomState(7)
    pts = rng.rand(3, 40000) * np.array([[1.0], [0.6], [0.3]])
    w = rng.rand(40000) + 0.5
    name = ["PCARegular", "PCAGeometric", "BoundingBoxRegular", "BoundingBoxGeometric"][strategy]
    trees = []
    for threads in (1, 3, 8):
        Htool.set_num_threads(threads)
        b = Htool.ClusterTreeBuilder()
        b.set_maximal_leaf_size(50)
        b.set_partitioning_strategy(getattr(Htool, name)())
        cl = b.create_cluster_tree(pts, 2, size_of_partition=4, weights=w)
        ints, dbl = cl._nodes()
        trees.append((np.asarray(cl.get_permutation()).copy(), np.asarray(ints).copy(), np.asarray(dbl).copy()))
    for t in trees[1:]:
        assert np.array_equal(t[0], trees[0][0]) and np.array_equal(t[1], trees[0][1]) and np.array_equal(t[2], trees[0][2])
    oc = oracle.Cluster(pts, n_children=2, size_of_partition=4, max_leaf=50, strategy=strategy, weights=w)
    assert np.array_equal(oc.perm, trees[0][0])
    # the node tables list the same nodes in another order (level by level here, depth first there): compare by (offset, size);
    # centres and radii bit for bit (the blocked sums are the same sums)
    mine = {(int(r[0]), int(r[1])): (int(r[2]), int(r[5]), int(r[6]), tuple(g)) for r, g in zip(trees[0][1], trees[0][2])}
    theirs = {(int(r[0]), int(r[1])): (int(r[2]), int(r[5]), int(r[6]), tuple(g)) for r, g in zip(oc.inodes, oc.dnodes)}
    assert mine == theirs
    assert I.check_cluster_tree(trees[0][1], trees[0][2], trees[0][0], pts, 2, 50, strategy, weights=w) > 100
    Htool.set_num_threads(8)


def test_python_surface_has_every_name_the_reference_module_declares(built):
    """Class / function names of the reference's pybind11 module on the hot path and its callers (src/htool/main.cpp:40-112 with
    the prefixes of local_operator.hpp:75-81, virtual_local_to_local_operator.hpp:92-95, distributed_operator/utility.hpp:15-17,
    solver/utility.hpp:11,46, solver/solver.hpp:17,69).  GenEO coarse-space classes (main.cpp:75-80,104-108) are out of scope."""
    import Htool

    real = ["Cluster", "VirtualPartitioning", "PCARegular", "PCAGeometric", "BoundingBoxRegular", "BoundingBoxGeometric", "ClusterTreeBuilder",
            "read_cluster_from", "VirtualGenerator", "IGenerator", "LowRankMatrix", "HMatrix", "VirtualLowRankGenerator", "VirtualDenseBlocksGenerator",
            "HMatrixTreeBuilder", "LocalRenumbering", "IGlobalToLocalOperator", "IRestrictedGlobalToLocalOperator", "RestrictedGlobalToLocalOperator",
            "ILocalToLocalOperator", "VirtualLocalToLocalOperator", "DistributedOperator", "CustomApproximationBuilder", "DefaultApproximationBuilder",
            "DefaultLocalApproximationBuilder", "Solver", "SolverDense", "DDMSolverBuilder", "DDMSolverWithDenseLocalSolver", "plot", "recompression",
            "openmp_recompression", "test_logger"]
    cplx = ["ComplexVirtualPartitioning", "ComplexLowRankMatrix", "ComplexHMatrix", "ComplexVirtualGenerator", "IComplexGenerator", "VirtualComplexLowRankGenerator",
            "ComplexVirtualDenseBlocksGenerator", "ComplexHMatrixTreeBuilder", "ComplexIGlobalToLocalOperator", "ComplexIRestrictedGlobalToLocalOperator",
            "ComplexRestrictedGlobalToLocalOperator", "ComplexILocalToLocalOperator", "ComplexVirtualLocalToLocalOperator", "ComplexDistributedOperator",
            "ComplexCustomApproximationBuilder", "ComplexDefaultApproximationBuilder", "ComplexDefaultLocalApproximationBuilder", "ComplexSolver",
            "ComplexSolverDense", "ComplexDDMSolverBuilder", "ComplexDDMSolverWithDenseLocalSolver"]
    missing = [n for n in real + cplx if not hasattr(Htool, n)]
    assert not missing, missing
