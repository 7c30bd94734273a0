"""CPU tests: the product's host code (cluster tree) and the C++ oracle (cluster tree, ACA) against the INDEPENDENT
numpy formulation of oracle/independent.py -- properties derived from the definitions (numpy eigh principal axis,
explicit-residual ACA, SVD epsilon-rank), not an equality between two implementations of the same recurrence."""
import numpy as np
import pytest

from oracle import independent as I

STRATEGIES = {"PCARegular": 0, "PCAGeometric": 1, "BoundingBoxRegular": 2, "BoundingBoxGeometric": 3}


def _cloud(dim, n, seed, stretch=True):
    rng = np.random.RandomState(seed)
    pts = rng.rand(dim, n)
    if stretch:
        pts *= np.array([4.0, 1.0, 0.5])[:dim, None]
    return pts


@pytest.mark.parametrize("strategy_name", sorted(STRATEGIES))
@pytest.mark.parametrize("dim,children,world", [(3, 2, 1), (2, 2, 2), (3, 3, 3), (2, 4, 4), (3, 2, 8)])
def test_cluster_tree_properties_independent(built, oracle, strategy_name, dim, children, world):
    """Every node: centre = weighted mean, all points within the radius (attained), children tile the parent and are
    separated by hyperplanes orthogonal to numpy's principal axis / the longest bounding-box edge, equal counts or equal
    widths, leaf rule.  Checked for the product's tree AND the oracle's."""
    import Htool

    pts = _cloud(dim, 1500, 10 + dim + children)
    sid = STRATEGIES[strategy_name]
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(12)
    b.set_partitioning_strategy(getattr(Htool, strategy_name)())
    cl = b.create_cluster_tree(pts, children, size_of_partition=world)
    ints, dbl = cl._nodes()
    n_checked = I.check_cluster_tree(ints, dbl, np.asarray(cl.get_permutation()), pts, children, 12, sid)
    assert n_checked >= 5
    oc = oracle.Cluster(pts, n_children=children, size_of_partition=world, max_leaf=12, strategy=sid)
    assert I.check_cluster_tree(oc.inodes, oc.dnodes, oc.perm, pts, children, 12, sid) >= 5


def test_cluster_tree_properties_with_radii_weights_and_given_partitions(built, oracle):
    import Htool

    rng = np.random.RandomState(5)
    n = 900
    pts, radii, weights = rng.rand(3, n), rng.rand(n) * 0.05, rng.rand(n) + 0.5
    b = Htool.ClusterTreeBuilder()
    cl = b.create_cluster_tree(pts, 2, radii=radii, weights=weights)
    ints, dbl = cl._nodes()
    assert I.check_cluster_tree(ints, dbl, np.asarray(cl.get_permutation()), pts, 2, 10, 0, radii, weights) >= 20
    # user-given partitions (global labels / local ranges): depth-1 children are the user's sets, below them the rule applies
    labels = (pts[0] * 3).astype(np.int32)
    cg = b.create_cluster_tree_from_global_partition(pts, 2, 3, labels)
    ints, dbl = cg._nodes()
    perm = np.asarray(cg.get_permutation())
    assert I.check_cluster_tree(ints, dbl, perm, pts, 2, 10, 0, given_partition=True) >= 20
    for p in range(3):
        sub = cg.get_cluster_on_partition(p)
        assert np.all(labels[perm[sub.get_offset():sub.get_offset() + sub.get_size()]] == p)
    part = np.array([[0, 300, 600], [300, 300, 300]])
    cloc = b.create_cluster_tree_from_local_partition(pts, 2, 3, part)
    ints, dbl = cloc._nodes()
    perm = np.asarray(cloc.get_permutation())
    assert I.check_cluster_tree(ints, dbl, perm, pts, 2, 10, 0, given_partition=True) >= 20
    for p in range(3):
        sub = cloc.get_cluster_on_partition(p)
        assert sorted(perm[sub.get_offset():sub.get_offset() + sub.get_size()]) == list(range(300 * p, 300 * (p + 1)))


@pytest.mark.parametrize("n,leaf,eta,eps,kind,p0,complex_", [
    (3000, 10, 10.0, 1e-3, 0, 0.1, False),
    (5000, 64, 10.0, 1e-4, 1, 0.0, False),
    (4000, 50, 5.0, 1e-6, 1, 0.0, False),
    (3000, 32, 10.0, 1e-4, 2, 6.0, True),
])
def test_oracle_aca_against_explicit_residual_aca_and_svd(oracle, n, leaf, eta, eps, kind, p0, complex_):
    """The C++ oracle's ACA (implicit residual, running Frobenius estimate) against the explicit-residual formulation and
    the SVD: same rank (same pivots), same product where the ranks agree, rank <= SVD-rank(eps / 10) + 2, error vs eps."""
    O = oracle
    np.random.seed(0)
    pts = O.points_in_sphere(n)
    oc = O.Cluster(pts, max_leaf=leaf)
    adm, _ = O.blocktree(oc, oc, eta)
    rng = np.random.RandomState(1)
    tp = np.ascontiguousarray(pts.T)
    ok = [b for b in rng.permutation(len(adm)) if max(oc.inodes[adm[b][0], 1], oc.inodes[adm[b][1], 1]) <= 500][:200]
    assert len(ok) >= 100
    same, errs, over = 0, [], []
    for b in ok:
        t, s = adm[b]
        rows = oc.perm[oc.inodes[t, 0]:oc.inodes[t, 0] + oc.inodes[t, 1]]
        cols = oc.perm[oc.inodes[s, 0]:oc.inodes[s, 0] + oc.inodes[s, 1]]
        A = O.kernel_block(kind, pts[:, rows], pts[:, cols], p0)
        ref = I.aca_full_residual(A, eps)
        got = O.aca(kind, tp, tp, p0, rows, cols, eps, is_complex=complex_)
        assert (ref is None) == (got is None)
        if ref is None:
            same += 1
            continue
        same += ref[0].shape[1] == got[0].shape[1]
        assert abs(ref[0].shape[1] - got[0].shape[1]) <= 1
        if ref[0].shape[1] == got[0].shape[1]:
            assert np.linalg.norm(ref[0] @ ref[1] - got[0] @ got[1]) <= 1e-11 * np.linalg.norm(A)
        err, r, rs = I.leaf_quality(A, got[0], got[1], eps)
        errs.append(err / eps)
        over.append(r - rs)
    errs, over = np.array(errs), np.array(over)
    assert same >= 0.97 * len(ok)
    assert np.mean(over <= 2) >= 0.99 and over.max() <= 4
    assert np.mean(errs <= 3.0) >= 0.9 and errs.max() <= 10.0


def test_explicit_residual_aca_rejects_and_handles_null_rows():
    rng = np.random.RandomState(0)
    assert I.aca_full_residual(rng.rand(30, 30), 1e-8) is None  # a random block is not compressible: k (m + n) > m n
    A = np.outer(rng.rand(40), rng.rand(25))
    A[0, :] = 0.0  # the first pivot row is null: the next unused row is taken
    U, V = I.aca_full_residual(A, 1e-10)
    assert U.shape[1] == 1 and np.linalg.norm(A - U @ V) <= 1e-14 * np.linalg.norm(A)
    U, V = I.aca_full_residual(A, 1e-10, transpose_role=True)
    assert U.shape == (40, 1) and np.linalg.norm(A - U @ V) <= 1e-14 * np.linalg.norm(A)
    assert I.svd_rank(A, 1e-10) == 1


@pytest.mark.parametrize("eta,leaf,children,world,min_depths", [(10.0, 10, 2, 1, (0, 0)), (3.0, 20, 2, 3, (0, 0)), (0.7, 12, 3, 1, (0, 0)),
                                                               (100.0, 10, 4, 4, (0, 0)), (10.0, 10, 2, 1, (4, 0)), (10.0, 16, 2, 2, (0, 5))])
def test_block_tree_from_its_definition(built, oracle, eta, leaf, children, world, min_depths):
    """The two work queues of the block tree (replaces the recursive pointer tree of
    htool::HMatrixTreeBuilder::build, hmatrix_tree_builder.hpp:36) verified WITHOUT a top-down visit: every leaf is a pair of
    cluster nodes where the visit stops, is reachable from the root pair by walking up through pairs whose split rule produces
    it, and the leaves tile the (row partition x all columns) exactly.  Square and rectangular, every rank's rows of a
    partition, minimal depths; for the product's queues AND the C++ oracle's."""
    import Htool

    T, S = _cloud(3, 1800, 3), _cloud(3, 1100, 4)
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(leaf)
    tcl = b.create_cluster_tree(T, children, size_of_partition=world)
    for scl, spts in ((tcl, T), (b.create_cluster_tree(S, children, size_of_partition=1), S)):
        tn, sn = tcl._nodes(), scl._nodes()
        for p in range(world):
            t_root = 0 if world == 1 else tcl.get_cluster_on_partition(p)._node_id()
            adm, dns = Htool.block_tree_queues(tcl, scl, eta, min_target_depth=min_depths[0], min_source_depth=min_depths[1],
                                               target_partition_number=p if world > 1 else -1)
            na, nd = I.check_block_tree(adm, dns, tn, sn, eta, t_root=t_root, min_target_depth=min_depths[0], min_source_depth=min_depths[1])
            assert nd > 0 and (na > 0 or eta < 1.0)
    # the oracle's block tree through the same check
    if True:
        oc = oracle.Cluster(T, n_children=children, size_of_partition=world, max_leaf=leaf)
        for p in range(world):
            oadm, odns = oracle.blocktree(oc, oc, eta, min_t=min_depths[0], min_s=min_depths[1], target_partition=p if world > 1 else -1)
            quad = lambda pairs: np.array([(oc.inodes[t, 0], oc.inodes[t, 1], oc.inodes[s, 0], oc.inodes[s, 1]) for t, s in pairs]).reshape(-1, 4)  # noqa: E731
            t_root = 0 if world == 1 else int(np.flatnonzero((oc.inodes[:, 2] == 1) & (oc.inodes[:, 6] == p))[0])
            I.check_block_tree(quad(oadm), quad(odns), (oc.inodes, oc.dnodes), (oc.inodes, oc.dnodes), eta, t_root=t_root,
                               min_target_depth=min_depths[0], min_source_depth=min_depths[1])


def test_block_tree_check_sees_a_wrong_leaf(built):
    """The check is not vacuous: a leaf list with one admissible block split further, or one block replaced by its parent, fails."""
    import Htool

    T = _cloud(3, 900, 5)
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(10)
    cl = b.create_cluster_tree(T, 2)
    nodes = cl._nodes()
    adm, dns = (np.asarray(a).copy() for a in Htool.block_tree_queues(cl, cl, 10.0))
    I.check_block_tree(adm, dns, nodes, nodes, 10.0)
    ints = np.asarray(nodes[0])
    of = {(int(r[0]), int(r[1])): k for k, r in enumerate(ints)}
    # split the largest admissible block on its target side: still a tiling by node pairs, but not where the visit stops / reaches
    k = int(np.argmax(adm[:, 1] * adm[:, 3]))
    t = of[(int(adm[k, 0]), int(adm[k, 1]))]
    assert ints[t, 5] > 0
    kids = [ints[ints[t, 4] + c] for c in range(ints[t, 5])]
    bad = np.vstack([np.delete(adm, k, axis=0)] + [np.array([[c[0], c[1], adm[k, 2], adm[k, 3]]]) for c in kids])
    with pytest.raises(AssertionError):
        I.check_block_tree(bad, dns, nodes, nodes, 10.0)
    # with a wrong eta the admissible leaves are no stopping places (or dense ones should have been)
    with pytest.raises(AssertionError):
        I.check_block_tree(adm, dns, nodes, nodes, 2.0)
