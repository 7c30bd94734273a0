"""The cluster tree built on the GPU (csrc/cluster_device.hip) against the host builder (csrc/cluster.cpp) and the oracle:
permutation, node table, centres and radii bit for bit -- the tree a build sees must not depend on where it was made
(reference entry: src/htool/clustering/cluster_tree_builder.hpp:19-56; algorithm SURVEY.md A.2)."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STRATEGIES = ["PCARegular", "PCAGeometric", "BoundingBoxRegular", "BoundingBoxGeometric"]


def _cloud(n, dim, seed, shape="box"):
    rng = np.random.RandomState(seed)
    if shape == "box":
        p = rng.rand(dim, n) * np.array([1.0, 0.6, 0.3])[:dim, None]
    elif shape == "blobs":  # clustered: very uneven Geometric pieces
        centres = rng.rand(dim, 7)
        p = centres[:, rng.randint(0, 7, n)] + 0.02 * rng.randn(dim, n)
    elif shape == "grid":  # many EQUAL projections: the sort must be stable
        side = int(round(n ** (1.0 / dim))) + 1
        idx = rng.permutation(side ** dim)[:n]
        p = np.stack([(idx // side ** k) % side for k in range(dim)]).astype(float) / side
    return np.asfortranarray(p)


def _tree(where, pts, nchild, leaf, strategy, partition=None, local=False, size_of_partition=None, radii=None, weights=None):
    import Htool

    os.environ["HTOOL_CLUSTER_TREE"] = where
    try:
        b = Htool.ClusterTreeBuilder()
        b.set_maximal_leaf_size(leaf)
        b.set_partitioning_strategy(getattr(Htool, STRATEGIES[strategy])())
        if partition is None:
            cl = b.create_cluster_tree(pts, nchild, size_of_partition, radii=radii, weights=weights)
        elif local:
            cl = b.create_cluster_tree_from_local_partition(pts, nchild, size_of_partition, partition, radii=radii, weights=weights)
        else:
            cl = b.create_cluster_tree_from_global_partition(pts, nchild, size_of_partition, partition, radii=radii, weights=weights)
    finally:
        del os.environ["HTOOL_CLUSTER_TREE"]
    ints, dbl = cl._nodes()
    return np.asarray(cl.get_permutation()).copy(), np.asarray(ints).copy(), np.asarray(dbl).copy()


def _same(a, b):
    assert np.array_equal(a[0], b[0]), "permutation differs at %d positions" % int((a[0] != b[0]).sum())
    assert a[1].shape == b[1].shape and np.array_equal(a[1], b[1]), "node table differs"
    assert np.array_equal(a[2].view(np.int64), b[2].view(np.int64)), "centres / radii differ in %d entries" % int((a[2] != b[2]).sum())


@pytest.mark.parametrize("strategy", [0, 1, 2, 3])
@pytest.mark.parametrize("n,dim,nchild,leaf,parts,shape", [
    (3000, 3, 2, 10, 1, "box"),       # LDS sort only
    (3000, 2, 3, 7, 4, "blobs"),
    (1500, 1, 2, 1, 2, "box"),        # one dimension, leaves of one point
    (70000, 3, 2, 50, 1, "box"),      # radix sort on the top levels, blocked sums
    (70000, 3, 4, 10, 8, "blobs"),    # four children, partition of 8, uneven pieces
    (50000, 3, 2, 25, 3, "grid"),     # ties everywhere
    (300000, 3, 2, 100, 1, "box"),
])
def test_device_tree_equals_host_tree(built, strategy, n, dim, nchild, leaf, parts, shape):
    pts = _cloud(n, dim, 11 + strategy, shape)
    rng = np.random.RandomState(5)
    w = rng.rand(n) + 0.5 if n % 7 == 0 else None
    r = 0.01 * rng.rand(n) if n % 3 == 0 else None
    host = _tree("host", pts, nchild, leaf, strategy, size_of_partition=parts, radii=r, weights=w)
    dev = _tree("device", pts, nchild, leaf, strategy, size_of_partition=parts, radii=r, weights=w)
    _same(host, dev)


def test_device_tree_with_given_partitions(built):
    n = 60000
    pts = _cloud(n, 3, 3)
    rng = np.random.RandomState(9)
    glob = rng.randint(0, 5, n).astype(np.int32)
    for strategy in (0, 3):
        _same(_tree("host", pts, 2, 20, strategy, partition=glob, size_of_partition=5),
              _tree("device", pts, 2, 20, strategy, partition=glob, size_of_partition=5))
    sizes = np.array([10000, 25000, 5000, 20000])
    local = np.asfortranarray(np.stack([np.concatenate([[0], np.cumsum(sizes)[:-1]]), sizes]).astype(np.int32))
    _same(_tree("host", pts, 2, 20, 0, partition=local, local=True, size_of_partition=4),
          _tree("device", pts, 2, 20, 0, partition=local, local=True, size_of_partition=4))


def test_device_tree_equals_the_oracle_and_its_definition(built, oracle):
    """... and the oracle's tree (a serial restatement), plus the property check derived from the definitions."""
    from oracle import independent as I

    n = 120000
    pts = _cloud(n, 3, 21)
    w = np.random.RandomState(2).rand(n) + 0.5
    for strategy in (0, 1, 2, 3):
        perm, ints, dbl = _tree("device", pts, 2, 50, strategy, size_of_partition=4, weights=w)
        oc = oracle.Cluster(pts, n_children=2, size_of_partition=4, max_leaf=50, strategy=strategy, weights=w)
        assert np.array_equal(oc.perm, perm)
        mine = {(int(r[0]), int(r[1])): (int(r[2]), int(r[5]), int(r[6]), tuple(g)) for r, g in zip(ints, dbl)}
        theirs = {(int(r[0]), int(r[1])): (int(r[2]), int(r[5]), int(r[6]), tuple(g)) for r, g in zip(oc.inodes, oc.dnodes)}
        assert mine == theirs
        assert I.check_cluster_tree(ints, dbl, perm, pts, 2, 50, strategy, weights=w) > 100


def test_device_tree_of_a_million_points_is_fast_and_equal(built):
    """BASELINE configs[3] geometry size: 10^6 points, leaf 100 (the bench) and leaf 10 (the reference's default)."""
    import Htool

    n = 1000000
    pts = _cloud(n, 3, 1)
    for leaf in (100, 10):
        dev = _tree("device", pts, 2, leaf, 0)
        t0 = time.perf_counter()
        dev = _tree("device", pts, 2, leaf, 0)
        t_dev = time.perf_counter() - t0
        t0 = time.perf_counter()
        host = _tree("host", pts, 2, leaf, 0)
        t_host = time.perf_counter() - t0
        print("cluster tree 1M leaf %d: device %.4f s, host %.4f s (%d threads)" % (leaf, t_dev, t_host, os.cpu_count()))
        _same(host, dev)
        assert t_dev < 0.25
    Htool.release_workspace()
