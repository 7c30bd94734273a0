"""CPU tests: the oracle (oracle/) against the golden fixtures and the reference's own assertions.

The fixtures (tests/golden/*.npz) hold the exact geometries of the reference's tests -- produced by
importing the reference's example/create_geometry.py, see tests/golden/make_golden.py -- and exact
dense products from the kernel definition (example/define_generators.py:14-17).  The bar is the one
the reference's tests set (tests/test_hmatrix.py:83, tests/test_distributed_operator.py:92):
|H x - A x| / |A x| < epsilon.
"""
import hashlib
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_geometry_generator_matches_reference_fingerprints(oracle):
    """Our re-implementation of the reference geometry draws the identical points
    (fingerprints recorded in SURVEY.md 8c from the reference's own script)."""
    g = np.load(os.path.join(GOLD, "hmatrix_500.npz"))
    t, s = oracle.random_geometries(3, 500, 500)
    assert np.array_equal(t, g["target"]) and np.array_equal(s, g["source"])
    assert hashlib.sha256(t.tobytes()).hexdigest().startswith("59c1e4eb97438e21")
    assert hashlib.sha256(s.tobytes()).hexdigest().startswith("51cd65409ca0714a")
    assert np.allclose(t[:, 0], [0.14127351, 0.74727739, -0.30321586])
    np.random.seed(0)
    assert np.array_equal(np.random.rand(500), g["x"])
    g2 = np.load(os.path.join(GOLD, "geometry_2d_500.npz"))
    t2, s2 = oracle.random_geometries(2, 500, 500)
    assert np.array_equal(t2, g2["target"]) and np.array_equal(s2, g2["source"])


def test_dense_oracle_matches_golden_products(oracle):
    g = np.load(os.path.join(GOLD, "hmatrix_500.npz"))
    y = oracle.dense_matvec(oracle.K_INV_DELTA, g["target"], g["source"], g["x"], 0.1)
    assert np.allclose(y, g["y"], rtol=1e-13, atol=0)
    Y = oracle.dense_matvec(oracle.K_INV_DELTA, g["target"], g["source"], g["X"], 0.1)
    assert np.allclose(Y, g["Y"], rtol=1e-13, atol=0)


@pytest.mark.parametrize("symmetric", [False, True])
def test_oracle_reference_hmatrix_case(oracle, symmetric):
    """tests/test_hmatrix.py:28-38: 500 x 500, d=3, eta=100, eps=1e-3, leaf 10, binary tree."""
    O = oracle
    g = np.load(os.path.join(GOLD, "hmatrix_500.npz"))
    T = g["target"]
    S = T if symmetric else g["source"]
    tc = O.Cluster(T, max_leaf=10)
    sc = tc if symmetric else O.Cluster(S, max_leaf=10)
    H = O.HMatrix(tc, sc, O.K_INV_DELTA, 0.1, eps=1e-3, eta=100.0)
    y = H.matvec(g["x"])
    y_exact = g["y_sym"] if symmetric else g["y"]
    assert np.linalg.norm(y - y_exact) / np.linalg.norm(y_exact) < 1e-3
    # product vs densified H-matrix (tests/test_hmatrix.py:84)
    D = H.to_dense()
    yd = np.zeros(500)
    yd[tc.perm] = D @ g["x"][sc.perm]
    assert np.linalg.norm(y - yd) / np.linalg.norm(yd) < 1e-10
    # leaves tile the matrix exactly once
    L = H.leaves.astype(np.int64)
    assert (L[:, 1] * L[:, 3]).sum() == 500 * 500
    cover = np.zeros((500, 500), dtype=np.int32)
    for t_off, m, s_off, n, _ in L:
        cover[t_off:t_off + m, s_off:s_off + n] += 1
    assert cover.min() == 1 and cover.max() == 1


@pytest.mark.parametrize("d", [2, 3])
@pytest.mark.parametrize("eps", [1e-3, 1e-6])
@pytest.mark.parametrize("shape", ["400x400", "400x200", "sym_L", "sym_U"])
def test_oracle_reference_distributed_cases(oracle, d, eps, shape):
    """tests/test_distributed_operator.py:9-26: eta=10, eps in {1e-3,1e-6}, d in {2,3}, 400x{400,200}, S/L, S/U."""
    O = oracle
    g = np.load(os.path.join(GOLD, f"distributed_400_d{d}.npz"))
    T = g["target"]
    if shape == "400x400":
        S, x, y = g["source400"], g["x400"], g["y400"]
    elif shape == "400x200":
        S, x, y = g["source200"], g["x200"], g["y200"]
    else:
        S, x, y = T, g["x400"], g["y_sym"]
    tc = O.Cluster(T, max_leaf=10)
    sc = tc if S is T else O.Cluster(S, max_leaf=10)
    kw = {}
    if shape.startswith("sym"):
        kw = {"symmetry": "S", "uplo": shape[-1]}
    H = O.HMatrix(tc, sc, O.K_INV_DELTA, 0.1, eps=eps, eta=10.0, **kw)
    yh = H.matvec(x)
    assert np.linalg.norm(yh - y) / np.linalg.norm(y) < eps
    if shape.startswith("sym"):
        # symmetric storage keeps one triangle only
        L = H.leaves
        if shape[-1] == "L":
            assert np.all(L[:, 2] <= L[:, 0])
        else:
            assert np.all(L[:, 0] <= L[:, 2])


def test_oracle_partitioned_rows(oracle):
    """example/use_distributed_operator.py geometry with 2 ranks: rows of partition p only (A.5)."""
    O = oracle
    g = np.load(os.path.join(GOLD, "partitioned_1000_w2.npz"))
    tc = O.Cluster(g["target"], size_of_partition=2, partition=g["partition"], partition_is_local=True, max_leaf=10)
    sc = O.Cluster(g["source"], max_leaf=10)
    y = np.zeros(1000)
    for p in range(2):
        H = O.HMatrix(tc, sc, O.K_INV_DELTA, 0.1, eps=1e-3, eta=10.0, target_partition=p)
        node = tc.partition_node(p)
        off, size = tc.inodes[node, 0], tc.inodes[node, 1]
        assert (off, size) == (500 * p, 500)
        assert np.all((H.leaves[:, 0] >= off) & (H.leaves[:, 0] + H.leaves[:, 1] <= off + size))
        y += H.matvec(g["x"])
    assert np.linalg.norm(y - g["y"]) / np.linalg.norm(g["y"]) < 1e-3


def test_oracle_aca_error_and_rejection(oracle):
    """Compressor contract (virtual_low_rank_generator.hpp:25-45): U m x r, V r x n with
    |A - U V|_F <~ eps |A|_F; blocks that are not worth it are rejected (r (m+n) > m n)."""
    O = oracle
    rng = np.random.RandomState(0)
    T = rng.rand(3, 300)
    S = rng.rand(3, 200) + np.array([[3.0], [0.0], [0.0]])
    tp, sp = np.ascontiguousarray(T.T), np.ascontiguousarray(S.T)
    rows, cols = np.arange(300, dtype=np.int32), np.arange(200, dtype=np.int32)
    A = O.kernel_block(O.K_INV_DELTA, T, S, 0.1)
    prev = 0
    for eps in (1e-2, 1e-4, 1e-8):
        U, V = O.aca(O.K_INV_DELTA, tp, sp, 0.1, rows, cols, eps)
        assert U.shape[0] == 300 and V.shape[1] == 200 and U.shape[1] == V.shape[0]
        assert np.linalg.norm(A - U @ V) / np.linalg.norm(A) < 5 * eps
        assert U.shape[1] >= prev
        prev = U.shape[1]
    # a near-field block is not compressible to 1e-12 at a worthwhile rank -> rejected
    S2 = rng.rand(3, 40)
    sp2 = np.ascontiguousarray(S2.T)
    assert O.aca(O.K_INV_DELTA, tp[:40].copy(), sp2, 1e-3, np.arange(40, dtype=np.int32), np.arange(40, dtype=np.int32), 1e-12) is None
    # complex (Helmholtz)
    U, V = O.aca(O.K_HELMHOLTZ, tp, sp, 4.0, rows, cols, 1e-6, is_complex=True)
    Ac = O.kernel_block(O.K_HELMHOLTZ, T, S, 4.0)
    assert np.linalg.norm(Ac - U @ V) / np.linalg.norm(Ac) < 5e-6
