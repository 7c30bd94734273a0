"""The device-resident native build (csrc/device_build2.inc: block tree, queues, rounds, re-splits and layout on the GPU) against
the host-driven driver (csrc/device_build.inc, HTOOL_BUILD=host) and the oracle: same leaves, same ranks, same factors, same
products.  Reference entry: src/htool/hmatrix/hmatrix_tree_builder.hpp:36 (block-tree rule SURVEY.md A.3, ACA A.4)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(monkeypatch, where, pts_t, pts_s, leaf, eps, eta, kind="laplace", sym="N", uplo="N", part=-1, nparts=1, one_triangle=True, param=0.0, mins=(0, 0)):
    import Htool

    if where == "host":
        monkeypatch.setenv("HTOOL_BUILD", "host")
    else:
        monkeypatch.delenv("HTOOL_BUILD", raising=False)
    cb = Htool.ClusterTreeBuilder()
    cb.set_maximal_leaf_size(leaf)
    ct = cb.create_cluster_tree(pts_t, 2, size_of_partition=nparts)
    cs = ct if pts_s is pts_t else cb.create_cluster_tree(pts_s, 2, size_of_partition=1)
    cplx = kind == "helmholtz"
    gen = (Htool.ComplexNativeGenerator if cplx else Htool.NativeGenerator)(kind, pts_t, pts_s, param)
    b = (Htool.ComplexHMatrixTreeBuilder if cplx else Htool.HMatrixTreeBuilder)(eps, eta, sym, uplo)
    b.set_symmetric_storage(one_triangle)
    b.set_minimal_target_depth(mins[0])
    b.set_minimal_source_depth(mins[1])
    H = b.build(gen, ct, cs, part)
    monkeypatch.delenv("HTOOL_BUILD", raising=False)
    return H, ct, cs


def _leaf_table(H):
    lv = np.asarray(H.leaves())
    order = np.lexsort((lv[:, 2], lv[:, 0], lv[:, 3], lv[:, 1]))
    return lv[order], order


CASES = [
    dict(n=3000, leaf=10, eps=1e-3, eta=10.0),                       # the reference's default leaf size: many re-split leaves
    dict(n=20000, leaf=50, eps=1e-4, eta=10.0),
    dict(n=20000, leaf=10, eps=1e-3, eta=3.0),
    dict(n=6000, leaf=25, eps=1e-5, eta=10.0, sym="S", uplo="L"),    # one-triangle storage
    dict(n=6000, leaf=25, eps=1e-5, eta=10.0, sym="S", uplo="U"),
    dict(n=8000, leaf=40, eps=1e-3, eta=10.0, kind="helmholtz", param=5.0),
    dict(n=9000, leaf=30, eps=1e-4, eta=10.0, part=1, nparts=3),     # the rows of one partition member
    dict(n=5000, leaf=20, eps=1e-3, eta=10.0, mins=(3, 2)),          # minimal depths
    dict(n=700, leaf=100, eps=1e-3, eta=10.0),                       # tiny: (nearly) everything dense
    dict(n=20000, leaf=50, eps=1e-4, eta=10.0, arena_mb=30),         # a forced small arena: several rounds, a batch per round
    dict(n=12000, leaf=10, eps=1e-3, eta=10.0, arena_mb=8, kind="helmholtz", param=3.0),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(v) for v in c.values()))
def test_device_resident_build_equals_host_driven_build(built, oracle, monkeypatch, case):
    from htool_python_amd.workloads import points_in_sphere

    c = dict(kind="laplace", sym="N", uplo="N", part=-1, nparts=1, param=0.0, mins=(0, 0), arena_mb=0)
    c.update(case)
    if c["arena_mb"]:
        monkeypatch.setenv("HTOOL_BUILD_ARENA_MB", str(c["arena_mb"]))
    pts = points_in_sphere(c["n"], seed=3)
    kw = dict(kind=c["kind"], sym=c["sym"], uplo=c["uplo"], part=c["part"], nparts=c["nparts"], param=c["param"], mins=c["mins"])
    Hd, ct, cs = _build(monkeypatch, "device", pts, pts, c["leaf"], c["eps"], c["eta"], **kw)
    Hh, _, _ = _build(monkeypatch, "host", pts, pts, c["leaf"], c["eps"], c["eta"], **kw)
    if c["arena_mb"]:
        assert int(Hd.get_local_information()["Number_of_batches"]) > 2, "the forced arena was meant to take several rounds"
    assert Hd.is_one_triangle() == Hh.is_one_triangle() == (c["sym"] == "S")
    ld, od = _leaf_table(Hd)
    lh, oh = _leaf_table(Hh)
    # the same leaves with the same ranks: the two drivers run the same kernels on the same blocks
    assert ld.shape == lh.shape and np.array_equal(ld, lh)
    # factors of sampled leaves bit for bit
    rng = np.random.RandomState(0)
    for k in rng.choice(len(ld), min(60, len(ld)), replace=False):
        a, b = Hd.leaf_panels(int(od[k])), Hh.leaf_panels(int(oh[k]))
        if a is None or b is None:
            assert a is None and b is None
            continue
        for p, q in zip(a if isinstance(a, tuple) else (a,), b if isinstance(b, tuple) else (b,)):
            assert np.array_equal(np.asarray(p), np.asarray(q))
    # products: same panels, another column order inside the tiles (the queues have another order): agreement to rounding
    nr, nc = Hd.shape
    cplx = c["kind"] == "helmholtz"
    x = rng.rand(nc) + (1j * rng.rand(nc) if cplx else 0)
    yd, yh = Hd * x, Hh * x
    assert np.linalg.norm(yd - yh) <= 1e-13 * np.linalg.norm(yh)
    assert np.array_equal(Hd * x, yd)
    X = np.asfortranarray(rng.rand(nc, 16) + (1j * rng.rand(nc, 16) if cplx else 0))
    assert np.linalg.norm(np.asarray(Hd @ X) - np.asarray(Hh @ X)) <= 1e-13 * np.linalg.norm(np.asarray(Hh @ X))
    if c["part"] < 0:
        w = rng.rand(nr) + (1j * rng.rand(nr) if cplx else 0)
        zd, zh = Hd.transposed_mul(w), Hh.transposed_mul(w)
        assert np.linalg.norm(zd - zh) <= 1e-13 * np.linalg.norm(zh)
    # and against the exact operator (the reference's bar: tests/test_hmatrix.py:83)
    if c["part"] < 0:
        kind = {"laplace": oracle.K_LAPLACE, "helmholtz": oracle.K_HELMHOLTZ}[c["kind"]]
        rows = rng.choice(nr, 100, replace=False)
        ye = oracle.dense_matvec(kind, pts, pts, x, c["param"], rows=rows)
        assert np.linalg.norm(yd[rows] - ye) / np.linalg.norm(ye) < c["eps"]


def test_device_resident_queues_equal_the_oracle_block_tree(built, oracle, monkeypatch):
    """The two work queues of the device-resident build are, as sets, the oracle's block tree (and the host's, which
    tests/test_host_logic.py compares with the oracle on the CPU)."""
    import Htool
    from htool_python_amd.workloads import points_in_sphere

    pts = points_in_sphere(4000, seed=9)
    for leaf, eta in ((10, 10.0), (30, 2.0)):
        H, ct, cs = _build(monkeypatch, "device", pts, pts, leaf, 1e-3, eta)
        adm, dns = Htool.block_tree_queues(ct, cs, eta)
        lv = np.asarray(H.leaves())
        got = set(map(tuple, lv[:, :4].tolist()))
        # every leaf of the build is a leaf of the block tree or lies inside an admissible one that was re-split
        tree = set(map(tuple, np.asarray(adm)[:, :4].tolist())) | set(map(tuple, np.asarray(dns)[:, :4].tolist()))
        inside = 0
        for t in got - tree:
            assert any(a[0] <= t[0] and t[0] + t[1] <= a[0] + a[1] and a[2] <= t[2] and t[2] + t[3] <= a[2] + a[3] for a in np.asarray(adm)[:, :4].tolist())
            inside += 1
        assert len(got & tree) > 0.5 * len(tree)
        assert sum(int(l[1]) * int(l[3]) for l in lv) == 4000 * 4000
