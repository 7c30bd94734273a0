"""Hierarchical LU on the device (htool_python_amd/csrc/hlu_device.hip) against the CPU checker (oracle/hlu_exec.cpp) that executes
the same plan, and against the dense solve of the same operator (the bar of the reference's tests/test_hmatrix.py:98-128)."""
import copy

import numpy as np
import pytest

import Htool
from oracle import hlu as ohlu

from .test_hlu_cpu import make_case

pytestmark = pytest.mark.gpu


def device_window(plan, host, w):
    """Window w executed by the device kernels on a copy of the CPU checker's arrays; returns that copy."""
    dev = copy.copy(host)
    for name in ("factor", "diag", "rank", "norm0", "norm2", "counters"):
        setattr(dev, name, getattr(host, name).copy())
    plan.debug_execute(w, w, dev.factor, dev.diag, dev.rank, dev.norm0, dev.norm2, dev.counters)
    return dev


def device_solve(plan, state, B, trans="N"):
    X = np.asfortranarray(np.array(B, dtype=np.float64, ndmin=2).T if np.ndim(B) == 1 else np.array(B, dtype=np.float64))
    X = np.asfortranarray(X)
    flat = X.ravel(order="F")
    assert np.shares_memory(flat, X)
    plan.debug_execute(-1 if trans == "N" else -2, 0, state.factor, state.diag, state.rank, state.norm0, state.norm2, state.counters, flat, X.shape[0], X.shape[1])
    return X[:, 0] if np.ndim(B) == 1 else X


@pytest.mark.parametrize("n,leaf,eta,children,sym", [(260, 40, 1e-4, 2, False), (900, 30, 10.0, 2, False), (700, 25, 10.0, 3, False), (1500, 60, 3.0, 2, False),
                                                     (260, 40, 1e-4, 2, True), (900, 30, 10.0, 2, True), (700, 25, 10.0, 3, True)])
def test_device_kernels_match_the_cpu_checker_window_by_window(n, leaf, eta, children, sym):
    """sym: the hierarchical Cholesky factorisation of the lower triangle (Params::symmetric)."""
    eps, eps_lu = 1e-3, 1e-4
    H, cl = make_case(n, leaf, eps, eta, children)
    ids = np.where(H.leaves[:, 0] >= H.leaves[:, 2])[0] if sym else np.arange(len(H.leaves))
    plan = Htool.HLUPlan(cl, H.leaves[ids], eps_lu, window_tasks=2000, symmetric=sym)
    host = ohlu.HostLU(plan, lambda i: H.leaf_data(int(ids[i])), eps_lu, run=False)
    n_win = host.info["windows"]
    n_leaves = host.n_leaves   # (the records behind the operator's leaves are the stage blocks of split update runs)
    lr = host.leaves["kind"][:n_leaves] == 1
    for w in range(n_win):
        dev = device_window(plan, host, w)   # the device starts every window from the CPU checker's state: kernels are compared, not histories
        host.run_window(w)
        same_rank = (dev.rank[: len(lr)][lr] == host.rank[: len(lr)][lr]).mean() if lr.any() else 1.0
        assert same_rank >= 0.9, (w, same_rank)
        worst_dense = worst_lr = 0.0
        for i in range(n_leaves):
            a, b = host.leaf_dense(i), dev.leaf_dense(i)
            err = np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-300)
            if lr[i]:
                worst_lr = max(worst_lr, err)
            else:
                worst_dense = max(worst_dense, err)
        # (a dense leaf can take an update through a low-rank leaf re-truncated in the same window: the two truncations agree to eps_lu, not to rounding)
        assert worst_dense < (1e-10 if not lr.any() else eps_lu), (w, worst_dense)
        assert worst_lr < 5 * eps_lu, (w, worst_lr)
        assert np.allclose(dev.diag, host.diag, rtol=1e-9, atol=1e-9 * np.abs(host.diag).max())
        assert dev.counters[4] == 0


@pytest.mark.parametrize("n,leaf,eta,sym", [(1000, 30, 10.0, False), (3000, 50, 10.0, False), (3000, 50, 10.0, True)])
def test_device_factorisation_solves_the_system(n, leaf, eta, sym):
    eps, eps_lu = 1e-3, 1e-4
    H, cl = make_case(n, leaf, eps, eta)
    ids = np.where(H.leaves[:, 0] >= H.leaves[:, 2])[0] if sym else np.arange(len(H.leaves))
    plan = Htool.HLUPlan(cl, H.leaves[ids], eps_lu, symmetric=sym)
    leaf_data = lambda i: H.leaf_data(int(ids[i]))  # noqa: E731
    st = ohlu.HostLU(plan, leaf_data, eps_lu, run=False)
    plan.debug_execute(0, st.info["windows"] - 1, st.factor, st.diag, st.rank, st.norm0, st.norm2, st.counters)
    plan.debug_execute(-3, 0, st.factor, st.diag, st.rank, st.norm0, st.norm2, st.counters)   # the explicit inverse factors of the small diagonal blocks
    assert st.counters[0] <= 0.02 * st.counters[1] and st.counters[4] == 0   # (truncations cut at a leaf's capacity: rare)
    A = H.to_dense()
    x_ref = np.ones(n)
    x = device_solve(plan, st, A @ x_ref)
    assert np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref) < eps
    B = np.random.default_rng(1).normal(size=(n, 11))
    Xd = np.linalg.solve(A, B)
    X = device_solve(plan, st, B)
    bar = 5 * eps_lu * max(1.0, np.sqrt(np.linalg.cond(A)) / 10)
    assert np.linalg.norm(X - Xd) / np.linalg.norm(Xd) < bar
    Xt = device_solve(plan, st, B, "T")
    assert np.linalg.norm(Xt - np.linalg.solve(A.T, B)) / np.linalg.norm(Xd) < bar
    # bitwise reproducible: the same plan on the same data gives the same factors
    st2 = ohlu.HostLU(plan, leaf_data, eps_lu, run=False)
    plan.debug_execute(0, st2.info["windows"] - 1, st2.factor, st2.diag, st2.rank, st2.norm0, st2.norm2, st2.counters)
    plan.debug_execute(-3, 0, st2.factor, st2.diag, st2.rank, st2.norm0, st2.norm2, st2.counters)
    assert np.array_equal(st.diag, st2.diag) and np.array_equal(st.ranks(), st2.ranks())
    assert np.array_equal(device_solve(plan, st2, B), X)


# ---- the product path: Htool.HMatrix.lu_factorization / lu_solve / cholesky_* on device-built operators ------------------------------
def _operator(n, leaf, eps, sym=("N", "N"), seed=3):
    from oracle import oracle as O
    from tests.helpers import cluster_of

    np.random.seed(seed)
    pts = O.points_in_sphere(n)
    cl = cluster_of(pts, leaf)
    H = Htool.HMatrixTreeBuilder(eps, 10.0, *sym).build(Htool.NativeGenerator("inv_delta", pts, pts, 0.1), cl, cl)
    return pts, cl, H


@pytest.mark.parametrize("n,leaf,sym", [(3000, 32, ("N", "N")), (12000, 100, ("N", "N")), (6000, 10, ("N", "N")), (5000, 50, ("S", "L")), (5000, 50, ("S", "U"))])
def test_lu_factorization_of_an_operator_is_hierarchical_and_solves(built, n, leaf, sym):
    """tests/test_hmatrix.py:98-128 on the HIP path: lu_solve recovers x_ref from H x_ref to epsilon; and against the dense solve of
    the operator's own dense copy ("against the dense LU at N <= 20 000")."""
    eps = 1e-3
    pts, cl, H = _operator(n, leaf, eps, sym)
    Hc = copy.deepcopy(H)
    Hc.lu_factorization()
    info = Hc.factorization_info()
    assert info["kind"] == "hierarchical" and info["truncations_at_capacity"] <= 0.02 * info["truncations"], info
    assert info["factor_bytes"] < 4 * 8 * n * n  # (hierarchical: nowhere near a dense copy... for n large enough)
    x_ref = np.ones(n)
    x = Hc.lu_solve("N", H * x_ref)
    assert np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref) < eps           # the reference's assertion
    X_ref = np.ones((n, 2))
    X = Hc.lu_solve("N", np.asfortranarray(H @ X_ref))
    assert np.linalg.norm(X - X_ref) / np.linalg.norm(X_ref) < eps
    perm = np.asarray(cl.get_permutation())
    A = np.asarray(H.to_dense_in_user_numbering())
    B = np.random.default_rng(0).normal(size=(n, 5))
    Xd = np.linalg.solve(A, B)
    bar = 1e-7  # (lu_solve refines against the operator's own product: it solves the H-matrix's system, not merely to the truncation tolerance)
    Xh = Hc.lu_solve("N", np.asfortranarray(B))
    assert np.linalg.norm(Xh - Xd) / np.linalg.norm(Xd) < bar
    Xt = Hc.lu_solve("T", np.asfortranarray(B))
    assert np.linalg.norm(Xt - np.linalg.solve(A.T, B)) / np.linalg.norm(Xd) < bar
    assert np.array_equal(Hc.lu_solve("N", np.asfortranarray(B)), Xh)          # bitwise repeatable
    # Cholesky surface of the reference (the same hierarchical factorisation underneath)
    Hd = copy.deepcopy(H)
    Hd.cholesky_factorization("L")
    assert Hd.factorization_info()["kind"] == "hierarchical"
    if sym[0] == "N":   # (the LU of the 'N' operator above against the Cholesky factorisation of its lower triangle: about half the tasks)
        assert Hd.factorization_info()["tasks"] < 0.7 * info["tasks"]
    xc = Hd.cholesky_solve("L", H * x_ref)
    assert np.linalg.norm(xc - x_ref) / np.linalg.norm(x_ref) < eps
    with pytest.raises(RuntimeError, match="lu_factorization first"):
        Hd.lu_solve("N", H * x_ref)
    # the operator itself is untouched by its factorisation
    assert np.array_equal(Hc * x_ref, H * x_ref)
    del perm


def test_shifted_factorisation_on_device_vectors(built):
    import torch

    n, eps = 8000, 1e-3
    pts, cl, H = _operator(n, 64, eps)
    H.lu_factorization_shifted(0.5)
    assert H.factorization_info()["kind"] == "hierarchical"
    perm = np.asarray(cl.get_permutation())
    A = np.asarray(H.to_dense()) + 0.5 * np.eye(n)     # cluster numbering
    X = np.random.default_rng(1).normal(size=(n, 3))
    Bt = torch.from_numpy(np.ascontiguousarray((A @ X).T)).cuda()   # (mu, n): row c = right-hand side c, cluster numbering
    H.factor_solve_device(1, "N", Bt.data_ptr(), n, 3, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.linalg.norm(Bt.cpu().numpy().T - X) / np.linalg.norm(X) < 20 * eps   # (the factors applied once: the truncation tolerance times a modest amplification)
    del perm


def test_fallbacks_are_still_there(built, monkeypatch):
    """Operators the hierarchical factorisation does not cover take the dense one: complex operators, tolerances below 1e-7; and
    HTOOL_FACTOR=hlu refuses instead."""
    n = 1500
    pts, cl, H = _operator(n, 32, 1e-9)
    H.lu_factorization()
    assert H.factorization_info()["kind"].startswith("dense")
    monkeypatch.setenv("HTOOL_FACTOR", "hlu")
    with pytest.raises(RuntimeError, match="1e-7"):
        H.lu_factorization()
    monkeypatch.setenv("HTOOL_FACTOR", "dense")
    pts, cl, H = _operator(n, 32, 1e-3)
    H.lu_factorization()
    assert H.factorization_info()["kind"].startswith("dense")
    x = H.lu_solve("N", H * np.ones(n))
    assert np.linalg.norm(x - 1) / np.sqrt(n) < 1e-9


def test_hierarchical_lu_of_the_per_gpu_block_of_c5(built, oracle):
    """the 62 500-unknown per-rank diagonal block of BASELINE config C5 (500 000 points on 8 GPUs),
    factorised hierarchically (round 3: a 31 GB dense copy, 2.5-4 s), and facto_one_level() on it.  The system is the one
    `bench.py --gmres` solves, (shift I + H) x = b with shift = 8e-3 |H|: the unshifted block has a condition
    number of ~1e6 (a kernel with an algebraic singularity, sampled 100 times more densely than its length scale 0.1), out of
    reach of ANY factorisation truncated at 1e-4 -- which is why the bench shifts it."""
    import time

    from htool_python_amd.solver import Solver
    from htool_python_amd.workloads import points_in_sphere

    n, world, p, eps = 500_000, 8, 3, 1e-3
    pts = points_in_sphere(n, seed=0)
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(100)
    cl = b.create_cluster_tree(pts, 2, size_of_partition=world)
    sub = cl.get_cluster_on_partition(p)
    size = sub.get_size()
    gen = Htool.NativeGenerator("inv_delta", pts, pts, 0.1)
    Hb = Htool.HMatrixTreeBuilder(eps, 10.0, "N", "N").build_local(gen, cl, cl, p, p)
    rng = np.random.RandomState(1)
    v = rng.rand(size)
    for _ in range(6):                      # |H| by power iterations, as the bench does
        w = Hb * v
        norm_h = np.linalg.norm(w) / np.linalg.norm(v)
        v = w / np.linalg.norm(w)
    shift = 8e-3 * norm_h
    x_ref = rng.rand(size)
    bb = Hb * x_ref + shift * x_ref
    t0 = time.time()
    Hb.lu_factorization_shifted(shift)
    t_fact = time.time() - t0
    info = Hb.factorization_info()
    print("H-LU of the 62 500 block: %.2f s" % t_fact, info)
    assert info["kind"] == "hierarchical" and info["truncations_at_capacity"] <= 0.02 * info["truncations"]
    assert info["factor_bytes"] < 0.35 * 8 * size * size   # (the dense copy is 31 GB)
    x = Hb.lu_solve("N", bb)
    assert np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref) < eps
    solver = Solver(hmatrix=Hb, block_diagonal_hmatrix=Hb, shift=shift)
    solver.set_hpddm_args("-hpddm_tol 1e-8 -hpddm_max_it 50 -hpddm_gmres_restart 20")
    solver.facto_one_level()
    xs = np.zeros(size)
    solver.solve(xs, bb)
    sinfo = solver.get_information()
    assert "hierarchical" in sinfo["Preconditioner"] and int(sinfo["Nb_it"]) <= 8, sinfo
    assert np.linalg.norm(xs - x_ref) / np.linalg.norm(x_ref) < 1e-6


def test_symmetric_operator_that_is_not_positive_definite_falls_back_to_the_lu(built):
    """lu_factorization of an 'S' operator tries the hierarchical Cholesky factorisation first; a negative shift makes the operator
    indefinite: the attempt reports it and the LU takes over."""
    n, eps = 3000, 1e-4
    pts, cl, H = _operator(n, 40, eps, ("S", "L"))
    A = np.asarray(H.to_dense_in_user_numbering())
    lam = np.linalg.eigvalsh(A)
    shift = -0.5 * (lam[-1] + lam[-2])                      # between the two largest eigenvalues: one positive, the others negative, far from singular
    H.lu_factorization_shifted(float(shift))
    assert H.factorization_info()["kind"] == "hierarchical"
    B = np.random.default_rng(2).normal(size=(n, 2))
    X = H.lu_solve("N", np.asfortranarray(B))
    Xd = np.linalg.solve(A + shift * np.eye(n), B)
    assert np.linalg.norm(X - Xd) / np.linalg.norm(Xd) < 1e-5
