"""Hierarchical LU, CPU side: the plan (htool_python_amd/csrc/hlu_symbolic.cpp, host code of the product) executed by the CPU
checker oracle/hlu_exec.cpp, against the dense solve of the same operator -- the bar of the reference's own test
(tests/test_hmatrix.py:98-128: lu_solve of y = A x_ref recovers x_ref to epsilon).  No GPU needed."""
import numpy as np
import pytest

import Htool
from oracle import hlu as ohlu
from oracle import oracle as orc


def make_case(n, leaf, eps, eta=10.0, children=2):
    np.random.seed(0)
    pts = orc.points_in_sphere(n)
    oc = orc.Cluster(pts, max_leaf=leaf, n_children=children)
    H = orc.HMatrix(oc, oc, orc.K_INV_DELTA, 0.1, eps=eps, eta=eta)
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(leaf)
    cl = b.create_cluster_tree(np.asfortranarray(pts), children, 2 if children == 2 else children)
    return H, cl


@pytest.mark.parametrize("n,leaf,eta,children", [(300, 40, 1e-4, 2), (900, 30, 10.0, 2), (700, 25, 10.0, 3), (1200, 60, 3.0, 2)])
def test_plan_executed_on_the_cpu_solves_the_system(n, leaf, eta, children):
    eps, eps_lu = 1e-3, 1e-4
    H, cl = make_case(n, leaf, eps, eta, children)
    plan = Htool.HLUPlan(cl, H.leaves, eps_lu, cap_factor=2.5 * np.log(eps_lu) / np.log(eps))
    lu = ohlu.HostLU(plan, H.leaf_data, eps_lu)
    A = H.to_dense()  # cluster numbering
    x_ref = np.ones(n)
    x = lu.solve(A @ x_ref)
    assert np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref) < eps  # the reference's bar
    B = np.random.default_rng(1).normal(size=(n, 3))
    Xd = np.linalg.solve(A, B)
    X = lu.solve(B)
    assert np.linalg.norm(X - Xd) / np.linalg.norm(Xd) < 5 * eps_lu * max(1.0, np.sqrt(np.linalg.cond(A)) / 10)
    Xt = lu.solve(B, "T")
    assert np.linalg.norm(Xt - np.linalg.solve(A.T, B)) / np.linalg.norm(Xd) < 5 * eps_lu * max(1.0, np.sqrt(np.linalg.cond(A)) / 10)
    assert lu.counters[0] == 0  # no leaf ran out of room


def test_all_dense_operator_is_factorised_exactly():
    H, cl = make_case(260, 40, 1e-8, eta=1e-4)
    assert (H.leaves[:, 4] < 0).all()
    plan = Htool.HLUPlan(cl, H.leaves, 1e-8)
    lu = ohlu.HostLU(plan, H.leaf_data, 1e-8)
    A = H.to_dense()
    B = np.random.default_rng(2).normal(size=(260, 2))
    assert np.linalg.norm(lu.solve(B) - np.linalg.solve(A, B)) / np.linalg.norm(B) < 1e-12


def test_levels_are_real_dependencies():
    """The independent work items of every launch executed in a random order give the same factors bit for bit."""
    H, cl = make_case(800, 30, 1e-3)
    plan = Htool.HLUPlan(cl, H.leaves, 1e-4, window_tasks=500)  # several windows
    assert plan.info()[7] > 2
    a = ohlu.HostLU(plan, H.leaf_data, 1e-4)
    b = ohlu.HostLU(plan, H.leaf_data, 1e-4, shuffle=11)
    assert np.array_equal(a.factor, b.factor) and np.array_equal(a.diag, b.diag) and np.array_equal(a.ranks(), b.ranks())
    rhs = np.random.default_rng(3).normal(size=(800, 2))
    assert np.array_equal(a.solve(rhs), b.solve(rhs, shuffle=5))
    assert np.array_equal(a.solve(rhs, "T"), b.solve(rhs, "T", shuffle=9))


def test_plan_refuses_leaves_that_do_not_tile():
    H, cl = make_case(400, 30, 1e-3)
    with pytest.raises(RuntimeError):
        Htool.HLUPlan(cl, H.leaves[:-3], 1e-3)


def test_long_update_runs_are_split_over_stage_blocks():
    """A run of many updates of one leaf in one launch is dealt out to several workgroups with a stage block each (hlu_symbolic.cpp:
    split_long_runs); the result solves the system as well as the unsplit plan's."""
    eps, eps_lu = 1e-3, 1e-4
    H, cl = make_case(1500, 30, eps)
    split = Htool.HLUPlan(cl, H.leaves, eps_lu)
    info = dict(zip(ohlu.INFO, split.info()))
    lu = ohlu.HostLU(split, H.leaf_data, eps_lu)
    assert len(lu.leaves) > lu.n_leaves            # stage blocks exist
    A = H.to_dense()
    B = np.random.default_rng(5).normal(size=(1500, 2))
    Xd = np.linalg.solve(A, B)
    assert np.linalg.norm(lu.solve(B) - Xd) / np.linalg.norm(Xd) < 5 * eps_lu * max(1.0, np.sqrt(np.linalg.cond(A)) / 10)
    other = ohlu.HostLU(split, H.leaf_data, eps_lu, shuffle=3)
    assert np.array_equal(lu.factor, other.factor)  # the parts of a run and their merge are ordered by the plan, not by the schedule
    assert info["leaves"] == len(H.leaves)


@pytest.mark.parametrize("n,leaf,children", [(900, 30, 2), (700, 25, 3), (1600, 50, 2)])
def test_symmetric_plan_is_a_hierarchical_cholesky(n, leaf, children):
    """Params::symmetric: the lower triangle only, A = L L^T; half the tasks of the LU, the same solution."""
    eps, eps_lu = 1e-3, 1e-4
    H, cl = make_case(n, leaf, eps, 10.0, children)
    lower = np.where(H.leaves[:, 0] >= H.leaves[:, 2])[0]
    plan = Htool.HLUPlan(cl, H.leaves[lower], eps_lu, symmetric=True)
    full = Htool.HLUPlan(cl, H.leaves, eps_lu)
    assert plan.info()[8] < 0.7 * full.info()[8]           # tasks
    lu = ohlu.HostLU(plan, lambda i: H.leaf_data(int(lower[i])), eps_lu)
    assert lu.counters[4] == 0                              # positive definite
    A = H.to_dense()
    B = np.random.default_rng(4).normal(size=(n, 3))
    Xd = np.linalg.solve(A, B)
    bar = 5 * eps_lu * max(1.0, np.sqrt(np.linalg.cond(A)) / 10)
    assert np.linalg.norm(lu.solve(B) - Xd) / np.linalg.norm(Xd) < bar
    assert np.linalg.norm(lu.solve(B, "T") - Xd) / np.linalg.norm(Xd) < bar
    x = lu.solve(A @ np.ones(n))
    assert np.linalg.norm(x - 1) / np.sqrt(n) < eps
    other = ohlu.HostLU(plan, lambda i: H.leaf_data(int(lower[i])), eps_lu, shuffle=7)
    assert np.array_equal(lu.factor, other.factor) and np.array_equal(lu.diag, other.diag)


@pytest.mark.parametrize("sym", [False, True])
def test_shortened_solve_programs_give_the_same_solution(sym):
    """The solve programs are shortened by explicit inverse factors of the small diagonal blocks and by private slots + REDUCE tasks at the
    tree nodes above (csrc/hlu.hpp: Super, T_REDUCE): far fewer dependent levels, the same factors, the same solution to rounding."""
    eps, eps_lu, n = 1e-3, 1e-4, 2400
    H, cl = make_case(n, 30, eps)
    ids = np.where(H.leaves[:, 0] >= H.leaves[:, 2])[0] if sym else np.arange(len(H.leaves))
    leaf_data = lambda i: H.leaf_data(int(ids[i]))  # noqa: E731
    plain = Htool.HLUPlan(cl, H.leaves[ids], eps_lu, symmetric=sym, super_rows=0, solve_slots=0)
    short = Htool.HLUPlan(cl, H.leaves[ids], eps_lu, symmetric=sym, super_rows=256)
    ip, is_ = dict(zip(ohlu.INFO, plain.info())), dict(zip(ohlu.INFO, short.info()))
    assert is_["solve_levels"] < 0.5 * ip["solve_levels"] and is_["factor_tasks"] == ip["factor_tasks"]
    a, b = ohlu.HostLU(plain, leaf_data, eps_lu), ohlu.HostLU(short, leaf_data, eps_lu)
    nf = min(len(a.factor), len(b.factor))
    assert np.array_equal(a.factor[:nf], b.factor[:nf])      # the factorisation itself is the same program
    B = np.random.default_rng(8).normal(size=(n, 11))        # (more columns than a slot holds: the solve runs in chunks)
    for trans in ("N", "T"):
        Xa, Xb = a.solve(B, trans), b.solve(B, trans)
        assert np.linalg.norm(Xa - Xb) / np.linalg.norm(Xa) < 1e-9
    assert np.array_equal(b.solve(B), ohlu.HostLU(short, leaf_data, eps_lu, shuffle=4).solve(B, shuffle=6))
