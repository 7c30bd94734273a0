"""CPU test of the restarted GMRES (htool_python_amd/krylov.py) on dense torch operators: Arnoldi residual = true residual,
across restarts, with and without a right preconditioner, real and complex."""
import numpy as np
import pytest
import torch

from htool_python_amd.krylov import gmres


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("restart", [60, 8])
@pytest.mark.parametrize("preconditioned", [False, True])
def test_gmres_residual_recurrence_and_restarts(cplx, restart, preconditioned):
    rng = np.random.RandomState(0)
    n = 300
    A = rng.rand(n, n) + (1j * rng.rand(n, n) if cplx else 0) + n * 0.05 * np.eye(n)
    At = torch.from_numpy(A)
    Minv = torch.from_numpy(np.linalg.inv(np.diag(np.diag(A))))
    x_ref = torch.from_numpy(rng.rand(n) + (1j * rng.rand(n) if cplx else 0))
    b = At @ x_ref
    seen = []
    x, info = gmres(lambda v: At @ v, b, tol=1e-9, restart=restart, max_it=600, precond=(lambda v: Minv @ v) if preconditioned else None,
                    callback=lambda it, res: seen.append(res))
    true_res = float(torch.linalg.norm(b - At @ x) / torch.linalg.norm(b))
    assert info["converged"] and info["residuals"][-1] <= 1e-9 and len(seen) == info["iterations"]
    assert abs(true_res - info["residuals"][-1]) <= 1e-3 * true_res + 1e-14
    assert float(torch.linalg.norm(x - x_ref) / torch.linalg.norm(x_ref)) < 1e-7
    if restart == 8:
        assert info["restarts"] >= 2


def _reference_gmres_history(A, b, restart, iters):
    """Textbook GMRES residual history (modified Gram-Schmidt + explicit least squares per step, numpy) -- shares nothing
    with krylov.py beyond the definition: min over the Krylov space of |b - A x|."""
    n = len(b)
    hist, x = [], np.zeros(n, dtype=A.dtype)
    bn = np.linalg.norm(b)
    while len(hist) < iters:
        r = b - A @ x
        beta = np.linalg.norm(r)
        V = np.zeros((restart + 1, n), dtype=A.dtype)
        H = np.zeros((restart + 1, restart), dtype=A.dtype)
        V[0] = r / beta
        for j in range(restart):
            w = A @ V[j]
            for i in range(j + 1):
                H[i, j] = np.vdot(V[i], w)
                w = w - H[i, j] * V[i]
            H[j + 1, j] = np.linalg.norm(w)
            V[j + 1] = w / H[j + 1, j]
            e1 = np.zeros(j + 2, dtype=A.dtype)
            e1[0] = beta
            y, *_ = np.linalg.lstsq(H[: j + 2, : j + 1], e1, rcond=None)
            hist.append(np.linalg.norm(e1 - H[: j + 2, : j + 1] @ y) / bn)
            if len(hist) >= iters:
                break
        x = x + V[: j + 1].T @ y
    return np.array(hist)


@pytest.mark.parametrize("cplx", [False, True])
def test_two_reduces_and_one_readback_per_iteration_same_history(cplx):
    """<= 2 reduce calls and 1 host synchronisation per iteration (the norm rides on the second
    reduce: |w - V h2|^2 = w.w - |h2|^2), and the residual history of the fused recurrence equals a textbook GMRES's to 1e-12
    (relative to the first residual)."""
    from htool_python_amd import krylov

    rng = np.random.RandomState(1)
    n = 240
    A = rng.rand(n, n) + (1j * rng.rand(n, n) if cplx else 0) + n * 0.08 * np.eye(n)
    b = rng.rand(n) + (1j * rng.rand(n) if cplx else 0)
    At, bt = torch.from_numpy(A), torch.from_numpy(b)
    restart, iters = 12, 30
    # the reduce of a one-rank run is the identity: what is counted is how often the recurrence asks for one
    calls = {"reduce": 0}

    def reduce(t):
        calls["reduce"] += 1
        return t

    s0, r0 = krylov.HOST_SYNCS, krylov.REDUCE_CALLS
    x, info = gmres(lambda v: At @ v, bt, tol=0.0, restart=restart, max_it=iters, reduce=reduce)
    syncs, reduces = krylov.HOST_SYNCS - s0, krylov.REDUCE_CALLS - r0
    cycles = info["restarts"]
    assert info["iterations"] == iters and cycles == 3
    # per iteration: 2 reduces + 1 read-back; per cycle: the norm of the restart residual (1 + 1); once: |b| (1 + 1)
    assert reduces == calls["reduce"] == 2 * iters + cycles + 1
    assert syncs == iters + cycles + 1
    ref = _reference_gmres_history(A, b, restart, iters)
    got = np.array(info["residuals"])
    assert np.max(np.abs(got - ref)) <= 1e-12 * ref[0] + 1e-15, np.max(np.abs(got - ref))
    assert abs(float(torch.linalg.norm(bt - At @ x) / torch.linalg.norm(bt)) - got[-1]) <= 1e-10


@pytest.mark.parametrize("cplx", [False, True])
def test_lockstep_block_of_right_hand_sides_equals_column_by_column(cplx):
    """b of shape (mu, n): one apply per iteration for all columns; every column's history and solution equal the single
    right-hand-side solve of that column (columns converge at different iterations and are frozen; one column is zero)."""
    rng = np.random.RandomState(2)
    n, mu = 200, 4
    A = rng.rand(n, n) + (1j * rng.rand(n, n) if cplx else 0) + n * 0.06 * np.eye(n)
    At = torch.from_numpy(A)
    B = rng.rand(mu, n) + (1j * rng.rand(mu, n) if cplx else 0)
    lam, vec = np.linalg.eig(A)
    top = vec[:, np.argmax(np.abs(lam))]   # an eigenvector as right-hand side: solved by the first iteration (Perron vector: real for real A)
    B[1] = top if cplx else top.real
    B[2] = 0.0                             # a zero right-hand side
    Bt = torch.from_numpy(B)
    applies = []

    def apply_block(Z):
        applies.append(tuple(Z.shape))
        return Z @ At.t()

    Minv = torch.from_numpy(np.linalg.inv(np.diag(np.diag(A))))
    for precond in (None, lambda Z: Z @ Minv.t()):
        applies.clear()
        X, info = gmres(apply_block, Bt, tol=1e-10, restart=25, max_it=300, precond=precond)
        assert info["converged"] and all(s == (mu, n) for s in applies)
        assert len(applies) <= info["iterations"] + 2 * info["restarts"] + 1  # (+ one look-ahead step per cycle: convergence is seen one step late)
        for c in range(mu):
            xc, ic = gmres(lambda v: At @ v, Bt[c], tol=1e-10, restart=25, max_it=300, precond=(None if precond is None else (lambda v: Minv @ v)))
            assert ic["iterations"] == info["iterations_per_column"][c]
            assert np.allclose(ic["residuals"], info["residuals"][c], rtol=1e-6, atol=1e-12)
            assert float(torch.linalg.norm(X[c] - xc)) <= 1e-9 * max(float(torch.linalg.norm(xc)), 1.0)
        assert info["iterations_per_column"][2] == 0 and float(torch.linalg.norm(X[2])) == 0.0
        assert len(set(info["iterations_per_column"])) >= 3   # the columns really stopped at different iterations
