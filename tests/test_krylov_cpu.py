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
