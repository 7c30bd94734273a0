"""Restarted GMRES on device-resident vectors (BASELINE config 5).

The operator application is the H-matrix product of this package (HIP kernels); the Krylov vector algebra
is a handful of library GEMV/AXPY calls on torch tensors (rocBLAS) -- plumbing, the hot op is the product.
Distributed runs keep one slice of every Krylov vector per GPU; the Gram-Schmidt coefficients of an
iteration are reduced with ONE all-reduce per orthogonalisation pass (classical Gram-Schmidt applied
twice, CGS2); the operator itself needs one all-gather of the iterate's slices.
The (restart+1) x restart Hessenberg least-squares problem is solved on the host with numpy.
"""
import math

import numpy as np
import torch


def gmres(apply, b, x0=None, tol=1e-6, restart=50, max_it=200, reduce=None, callback=None, precond=None):
    """Solve A x = b.  `apply(v)` returns A v for a device tensor v (this rank's slice); `reduce(t)` sums
    the small device tensor t over the ranks in place (None for one rank); `precond(v)` applies M^-1 to this
    rank's slice (right preconditioning: A M^-1 u = b, x = M^-1 u; `-hpddm_variant right`).  Returns (x, info);
    info["residuals"][k] is the relative residual |b - A x_k| / |b| given by the Arnoldi recurrence."""
    dev, dt = b.device, b.dtype
    n = b.numel()

    def allsum(t):
        if reduce is not None:
            reduce(t)
        return t

    def norm(v):
        sq = torch.sum(v.real * v.real + v.imag * v.imag) if v.is_complex() else torch.sum(v * v)
        return math.sqrt(float(allsum(sq.reshape(1).to(torch.float64))[0]))

    x = torch.zeros_like(b) if x0 is None else x0.clone()
    bnorm = norm(b)
    info = {"residuals": [], "iterations": 0, "converged": False, "restarts": 0}
    if bnorm == 0.0:
        info["converged"] = True
        return torch.zeros_like(b), info
    m = max(1, int(restart))
    npdt = np.complex128 if dt.is_complex else np.float64
    V = torch.empty(m + 1, n, dtype=dt, device=dev)
    first = x0 is None
    while info["iterations"] < max_it:
        r = b.clone() if first else b - apply(x)
        first = False
        beta = norm(r)
        if beta / bnorm <= tol:
            info["converged"] = True
            break
        V[0] = r / beta
        Hh = np.zeros((m + 1, m), dtype=npdt)
        y, k = None, 0
        for j in range(m):
            w = apply(V[j] if precond is None else precond(V[j]))
            Vj = V[: j + 1]
            h = allsum(torch.mv(Vj.conj(), w))
            w = w - torch.mv(Vj.t(), h)
            h2 = allsum(torch.mv(Vj.conj(), w))
            w = w - torch.mv(Vj.t(), h2)
            hn = norm(w)
            Hh[: j + 1, j] = (h + h2).cpu().numpy()
            Hh[j + 1, j] = hn
            if hn > 0:
                V[j + 1] = w / hn
            k = j + 1
            rhs = np.zeros(k + 1, dtype=npdt)
            rhs[0] = beta
            y = np.linalg.lstsq(Hh[: k + 1, :k], rhs, rcond=None)[0]
            res = float(np.linalg.norm(rhs - Hh[: k + 1, :k] @ y)) / bnorm
            info["iterations"] += 1
            info["residuals"].append(res)
            if callback is not None:
                callback(info["iterations"], res)
            if res <= tol or info["iterations"] >= max_it or hn == 0:
                break
        dx = torch.mv(V[:k].t(), torch.from_numpy(y).to(device=dev, dtype=dt))
        x = x + (dx if precond is None else precond(dx))
        info["restarts"] += 1
        if info["residuals"][-1] <= tol:
            info["converged"] = True
            break
    return x, info
