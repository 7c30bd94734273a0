"""Restarted GMRES on device-resident vectors (BASELINE config 5).

The operator application is the H-matrix product of this package (HIP kernels); the Krylov vector algebra
is a handful of library GEMV/AXPY calls on torch tensors (rocBLAS) -- plumbing, the hot op is the product.
Distributed runs keep one slice of every Krylov vector per GPU; the Gram-Schmidt coefficients of an
iteration are reduced with ONE all-reduce per orthogonalisation pass (classical Gram-Schmidt applied
twice, CGS2); the operator itself needs one all-gather of the iterate's slices.
The (restart+1) x restart Hessenberg least-squares problem is kept triangular on the host by Givens rotations (O(k) work per
iteration on the k + 1 coefficients that come back from the device with the one synchronisation an iteration needs anyway).
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
        # Hessenberg matrix reduced to upper triangular form R by Givens rotations as its columns arrive; g = rotated beta e_1,
        # |g[k]| = residual norm after k iterations
        R = np.zeros((m + 1, m), dtype=npdt)
        cs, sn = np.zeros(m, dtype=npdt), np.zeros(m, dtype=npdt)
        g = np.zeros(m + 1, dtype=npdt)
        g[0] = beta
        k = 0
        for j in range(m):
            w = apply(V[j] if precond is None else precond(V[j]))
            Vj = V[: j + 1]
            h = allsum(torch.mv(Vj.conj(), w))
            w = w - torch.mv(Vj.t(), h)
            h2 = allsum(torch.mv(Vj.conj(), w))
            w = w - torch.mv(Vj.t(), h2)
            hn = norm(w)
            col = np.zeros(j + 2, dtype=npdt)
            col[: j + 1] = (h + h2).cpu().numpy()
            col[j + 1] = hn
            if hn > 0:
                V[j + 1] = w / hn
            for i in range(j):  # earlier rotations
                t = cs[i] * col[i] + sn[i] * col[i + 1]
                col[i + 1] = -np.conj(sn[i]) * col[i] + cs[i] * col[i + 1]
                col[i] = t
            ha, hb = col[j], col[j + 1]
            denom = math.sqrt(abs(ha) ** 2 + abs(hb) ** 2)
            if denom == 0.0:
                cs[j], sn[j] = 1.0, 0.0
            elif ha == 0:
                cs[j], sn[j] = 0.0, 1.0
            else:  # c real, s complex: [c s; -conj(s) c] [ha; hb] = [r; 0]
                cs[j] = abs(ha) / denom
                sn[j] = (ha / abs(ha)) * np.conj(hb) / denom
            col[j] = cs[j] * ha + sn[j] * hb
            col[j + 1] = 0.0
            R[: j + 2, j] = col
            g[j + 1] = -np.conj(sn[j]) * g[j]
            g[j] = cs[j] * g[j]
            k = j + 1
            res = abs(g[k]) / bnorm
            info["iterations"] += 1
            info["residuals"].append(float(res))
            if callback is not None:
                callback(info["iterations"], float(res))
            if res <= tol or info["iterations"] >= max_it or hn == 0:
                break
        y = np.zeros(k, dtype=npdt)
        for i in range(k - 1, -1, -1):  # back substitution R y = g
            y[i] = (g[i] - R[i, i + 1: k] @ y[i + 1: k]) / R[i, i] if R[i, i] != 0 else 0.0
        dx = torch.mv(V[:k].t(), torch.from_numpy(y).to(device=dev, dtype=dt))
        x = x + (dx if precond is None else precond(dx))
        info["restarts"] += 1
        if info["residuals"][-1] <= tol:
            info["converged"] = True
            break
    return x, info
