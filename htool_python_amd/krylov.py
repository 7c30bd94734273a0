"""Restarted GMRES on device-resident vectors (BASELINE config 5).

The operator application is the H-matrix product of this package (HIP kernels); the Krylov vector algebra
is a handful of library GEMV/AXPY calls on torch tensors (rocBLAS) -- plumbing, the hot op is the product.
Distributed runs keep one slice of every Krylov vector per GPU.  An iteration costs, besides the product (one all-gather of
the iterate's slices inside the library):

  * TWO small all-reduces -- classical Gram-Schmidt applied twice (CGS2); the second reduce carries, behind the second
    pass's coefficients h2 = V^H w, the square w.w of the vector it was taken from, and the norm of the orthogonalised
    vector follows from |w - V h2|^2 = w.w - |h2|^2 (V has orthonormal rows), so there is no third reduce for the norm;
  * ONE device -> host read-back (j + 2 coefficients per right-hand side), the only host synchronisation of the iteration.

The (restart+1) x restart Hessenberg least-squares problem is kept triangular on the host by Givens rotations (O(k) work per
iteration on the coefficients that came back with that one read-back).

Several right-hand sides (`b` of shape (mu, n)) are solved as mu independent GMRES recurrences IN LOCKSTEP: one operator
application per iteration for all columns (the H-matrix sweeps up to 8 / 16 columns per pass over its panels), one
batched GEMM per Gram-Schmidt pass, the same two reduces and one read-back; a column that has converged is frozen
(its later Krylov vectors are zero) while the others go on.
"""
import math

import numpy as np
import torch

# instrumentation for the tests: device -> host read-backs and reduce calls issued by gmres() since import
HOST_SYNCS = 0
REDUCE_CALLS = 0


def _to_host(t):
    global HOST_SYNCS
    HOST_SYNCS += 1
    return t.cpu().numpy()


def _givens(ha, hb):
    """c real, s complex with [c s; -conj(s) c] [ha; hb] = [r; 0]."""
    denom = math.sqrt(abs(ha) ** 2 + abs(hb) ** 2)
    if denom == 0.0:
        return 1.0, 0.0
    if ha == 0:
        return 0.0, 1.0
    return abs(ha) / denom, (ha / abs(ha)) * np.conj(hb) / denom


def gmres(apply, b, x0=None, tol=1e-6, restart=50, max_it=200, reduce=None, callback=None, precond=None):
    """Solve A x = b.  `apply(v)` returns A v for a device tensor v (this rank's slice); `reduce(t)` sums
    the small device tensor t over the ranks in place (None for one rank); `precond(v)` applies M^-1 to this
    rank's slice (right preconditioning: A M^-1 u = b, x = M^-1 u; `-hpddm_variant right`).  Returns (x, info);
    info["residuals"][k] is the relative residual |b - A x_k| / |b| given by the Arnoldi recurrence.

    `b` of shape (mu, n) solves mu systems in lockstep; `apply` / `precond` then receive and return (mu, n) tensors,
    info["residuals"] is a list of mu lists, info["iterations"] the largest iteration count of a column and
    info["iterations_per_column"] all of them."""
    global REDUCE_CALLS
    batched = b.dim() == 2
    B = b if batched else b.unsqueeze(0)
    mu, n = B.shape
    dev, dt = B.device, B.dtype
    cplx = dt.is_complex
    npdt = np.complex128 if cplx else np.float64
    has_out = bool(getattr(apply, "supports_out", False))  # apply(v, out=w) writes A v into w (no temporary, no copy)

    def allsum(t):
        global REDUCE_CALLS
        if reduce is not None:
            REDUCE_CALLS += 1
            reduce(t)
        return t

    def op(Z):  # (mu, n) in and out
        return apply(Z) if batched else apply(Z[0]).unsqueeze(0)

    def prec(Z):
        if precond is None:
            return Z
        return precond(Z) if batched else precond(Z[0]).unsqueeze(0)

    def dots(Vc, W):  # Vc (mu, k, n), W (mu, n) -> (mu, k): rows of Vc (conjugated) against W
        # V^H w = conj(V conj(w)): the conjugations touch vectors of n and k entries, never the k x n block
        Wc = W.conj_physical() if cplx else W
        if mu == 1:
            h = torch.mv(Vc[0], Wc[0]).unsqueeze(0)
        else:
            h = torch.bmm(Vc, Wc.unsqueeze(2)).squeeze(2)
        return h.conj_physical() if cplx else h

    def subtract(W, Vc, H):  # W (mu, n) -= sum_k H[:, k] Vc[:, k, :]   (in place)
        if mu == 1:
            W[0].addmv_(Vc[0].t(), H[0], alpha=-1)
        else:
            W.unsqueeze(2).baddbmm_(Vc.transpose(1, 2), H.unsqueeze(2), alpha=-1)

    def norms(W):  # (mu,) host floats; one reduce, one read-back
        sq = (W.real * W.real + W.imag * W.imag).sum(dim=1) if W.is_complex() else (W * W).sum(dim=1)
        return np.sqrt(_to_host(allsum(sq.to(torch.float64))))

    X = torch.zeros_like(B) if x0 is None else (x0 if batched else x0.unsqueeze(0)).clone()
    bnorm = norms(B)
    res_hist = [[] for _ in range(mu)]
    its = [0] * mu
    converged = [bool(bn == 0.0) for bn in bnorm]
    if all(converged):
        X = torch.zeros_like(B)
        info = {"residuals": res_hist if batched else res_hist[0], "iterations": 0, "converged": True, "restarts": 0, "iterations_per_column": its}
        return (X if batched else X[0]), info
    for c in range(mu):
        if converged[c]:
            X[c] = 0
    m = max(1, int(restart))
    V = torch.empty(mu, m + 1, n, dtype=dt, device=dev)
    restarts = 0
    first = x0 is None
    while max(its) < max_it and not all(converged):
        Rr = B.clone() if first else B - op(X)
        first = False
        beta = norms(Rr)
        scale = np.zeros(mu)
        active = [False] * mu
        for c in range(mu):
            if converged[c]:
                continue
            if beta[c] / bnorm[c] <= tol:
                converged[c] = True
                continue
            active[c] = True
            scale[c] = 1.0 / beta[c]
        if not any(active):
            break
        V[:, 0] = Rr * (torch.from_numpy(scale).to(device=dev, dtype=dt).unsqueeze(1) if mu > 1 else scale[0])
        # Hessenberg matrices reduced to upper triangular form R by Givens rotations as their columns arrive; g = rotated beta e_1,
        # |g[k]| = residual norm after k iterations
        R = np.zeros((mu, m + 1, m), dtype=npdt)
        cs, sn = np.zeros((mu, m), dtype=npdt), np.zeros((mu, m), dtype=npdt)
        g = np.zeros((mu, m + 1), dtype=npdt)
        g[:, 0] = beta * np.asarray(active, dtype=np.float64)
        kdone = [0] * mu  # Krylov dimension each column ends this cycle with
        running = list(active)
        for j in range(m):
            W = V[:, j + 1]
            Z = prec(V[:, j])
            if has_out:
                apply(Z if batched else Z[0], out=W if batched else W[0])
            else:
                W.copy_(op(Z))
            Vj = V[:, : j + 1]
            h1 = allsum(dots(Vj, W))
            subtract(W, Vj, h1)
            t2 = allsum(dots(V[:, : j + 2], W))  # [h2 ; w.w]: the last row of V[:, : j + 2] is W itself
            subtract(W, Vj, t2[:, : j + 1])
            host = _to_host(torch.cat([h1, t2], dim=1))  # the iteration's one synchronisation
            h1h, h2h, ww = host[:, : j + 1], host[:, j + 1: 2 * j + 2], host[:, 2 * j + 2].real
            hn2 = ww - np.sum(np.abs(h2h) ** 2, axis=1)
            if np.any(hn2[running] < 1e-2 * ww[running]):  # the second pass removed most of w: do not trust the difference
                hn_all = norms(W)
                hn2 = np.where(hn2 < 1e-2 * ww, hn_all ** 2, hn2)
            hn = np.sqrt(np.maximum(hn2, 0.0))
            inv = np.zeros(mu)
            stop = False
            for c in range(mu):
                if not running[c]:
                    continue
                col = np.zeros(j + 2, dtype=npdt)
                col[: j + 1] = h1h[c] + h2h[c]
                col[j + 1] = hn[c]
                for i in range(j):  # earlier rotations
                    t = cs[c, i] * col[i] + sn[c, i] * col[i + 1]
                    col[i + 1] = -np.conj(sn[c, i]) * col[i] + cs[c, i] * col[i + 1]
                    col[i] = t
                cs[c, j], sn[c, j] = _givens(col[j], col[j + 1])
                col[j] = cs[c, j] * col[j] + sn[c, j] * col[j + 1]
                col[j + 1] = 0.0
                R[c, : j + 2, j] = col
                g[c, j + 1] = -np.conj(sn[c, j]) * g[c, j]
                g[c, j] = cs[c, j] * g[c, j]
                kdone[c] = j + 1
                res = abs(g[c, j + 1]) / bnorm[c]
                its[c] += 1
                res_hist[c].append(float(res))
                if callback is not None and not batched:
                    callback(its[c], float(res))
                if res <= tol or its[c] >= max_it or hn[c] == 0:
                    running[c] = False  # frozen: its next Krylov vector is zeroed below
                else:
                    inv[c] = 1.0 / hn[c]
            if not any(running):
                stop = True
            if not stop and j + 1 < m:
                if mu == 1:
                    W.mul_(inv[0])
                else:
                    W.mul_(torch.from_numpy(inv).to(device=dev, dtype=dt).unsqueeze(1))
            if stop:
                break
        kmax = max(kdone)
        Y = np.zeros((mu, kmax), dtype=npdt)
        for c in range(mu):
            k = kdone[c]
            for i in range(k - 1, -1, -1):  # back substitution R y = g
                Y[c, i] = (g[c, i] - R[c, i, i + 1: k] @ Y[c, i + 1: k]) / R[c, i, i] if R[c, i, i] != 0 else 0.0
        Yd = torch.from_numpy(Y).to(device=dev, dtype=dt)
        if mu == 1:
            dx = torch.mv(V[0, :kmax].t(), Yd[0]).unsqueeze(0)
        else:
            dx = torch.bmm(V[:, :kmax].transpose(1, 2), Yd.unsqueeze(2)).squeeze(2)
        X = X + prec(dx)
        restarts += 1
        for c in range(mu):
            if active[c] and res_hist[c] and res_hist[c][-1] <= tol:
                converged[c] = True
    info = {"residuals": res_hist if batched else res_hist[0], "iterations": max(its), "converged": all(converged), "restarts": restarts,
            "iterations_per_column": its}
    return (X if batched else X[0]), info
