"""Restarted GMRES on device-resident vectors (BASELINE config 5).

The operator application is the H-matrix product of this package (HIP kernels); the Krylov vector algebra
is a handful of library GEMV/AXPY calls on torch tensors (rocBLAS) -- plumbing, the hot op is the product.
Distributed runs keep one slice of every Krylov vector per GPU.  An iteration costs, besides the product (one all-gather of
the iterate's slices inside the library):

  * TWO small all-reduces -- classical Gram-Schmidt applied twice (CGS2); the second reduce carries, behind the second
    pass's coefficients h2 = V^H w, the square w.w of the vector it was taken from, and the norm of the orthogonalised
    vector follows from |w - V h2|^2 = w.w - |h2|^2 (V has orthonormal rows), so there is no third reduce for the norm;
  * ONE device -> host read-back (2 j + 4 coefficients per right-hand side), the only host synchronisation of the iteration --
    and a lagging one: the orthogonalised vector is normalised ON THE DEVICE (1 / sqrt(w.w - |h2|^2)), step j + 1 is enqueued
    before the host looks at step j's coefficients (pinned buffer + event), so the device never waits for the Givens rotations.
    Convergence is therefore noticed one step late: the step enqueued for nothing is ignored.

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


class _Readback:
    """Small device -> host transfers that do not stall the stream: the tensor is copied into one of two pinned buffers, an event
    marks the copy, and the host waits for that event only when it needs the numbers -- one step later (gmres)."""

    def __init__(self, dev, dt, count):
        self.cuda = dev.type == "cuda"
        self.turn = 0
        if self.cuda:
            self.buf = [torch.empty(count, dtype=dt).pin_memory() for _ in range(2)]
            self.ev = [torch.cuda.Event() for _ in range(2)]

    def post(self, t):
        if not self.cuda:
            return t.clone()
        s, self.turn = self.turn, self.turn ^ 1
        view = self.buf[s][: t.numel()].view(t.shape)
        view.copy_(t, non_blocking=True)
        self.ev[s].record()
        return (s, view)

    def wait(self, handle):
        global HOST_SYNCS
        HOST_SYNCS += 1
        if not self.cuda:
            return handle.numpy()
        s, view = handle
        self.ev[s].synchronize()
        return view.numpy().copy()


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
    rb = _Readback(dev, dt, mu * (2 * m + 4))
    finish_kernel = None
    if dev.type == "cuda":
        # the tail of a step as ONE launch of the library (csrc/krylov_device.hip: second projection taken out, norm, scaling,
        # coefficient row) instead of a GEMV and ten one-element tensor operations; the two Gram-Schmidt dot-product passes
        # stay library GEMVs (kernels of our own for them were slower: include/htool_mi355x.h)
        from . import Htool as _core

        finish_kernel = _core.krylov_finish_step
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
        mask_host = np.asarray(running, dtype=np.float64)
        mask_dev = torch.from_numpy(mask_host.copy()).to(dev) if mu > 1 else None

        def enqueue_step(j):
            """Device work of Arnoldi step j -- product, CGS2 with the norm folded into the second reduce, normalisation -- without
            any host synchronisation; the coefficients go to the host asynchronously.  Row layout of what is posted, per column:
            [h1 (j+1) | h2 (j+1) | w.w | hn^2]."""
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
            h2 = t2[:, : j + 1]
            if finish_kernel is not None:
                # one launch of the library for the tail: w -= V h2, hn^2 = w.w - |h2|^2, w scaled by mask / hn, the coefficient row packed
                coef = torch.empty(mu, 2 * j + 4, dtype=dt, device=dev)
                finish_kernel(W.data_ptr(), W.stride(0) if mu > 1 else n, n, mu, cplx, h1.data_ptr(), t2.data_ptr(), j, mask_dev.data_ptr() if mask_dev is not None else 0,
                              coef.data_ptr(), j + 1 < m, V.data_ptr(), n, (m + 1) * n, torch.cuda.current_stream().cuda_stream)
                return rb.post(coef)
            subtract(W, Vj, h2)
            ww = t2[:, j + 1].real if cplx else t2[:, j + 1]
            hn2 = ww - ((h2.real * h2.real + h2.imag * h2.imag) if cplx else h2 * h2).sum(dim=1)  # |w - V h2|^2 = w.w - |h2|^2
            if j + 1 < m:
                inv = torch.where(hn2 > 0, torch.rsqrt(hn2.clamp_min(1e-300)), torch.zeros_like(hn2))
                if mask_dev is not None:
                    inv = inv * mask_dev  # columns known to be finished stay zero
                W.mul_(inv.unsqueeze(1))
            return rb.post(torch.cat([h1, t2, hn2.unsqueeze(1).to(dt)], dim=1))

        def process_step(j, host):
            """Host work of step j on the coefficients that came back: Givens rotations, residuals, who goes on.  Returns True when
            the cycle has to end after this step."""
            nonlocal mask_dev
            h1h, h2h = host[:, : j + 1], host[:, j + 1: 2 * j + 2]
            ww, hn2 = host[:, 2 * j + 2].real, host[:, 2 * j + 3].real
            hn = np.sqrt(np.maximum(hn2, 0.0))
            end_cycle = False
            changed = False
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
                    running[c] = False  # frozen: its later Krylov vectors are ignored (and zeroed once the device knows)
                    changed = True
                elif hn2[c] < 1e-2 * ww[c]:
                    # the second pass removed most of w (orthogonality nearly lost, or close to a breakdown): the difference
                    # w.w - |h2|^2 is then inaccurate and so is the next basis vector's length -- end the cycle here and restart
                    # from the true residual
                    end_cycle = True
            if changed and mask_dev is not None and any(running):
                mask_dev = torch.from_numpy(np.asarray(running, dtype=np.float64)).to(dev)
            return end_cycle or not any(running)

        # Pipelined: step j is enqueued BEFORE the host looks at step j - 1 (whose coefficients have long arrived), so the device
        # never waits for the host; a step enqueued for nothing (convergence is only known one step late) is simply ignored.
        posted = None
        enq = list(its)  # iterations counted including the enqueued, not yet processed steps
        for j in range(m):
            if not any(running[c] and enq[c] < max_it for c in range(mu)):
                break
            handle = enqueue_step(j)
            for c in range(mu):
                enq[c] += 1 if running[c] else 0
            if posted is not None:
                stop = process_step(posted[0], rb.wait(posted[1]))
                posted = None
                if stop:
                    rb.wait(handle)  # (drain: the slot is reused by the next cycle)
                    break
            posted = (j, handle)
        if posted is not None:
            process_step(posted[0], rb.wait(posted[1]))
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
