"""Synthetic workloads of BASELINE.md section 2 (host-side helpers for examples and bench.py)."""
import numpy as np


def points_in_sphere(n, seed=0):
    """N points uniform in the unit ball; same construction and draw order as the reference's
    example/create_geometry.py:13-22 (u, theta, phi from numpy's global RNG after np.random.seed)."""
    np.random.seed(seed)
    u = np.random.rand(n)
    theta = 2 * np.pi * np.random.rand(n)
    phi = np.arccos(2 * np.random.rand(n) - 1)
    r = np.cbrt(u)
    return np.array([r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)])


def algorithmic_bytes(leaves, n_source, n_rows, elem_bytes):
    """B = s [ sum_dense m n + sum_lowrank r (m + n) ] + s (N_src + N_tgt),
    split per product phase: (phase A: V panels + x, phase B: U and dense panels + y)."""
    L = np.asarray(leaves, dtype=np.int64)
    dense = L[:, 4] < 0
    d = int((L[dense, 1] * L[dense, 3]).sum())
    ru = int((L[~dense, 4] * L[~dense, 1]).sum())
    rv = int((L[~dense, 4] * L[~dense, 3]).sum())
    phase_a = elem_bytes * (rv + n_source)
    phase_b = elem_bytes * (d + ru + n_rows)
    return {"total": phase_a + phase_b, "phase_a": phase_a, "phase_b": phase_b, "dense_elements": d, "u_elements": ru, "v_elements": rv}


def usable_cpus():
    """CPUs this process may really use: min(affinity mask, cgroup CPU quota)."""
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def gmres_shift(apply, n, dtype, fraction=None, power_its=6, seed=3, reduce=None):
    """Diagonal shift of the GMRES workloads (BASELINE config 5): a fixed fraction of the operator norm, estimated with a
    few power iterations of `apply` (device tensors), so that the shifted system (shift I + A) is well posed but not
    trivial -- the Krylov method needs its iterations.  `n` is this rank's slice length and `reduce(t)` sums a small device
    tensor over the ranks (distributed runs).  Returns (shift, norm estimate)."""
    import math

    import torch

    def norm(t):
        sq = torch.sum(t.real * t.real + t.imag * t.imag) if t.is_complex() else torch.sum(t * t)
        sq = sq.reshape(1).to(torch.float64)
        if reduce is not None:
            reduce(sq)
        return math.sqrt(float(sq[0]))

    g = torch.Generator(device="cpu").manual_seed(seed)
    v = torch.rand(n, dtype=torch.float64, generator=g).to(dtype).cuda()
    lam = 0.0
    for _ in range(power_its):
        v = v / norm(v)
        w = apply(v)
        lam = norm(w)
        v = w
    return (GMRES_SHIFT_FRACTION if fraction is None else fraction) * lam, lam


# fraction of |A| used as the diagonal shift of the GMRES workloads: tuned on the 500 000-point Laplace operator so that
# the relative residual 1e-6 is reached in about 40 iterations (33 at 1e-2, 62 at 5e-3) (see profiles/r02_gmres_shift_scan.json)
GMRES_SHIFT_FRACTION = 8e-3
