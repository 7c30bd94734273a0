"""Independent numpy checks of the cluster tree, of the block tree and of ACA -- a SECOND formulation, sharing no code and
no recurrences with oracle/hmat_oracle.cpp or with the product's host code.

TEST INFRASTRUCTURE ONLY (same rule as the rest of oracle/): imported by tests/ only.

Why it exists: oracle/hmat_oracle.cpp restates the path with the same recurrences as the product's host code
(Jacobi sweeps for the principal axis, implicit-residual ACA with the running Frobenius estimate), so an
equality test between the two cannot see a shared misreading of the algorithm.  Everything here is derived
from the DEFINITIONS instead:

  * cluster tree (argument conventions src/htool/clustering/cluster_tree_builder.hpp:19-67):
    checked as a set of properties of the node table -- centre = weighted mean, radius = max(|p - c| + radii),
    children tile the parent, children are separated by hyperplanes orthogonal to the split direction computed
    with numpy.linalg.eigh (PCA*) or from the bounding box (BoundingBox*), equal counts (Regular) or equal
    widths (Geometric), and the leaf rule sz // n_children < maximal_leaf_size.
  * ACA (compressor contract src/htool/hmatrix/interfaces/virtual_low_rank_generator.hpp:25-45):
    `aca_full_residual` keeps the EXPLICIT residual matrix R = A - sum u_k v_k and measures |sum u_k v_k|_F on the
    explicit matrix, where the engines use the implicit row/column updates and the running estimate.
  * block tree (replaces htool::HMatrixTreeBuilder::build, hmatrix_tree_builder.hpp:36): `check_block_tree`
    never runs the top-down visit -- leaves must be stopping places, reachable by walking UP through pairs whose split rule
    produces them, and tile the matrix exactly.
  * the acceptance semantics of epsilon of the reference's own compressor
    (example/advanced/define_custom_low_rank_generator.py:16-27): `svd_rank`.
"""
import numpy as np


# ----------------------------------------------------------------------------------------------
# cluster tree
# ----------------------------------------------------------------------------------------------
def split_direction(P, w, centre, strategy):
    """Unit split direction of the point set P (d, k) with weights w: principal axis of the weighted covariance about
    `centre` (strategies 0, 1) through numpy.linalg.eigh, or the longest bounding-box edge (2, 3)."""
    d = P.shape[0]
    if strategy in (0, 1):
        X = P - centre[:d, None]
        cov = (X * w) @ X.T
        vals, vecs = np.linalg.eigh(cov)
        return vecs[:, -1], (vals[-1] - vals[-2]) / max(vals[-1], 1e-300) if d > 1 else 1.0
    ext = P.max(axis=1) - P.min(axis=1)
    e = np.zeros(d)
    e[int(np.argmax(ext))] = 1.0
    srt = np.sort(ext)
    return e, (srt[-1] - srt[-2]) / max(srt[-1], 1e-300) if d > 1 else 1.0


def check_cluster_tree(ints7, dbl4, perm, points, n_children, max_leaf, strategy=0, radii=None, weights=None,
                       given_partition=False, rtol=1e-12):
    """Asserts the defining properties of a cluster tree given as node table (ints7: offset, size, depth, parent,
    first_child, n_children, partition; dbl4: cx, cy, cz, radius), permutation and (d, N) points.  Returns the
    number of split nodes whose direction was checked."""
    ints7, dbl4, perm = np.asarray(ints7), np.asarray(dbl4), np.asarray(perm)
    d, n = points.shape
    assert sorted(perm.tolist()) == list(range(n)), "permutation is not a permutation"
    w_all = np.ones(n) if weights is None else np.asarray(weights, dtype=float)
    r_all = np.zeros(n) if radii is None else np.asarray(radii, dtype=float)
    scale = max(float(np.abs(points).max()), 1.0)
    checked = 0
    assert ints7[0, 0] == 0 and ints7[0, 1] == n and ints7[0, 2] == 0
    for idx, (row, geo) in enumerate(zip(ints7, dbl4)):
        off, sz, depth, parent, first, nch, part = (int(v) for v in row)
        ids = perm[off:off + sz]
        P, w = points[:, ids], w_all[ids]
        # centre and radius from their definitions
        c = (P * w).sum(axis=1) / w.sum()
        assert np.allclose(geo[:d], c, rtol=0, atol=rtol * scale * max(1.0, np.sqrt(sz))), f"node {idx}: centre"
        dist = np.sqrt(((P - geo[:d, None]) ** 2).sum(axis=0)) + r_all[ids]
        assert dist.max() <= geo[3] * (1 + rtol) + 1e-300, f"node {idx}: a point lies outside the radius"
        assert np.isclose(dist.max(), geo[3], rtol=1e-10, atol=1e-300), f"node {idx}: radius is not attained"
        if nch == 0:
            continue
        ch = ints7[first:first + nch]
        assert np.all(ch[:, 3] == idx) and np.all(ch[:, 2] == depth + 1), f"node {idx}: children links"
        assert ch[0, 0] == off and np.all(ch[1:, 0] == ch[:-1, 0] + ch[:-1, 1]) and ch[:, 1].sum() == sz, f"node {idx}: children do not tile the parent"
        if depth == 0 and given_partition:
            continue  # depth-1 children are the user's partition, not a geometric split
        assert nch == n_children or depth == 0, f"node {idx}: number of children"
        regular = strategy in (0, 2)
        if regular:
            assert ch[:, 1].max() - ch[:, 1].min() <= nch - 1 and np.all(ch[:-1, 1] == sz // nch), f"node {idx}: Regular split is not in equal counts"
        assert np.all(ch[:, 1] >= max_leaf) or depth == 0, f"node {idx}: child below the minimum cluster size"
        direction, gap = split_direction(P, w, geo[:3], strategy)
        if gap < 1e-6:
            continue  # (nearly) degenerate principal axis: the direction is not unique, nothing to compare
        proj = direction @ (P - geo[:d, None])
        tol = 1e-9 * max(float(np.ptp(proj)), 1e-300)
        pieces = np.split(proj, np.cumsum(ch[:-1, 1]))
        pieces = [p for p in pieces if len(p)]
        lo = np.array([p.min() for p in pieces])
        hi = np.array([p.max() for p in pieces])
        inc = np.all(hi[:-1] <= lo[1:] + tol)
        dec = np.all(lo[:-1] >= hi[1:] - tol)  # the eigenvector's sign is free
        assert inc or dec, f"node {idx}: children are not separated by hyperplanes orthogonal to the split direction"
        if not regular:  # Geometric: equal widths along the direction
            pr = proj if inc else -proj
            cuts = pr.min() + (np.arange(1, nch) / nch) * np.ptp(pr)
            bounds = np.cumsum(ch[:-1, 1])
            spr = np.sort(pr)
            for cut, b in zip(cuts, bounds):
                below = int(np.searchsorted(spr, cut - tol, side="left")), int(np.searchsorted(spr, cut + tol, side="left"))
                assert below[0] <= b <= below[1], f"node {idx}: Geometric split is not in equal widths"
        checked += 1
    # leaf rule: a leaf could not have been split (Regular: every child would get sz // n_children points)
    for idx, row in enumerate(ints7):
        off, sz, depth, parent, first, nch, part = (int(v) for v in row)
        if nch == 0 and strategy in (0, 2) and not (depth == 0 and given_partition):
            assert sz // n_children < max_leaf, f"leaf {idx} of size {sz} should have been split"
    return checked


# ----------------------------------------------------------------------------------------------
# ACA, explicit-residual formulation
# ----------------------------------------------------------------------------------------------
def aca_full_residual(A, eps, reqrank=-1, transpose_role=False):
    """Partially pivoted ACA of the explicit matrix A on the EXPLICIT residual.

    Returns (U, V) with A ~ U @ V (U m x r, V r x n), or None when the block is rejected (r (m + n) > m n).
    transpose_role: run on A^T and transpose the factors back (the engines' rule for leaves below the diagonal)."""
    if transpose_role:
        res = aca_full_residual(np.asarray(A).T, eps, reqrank, False)
        return None if res is None else (res[1].T.copy(), res[0].T.copy())
    A = np.array(A)
    m, n = A.shape
    R = A.copy()
    S = np.zeros_like(A)  # explicit sum of the rank-one terms
    used_r, used_c = np.zeros(m, bool), np.zeros(n, bool)
    us, vs = [], []
    i = 0
    while len(us) < min(m, n):
        if reqrank >= 0 and len(us) >= reqrank:
            break
        row = R[i].copy()
        used_r[i] = True
        cand = np.where(used_c, -1.0, np.abs(row) ** 2)
        j = int(np.argmax(cand))  # first maximum = lowest index on ties
        if cand[j] < 0:
            break
        if np.sqrt(cand[j]) <= 1e-15:  # null row: next unused one
            free = np.flatnonzero(~used_r)
            if len(free) == 0:
                break
            i = int(free[0])
            continue
        col = R[:, j] / row[j]
        used_c[j] = True
        term = np.outer(col, row)
        R -= term
        S += term
        us.append(col)
        vs.append(row)
        k = len(us)
        if k * (m + n) > m * n:
            return None
        if reqrank < 0 and np.linalg.norm(col) * np.linalg.norm(row) <= eps * np.linalg.norm(S):
            break
        cand = np.where(used_r, -1.0, np.abs(col) ** 2)
        i = int(np.argmax(cand))
        if cand[i] < 0:
            break
    if not us:
        return np.zeros((m, 0), A.dtype), np.zeros((0, n), A.dtype)
    return np.array(us).T.copy(), np.array(vs)


def svd_rank(A, eps):
    """Smallest r with sqrt(sum_{i >= r} sigma_i^2) / |A|_F < eps -- the reference's own reading of epsilon
    (example/advanced/define_custom_low_rank_generator.py:16-27)."""
    s = np.linalg.svd(np.asarray(A), compute_uv=False)
    tail = np.sqrt(np.cumsum((s ** 2)[::-1])[::-1])  # tail[r] = |sigma[r:]|
    ok = np.flatnonzero(tail / max(np.linalg.norm(s), 1e-300) < eps)
    return int(ok[0]) if len(ok) else len(s)


def leaf_quality(A, U, V, eps):
    """(relative Frobenius error of U V against A, rank, SVD rank at eps / 10)."""
    err = np.linalg.norm(A - U @ V) / max(np.linalg.norm(A), 1e-300)
    return float(err), U.shape[1], svd_rank(A, eps / 10)


# ---------------------------------------------------------------------------------------------------------------------
# block tree, verified from its definition WITHOUT running the top-down visit the engine and the C++
# oracle both implement: a leaf list is right iff (1) every leaf is a pair of cluster nodes, (2) the leaves tile the
# matrix exactly, (3) every leaf is a place where the visit stops (admissible at sufficient depth -> low rank; two cluster
# leaves -> dense) and (4) every leaf is REACHABLE: walking UP, there is a chain of pairs from it to the root pair in which
# every pair is one where the visit does not stop and whose split rule produces the next pair of the chain.
# ---------------------------------------------------------------------------------------------------------------------
def check_block_tree(adm, dns, target_nodes, source_nodes, eta, t_root=0, s_root=0, min_target_depth=0, min_source_depth=0, paint_limit=4000):
    """adm, dns: (k, 4) arrays of (t_off, m, s_off, n); target_nodes / source_nodes: (ints7, dbl4) node tables as in
    check_cluster_tree.  Returns (#admissible leaves checked, #dense leaves checked)."""
    ti, td = (np.asarray(a) for a in target_nodes)
    si, sd = (np.asarray(a) for a in source_nodes)
    adm, dns = np.asarray(adm).reshape(-1, 4), np.asarray(dns).reshape(-1, 4)
    t_of = {(int(r[0]), int(r[1])): k for k, r in enumerate(ti)}
    s_of = {(int(r[0]), int(r[1])): k for k, r in enumerate(si)}

    def admissible(t, s):
        rt, rs = td[t, 3], sd[s, 3]
        dist = float(np.linalg.norm(td[t, :3] - sd[s, :3]))
        return 2.0 * min(rt, rs) < eta * max(0.0, dist - rt - rs)

    def stops(t, s):  # 1: low-rank leaf, 2: dense leaf, 0: the visit goes on
        if admissible(t, s) and ti[t, 2] >= min_target_depth and si[s, 2] >= min_source_depth:
            return 1
        if ti[t, 5] == 0 and si[s, 5] == 0:
            return 2
        return 0

    def rule(t, s):  # which side(s) a pair that does not stop is split on
        t_leaf, s_leaf = ti[t, 5] == 0, si[s, 5] == 0
        if s_leaf or (not t_leaf and ti[t, 1] > si[s, 1]):
            return "t"
        if t_leaf or si[s, 1] > ti[t, 1]:
            return "s"
        return "ts"

    memo = {(t_root, s_root): True}

    def reachable(t, s):
        key = (t, s)
        if key in memo:
            return memo[key]
        memo[key] = False  # (no cycles: every candidate is strictly higher in one of the trees)
        pt, ps = int(ti[t, 3]), int(si[s, 3])
        cands = []
        if t != t_root and pt >= 0:
            cands.append((pt, s, "t"))
        if s != s_root and ps >= 0:
            cands.append((t, ps, "s"))
        if t != t_root and s != s_root and pt >= 0 and ps >= 0:
            cands.append((pt, ps, "ts"))
        ok = any(stops(a, b) == 0 and rule(a, b) == how and reachable(a, b) for a, b, how in cands)
        memo[key] = ok
        return ok

    rows0, nrows = int(ti[t_root, 0]), int(ti[t_root, 1])
    cols0, ncols = int(si[s_root, 0]), int(si[s_root, 1])
    area = 0
    paint = np.zeros((nrows, ncols), dtype=np.uint8) if nrows * ncols <= paint_limit * paint_limit else None
    for kind, leaves in ((1, adm), (2, dns)):
        for t_off, m, s_off, n in leaves:
            t, s = t_of.get((int(t_off), int(m))), s_of.get((int(s_off), int(n)))
            assert t is not None and s is not None, f"leaf ({t_off},{m},{s_off},{n}) is not a pair of cluster nodes"
            assert stops(t, s) == kind, f"leaf ({t_off},{m},{s_off},{n}): the visit does not stop here as a {'low-rank' if kind == 1 else 'dense'} leaf"
            assert reachable(t, s), f"leaf ({t_off},{m},{s_off},{n}) is not reached by the visit from the root pair"
            assert rows0 <= t_off and t_off + m <= rows0 + nrows and cols0 <= s_off and s_off + n <= cols0 + ncols
            area += int(m) * int(n)
            if paint is not None:
                paint[t_off - rows0:t_off - rows0 + m, s_off - cols0:s_off - cols0 + n] += 1
    assert area == nrows * ncols, "the leaves do not cover the matrix exactly once (area)"
    if paint is not None:
        assert paint.min() == 1 and paint.max() == 1, "the leaves overlap or leave a gap"
    return len(adm), len(dns)
