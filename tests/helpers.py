"""Shared helpers of the parity tests (host-language mirrors of the reference's example generators)."""
import numpy as np

import Htool
from oracle import oracle as O


class NumpyGenerator(Htool.VirtualGenerator):
    """Callback generator: same kernel as the reference's CustomGenerator
    (example/define_generators.py:6-27), evaluated block-wise with numpy instead of per entry."""

    def __init__(self, target_points, source_points, kind=O.K_INV_DELTA, p0=0.1):
        super().__init__()
        self.target_points = target_points
        self.source_points = source_points
        self.kind, self.p0 = kind, p0
        self.nb_rows = target_points.shape[1]
        self.nb_cols = source_points.shape[1]

    def build_submatrix(self, J, K, mat):
        mat[:, :] = O.kernel_block(self.kind, self.target_points[:, J], self.source_points[:, K], self.p0)

    def mat_vec(self, x):
        return O.dense_matvec(self.kind, self.target_points, self.source_points, x, self.p0)

    def mat_mat(self, X):
        return O.dense_matvec(self.kind, self.target_points, self.source_points, X, self.p0)


class ComplexNumpyGenerator(Htool.ComplexVirtualGenerator):
    def __init__(self, target_points, source_points, kappa):
        super().__init__()
        self.target_points, self.source_points, self.kappa = target_points, source_points, kappa

    def build_submatrix(self, J, K, mat):
        mat[:, :] = O.kernel_block(O.K_HELMHOLTZ, self.target_points[:, J], self.source_points[:, K], self.kappa)

    def mat_vec(self, x):
        return O.dense_matvec(O.K_HELMHOLTZ, self.target_points, self.source_points, x, self.kappa)


class CustomSVD(Htool.VirtualLowRankGenerator):
    """User compressor: truncated SVD of the block.  Acceptance rule as in the reference's example compressor
    (example/advanced/define_custom_low_rank_generator.py:16-27): trailing singular values are dropped while the
    discarded energy stays below (epsilon |A|_F)^2, and the block is refused when r (m + n) > m n."""

    def __init__(self, generator, allow_copy=True):
        super().__init__(allow_copy)
        self.generator = generator

    def build_low_rank_approximation(self, rows, cols, epsilon):
        m, n = len(rows), len(cols)
        block = np.zeros((m, n), order="F")
        self.generator.build_submatrix(rows, cols, block)
        left, sigma, right = np.linalg.svd(block, full_matrices=False)
        # tail[k] = energy of sigma[k:]; keep the smallest k >= 1 (scanning from the end, one value at a time, and
        # always discarding the last one, like the reference's loop) with sqrt(tail) / |A|_F < epsilon
        discarded, rank = 0.0, len(sigma)
        while rank > 1 and np.sqrt(discarded) / np.linalg.norm(block) < epsilon:
            rank -= 1
            discarded += sigma[rank] ** 2
        if rank * (m + n) > m * n:
            return False
        self.set_U(left[:, :rank] * sigma[:rank])
        self.set_V(right[:rank, :])
        return True


class CustomDenseBlocksGenerator(Htool.VirtualDenseBlocksGenerator):
    """example/advanced/define_custom_dense_blocks_generator.py"""

    def __init__(self, generator, target_cluster, source_cluster):
        super().__init__(target_cluster, source_cluster)
        self.generator = generator

    def build_dense_blocks(self, rows_offsets, cols_offsets, blocks):
        for i in range(len(blocks)):
            self.generator.build_submatrix(rows_offsets[i], cols_offsets[i], blocks[i])


def cluster_of(points, max_leaf=10, children=2, **kw):
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(max_leaf)
    return b.create_cluster_tree(points, children, **kw)


def cpu_leaf_loop(hmatrix, x_user, is_complex=False):
    """Download every leaf's panels from HBM (one bulk call) and run the oracle's CPU leaf loop on them:
    y (user numbering) = sum over leaves, on panels IDENTICAL to what the HIP product streams."""
    leaves = np.asarray(hmatrix.leaves())
    tc, sc = hmatrix.get_target_cluster(), hmatrix.get_source_cluster()
    pt, ps = np.asarray(tc.get_permutation()), np.asarray(sc.get_permutation())
    dt = np.complex128 if is_complex else np.float64
    sel, offs, panels = O.leaf_sample_panels(hmatrix, np.arange(len(leaves)), is_complex)
    xp = np.asarray(x_user, dtype=dt)[ps]
    yp = O.leaf_loop(sel, offs, panels, len(pt), xp, is_complex)
    y = np.zeros(len(pt), dtype=dt)
    y[pt] = yp
    return y


class CustomRestrictedGlobalToLocalOperator(Htool.RestrictedGlobalToLocalOperator):
    """Dense extra-diagonal block as a user operator (example/advanced/define_custom_local_operator.py:6-50)."""

    def __init__(self, generator, target_local_renumbering, source_local_renumbering, target_use_permutation_to_mvprod=False, source_use_permutation_to_mvprod=False):
        super().__init__(target_local_renumbering, source_local_renumbering, target_use_permutation_to_mvprod, source_use_permutation_to_mvprod)
        t, s = target_local_renumbering, source_local_renumbering
        self.data = np.zeros((t.size, s.size), order="F")
        generator.build_submatrix(t.permutation[t.offset:t.offset + t.size], s.permutation[s.offset:s.offset + s.size], self.data)

    def add_vector_product(self, trans, alpha, input, beta, output):
        output *= beta
        output += alpha * (self.data if trans == "N" else self.data.T).dot(input)

    def add_matrix_product_row_major(self, trans, alpha, input, beta, output):
        output *= beta
        output += alpha * (self.data if trans == "N" else self.data.T) @ input


class CustomLocalToLocalOperator(Htool.VirtualLocalToLocalOperator):
    """Dense diagonal block as a user operator (example/advanced/define_custom_local_operator.py:53-100)."""

    def __init__(self, generator, target_local_renumbering, source_local_renumbering):
        super().__init__(target_local_renumbering, source_local_renumbering)
        t, s = target_local_renumbering, source_local_renumbering
        self.data = np.zeros((t.size, s.size), order="F")
        generator.build_submatrix(t.permutation[t.offset:t.offset + t.size], s.permutation[s.offset:s.offset + s.size], self.data)

    def local_add_vector_product(self, trans, alpha, input, beta, output):
        output *= beta
        output += alpha * (self.data if trans == "N" else self.data.T).dot(input)

    def local_add_matrix_product_row_major(self, trans, alpha, input, beta, output):
        output *= beta
        output += alpha * (self.data if trans == "N" else self.data.T) @ input


def independent_leaf_checks(hmatrix, points_t, points_s, kind, p0, eps, n_sample=200, seed=0, max_block=600, transpose_rule=True, min_leaves=50):
    """Independent (numpy, oracle/independent.py) checks of the low-rank leaves of a built H-matrix, none of which goes
    through the C++ oracle: for a random sample of admissible leaves
      * rank vs the explicit-residual ACA of the exact block (same pivots => same rank, +-1 on borderline leaves),
      * rank <= SVD-rank(eps / 10) + 2 on >= 97 % of the sample, + 4 at most (the reference's reading of epsilon, define_custom_low_rank_generator.py:16-27),
      * |A - U V|_F / |A|_F against eps (partial pivoting stops on a heuristic: most leaves within 3 eps, all within 10 or where
        the explicit-residual ACA of the same block is as far off).
    Returns a dict of the statistics that were asserted."""
    from oracle import independent as I

    L = np.asarray(hmatrix.leaves()).astype(np.int64)
    pt = np.asarray(hmatrix.get_target_cluster().get_permutation())
    ps = np.asarray(hmatrix.get_source_cluster().get_permutation())
    cand = np.flatnonzero((L[:, 4] > 0) & (L[:, 1] <= max_block) & (L[:, 3] <= max_block))
    rng = np.random.RandomState(seed)
    pick = rng.choice(cand, min(n_sample, len(cand)), replace=False)
    if len(pick) == 0:  # (tiny blocks at a tight tolerance: every admissible block was refused, r (m + n) > m n)
        assert min_leaves == 0, "no low-rank leaf to check"
        return None
    same = pm1 = 0
    errs, ref_errs, over_svd, e2, a2 = [], [], [], 0.0, 0.0
    for i in pick:
        t_off, m, s_off, n, r = L[i]
        A = O.kernel_block(kind, points_t[:, pt[t_off:t_off + m]], points_s[:, ps[s_off:s_off + n]], p0)
        U, V = hmatrix.leaf_panels(int(i))
        U, V = np.asarray(U), np.asarray(V)
        assert U.shape == (m, r) and V.shape == (r, n)
        ref = I.aca_full_residual(A, eps, transpose_role=transpose_rule and t_off > s_off)
        assert ref is not None, "the explicit-residual ACA rejects a leaf the engine stored as low rank"
        same += ref[0].shape[1] == r
        pm1 += abs(ref[0].shape[1] - r) <= 1
        err, _, rs = I.leaf_quality(A, U, V, eps)
        errs.append(err / eps)
        # the same block through the independent formulation: a leaf may only exceed 10 eps where partial pivoting itself does
        ref_errs.append(np.linalg.norm(A - ref[0] @ ref[1]) / max(np.linalg.norm(A), 1e-300) / eps)
        over_svd.append(r - rs)
        e2 += np.linalg.norm(A - U @ V) ** 2
        a2 += np.linalg.norm(A) ** 2
    errs, ref_errs, over_svd = np.array(errs), np.array(ref_errs), np.array(over_svd)
    k = len(pick)
    stats = {"leaves": k, "same_rank": same / k, "within_one": pm1 / k, "err_over_eps_max": float(errs.max()), "err_over_eps_p90": float(np.percentile(errs, 90)),
             "rank_minus_svd_rank_max": int(over_svd.max()), "aggregate_err_over_eps": float(np.sqrt(e2 / a2) / eps)}
    assert k >= min(n_sample, min_leaves), f"only {k} admissible leaves to sample"
    assert stats["same_rank"] >= 0.97 and stats["within_one"] == 1.0, stats
    assert np.mean(over_svd <= 2) >= 0.97 and over_svd.max() <= 4, stats  # (a property of partial pivoting: the explicit-residual ACA has the same ranks)
    assert np.mean(errs <= 3.0) >= 0.9 and np.mean(errs <= 10.0) >= 0.99 and stats["aggregate_err_over_eps"] <= 2.0, stats
    assert np.all((errs <= 10.0) | (errs <= 1.05 * ref_errs)), stats  # (the heuristic stop of partial pivoting, not the engine)
    return stats


def single_leaf_product_checks(hmatrix, cluster_t, cluster_s, dtype, n_sources=3, n_targets=12, seed=0, max_leaf_elements=4e7, row_window=None):
    """CPU leaf loop on a sampled subset of the device's OWN panels, isolated through the tiling property: with x supported
    on one source cluster leaf s, the rows of a target cluster leaf t receive the contribution of exactly ONE H-matrix leaf
    (the one containing t x s), so   y[t] == U[t, :] (V[:, s] x_s)   (or D[t, s] x_s) on that leaf's downloaded panels.
    Works at any size (one device product per source leaf).  row_window = (offset, size): operator built on a partition,
    whose host product returns its local rows in cluster order.  Returns the number of (t, s) pairs compared."""
    L = np.asarray(hmatrix.leaves()).astype(np.int64)
    ti, _ = cluster_t._nodes()
    si, _ = cluster_s._nodes()
    pt, ps = np.asarray(cluster_t.get_permutation()), np.asarray(cluster_s.get_permutation())
    r_off, r_size = row_window if row_window is not None else (0, len(pt))
    tl = np.array([(o, s) for o, s, _, _, _, nc, _ in ti if nc == 0 and r_off <= o < r_off + r_size])
    sl = np.array([(o, s) for o, s, _, _, _, nc, _ in si if nc == 0])
    rng = np.random.RandomState(seed)
    n_src = len(ps)
    done = 0
    for so, ss in sl[rng.choice(len(sl), n_sources, replace=False)]:
        xs = rng.rand(ss) + (1j * rng.rand(ss) if np.dtype(dtype).kind == "c" else 0)
        x = np.zeros(n_src, dtype=dtype)
        x[ps[so:so + ss]] = xs
        y = hmatrix * x
        scale = np.abs(y).max()
        for to, ts in tl[rng.choice(len(tl), n_targets, replace=False)]:
            hit = np.flatnonzero((L[:, 0] <= to) & (to + ts <= L[:, 0] + L[:, 1]) & (L[:, 2] <= so) & (so + ss <= L[:, 2] + L[:, 3]))
            assert len(hit) == 1, "the leaves do not tile the matrix"
            t_off, m, s_off, n, r = L[hit[0]]
            if (m * n if r < 0 else r * (m + n)) > max_leaf_elements:
                continue
            A, B = hmatrix.leaf_panels(int(hit[0]))
            A = np.asarray(A)
            if r < 0:
                expect = A[to - t_off:to - t_off + ts, so - s_off:so - s_off + ss] @ xs
            else:
                expect = A[to - t_off:to - t_off + ts, :] @ (np.asarray(B)[:, so - s_off:so - s_off + ss] @ xs)
            got = y[to - r_off:to - r_off + ts] if row_window is not None else y[pt[to:to + ts]]
            assert np.abs(got - expect).max() <= 1e-12 * scale + 1e-300, (to, so, r, np.abs(got - expect).max(), scale)
            done += 1
    assert done >= n_sources * n_targets // 2
    return done
