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
