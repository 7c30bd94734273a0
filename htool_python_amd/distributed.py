"""Distributed-operator extension points of the reference, host-language side.

Reference surface mirrored here (all user-subclassable operators run on the host by definition -- they are
written in Python -- so there is nothing to accelerate; only the H-matrix parts run on the GPU):

  LocalRenumbering                       src/htool/local_operator/local_renumbering.hpp:8-21
  RestrictedGlobalToLocalOperator        src/htool/local_operator/local_operator.hpp:8-87
  VirtualLocalToLocalOperator            src/htool/local_operator/virtual_local_to_local_operator.hpp:8-99
  DistributedOperator                    src/htool/distributed_operator/distributed_operator.hpp:13-79
  DefaultApproximationBuilder            src/htool/distributed_operator/utility.hpp:25-32
  DefaultLocalApproximationBuilder       src/htool/distributed_operator/utility.hpp:34-41
  CustomApproximationBuilder             src/htool/distributed_operator/utility.hpp:19-23

A distributed operator is the sum of its registered operators; every rank holds the whole input vector
(replicated, user numbering), computes the rows of its target partition and the slices are all-gathered
(SURVEY.md 3.3).  The default H-matrix part goes through the C ABI (htool_distributed_*).
"""
import numpy as np

from . import Htool as _core


class LocalRenumbering:
    """(offset, size) window of a cluster tree's permutation."""

    def __init__(self, *args):
        if len(args) == 1:  # LocalRenumbering(cluster)
            cluster = args[0]
            self._offset, self._size = cluster.get_offset(), cluster.get_size()
            self._permutation = np.asarray(cluster.get_permutation())
        elif len(args) == 3:  # LocalRenumbering(offset, size, permutation)
            self._offset, self._size = int(args[0]), int(args[1])
            self._permutation = np.asarray(args[2])
        else:
            raise TypeError("LocalRenumbering(cluster) or LocalRenumbering(offset, size, permutation)")

    offset = property(lambda self: self._offset)
    size = property(lambda self: self._size)
    global_size = property(lambda self: len(self._permutation))
    is_stable = property(lambda self: True)
    permutation = property(lambda self: self._permutation)


class IGlobalToLocalOperator:
    pass


class IRestrictedGlobalToLocalOperator(IGlobalToLocalOperator):
    pass


class RestrictedGlobalToLocalOperator(IRestrictedGlobalToLocalOperator):
    """User operator acting on the source window `local_source_renumbering` (cluster numbering) and producing the
    rows of `local_target_renumbering`.  Subclasses implement add_vector_product / add_matrix_product_row_major
    (in-place on `out`)."""

    def __init__(self, local_target_renumbering, local_source_renumbering, target_use_permutation_to_mvprod=False, source_use_permutation_to_mvprod=False):
        self._t, self._s = local_target_renumbering, local_source_renumbering
        self._tperm, self._sperm = target_use_permutation_to_mvprod, source_use_permutation_to_mvprod

    local_target_renumbering = property(lambda self: self._t)
    local_source_renumbering = property(lambda self: self._s)

    def add_vector_product(self, trans, alpha, input, beta, output):  # pragma: no cover - pure virtual
        raise RuntimeError('Tried to call pure virtual function "RestrictedGlobalToLocalOperator::add_vector_product"')

    def add_matrix_product_row_major(self, trans, alpha, input, beta, output):  # pragma: no cover - pure virtual
        raise RuntimeError('Tried to call pure virtual function "RestrictedGlobalToLocalOperator::add_matrix_product_row_major"')


class ILocalToLocalOperator:
    pass


class VirtualLocalToLocalOperator(ILocalToLocalOperator):
    """User operator from this rank's source window to this rank's target rows."""

    def __init__(self, local_target_renumbering, local_source_renumbering):
        self._t, self._s = local_target_renumbering, local_source_renumbering

    local_target_renumbering = property(lambda self: self._t)
    local_source_renumbering = property(lambda self: self._s)

    def local_add_vector_product(self, trans, alpha, input, beta, output):  # pragma: no cover - pure virtual
        raise RuntimeError('Tried to call pure virtual function "VirtualLocalToLocalOperator::local_add_vector_product"')

    def local_add_matrix_product_row_major(self, trans, alpha, input, beta, output):  # pragma: no cover - pure virtual
        raise RuntimeError('Tried to call pure virtual function "VirtualLocalToLocalOperator::local_add_matrix_product_row_major"')


class _HMatrixLocalToLocal(ILocalToLocalOperator):
    """(target partition p) x (source partition q) H-matrix block living in HBM, as a local-to-local operator."""

    def __init__(self, hmatrix, target_renumbering, source_renumbering):
        self.hmatrix, self._t, self._s = hmatrix, target_renumbering, source_renumbering

    local_target_renumbering = property(lambda self: self._t)
    local_source_renumbering = property(lambda self: self._s)

    def local_add_vector_product(self, trans, alpha, input, beta, output):
        output *= beta
        if trans == "N":
            output += alpha * (self.hmatrix * np.ascontiguousarray(input))
        else:  # 'T' / 'C': the transposed sweeps of the same panels (include/htool_mi355x.h: htool_hmatrix_matvec)
            output += alpha * self.hmatrix.transposed_mul(np.ascontiguousarray(input), trans)

    def local_add_matrix_product_row_major(self, trans, alpha, input, beta, output):
        output *= beta
        if trans == "N":
            output += alpha * np.asarray(self.hmatrix @ np.asfortranarray(input))
        else:
            output += alpha * np.asarray(self.hmatrix.transposed_mul(np.asfortranarray(input), trans))


class DistributedOperator:
    """Sum of global-to-local and local-to-local operators over a row partition."""

    def __init__(self, target_cluster, source_cluster, comm, core=None, dtype=np.float64):
        self._tc, self._sc, self._comm, self._core, self._dtype = target_cluster, source_cluster, comm, core, dtype
        self._g2l, self._l2l = [], []
        self._rank, self._world = comm.Get_rank(), comm.Get_size()
        self._tperm = np.asarray(target_cluster.get_permutation())
        self._sperm = np.asarray(source_cluster.get_permutation())
        self._parts = [(target_cluster.get_cluster_on_partition(p).get_offset(), target_cluster.get_cluster_on_partition(p).get_size()) for p in range(self._world)]

    # -- reference API
    @property
    def shape(self):
        return (len(self._tperm), len(self._sperm))

    def add_global_to_local_operator(self, op):
        self._g2l.append(op)

    def add_local_to_local_operator(self, op):
        self._l2l.append(op)

    def __mul__(self, x):
        x = np.asarray(x)
        if x.ndim != 1:
            raise RuntimeError("Wrong dimension for DistributedOperator-vector product")
        if x.shape[0] != self.shape[1]:
            raise RuntimeError("Wrong size for DistributedOperator-vector product")
        return self._apply(x.astype(self._dtype, copy=False).reshape(-1, 1))[:, 0].copy()

    def __matmul__(self, X):
        X = np.asarray(X)
        if X.ndim != 2:
            raise RuntimeError("Wrong dimension for HMatrix-matrix product")
        if X.shape[0] != self.shape[1]:
            raise RuntimeError("Wrong size for HMatrix-matrix product")
        return np.asfortranarray(self._apply(X.astype(self._dtype, copy=False)))

    def internal_sub_vector_product_global_to_local(self, x_sub, offset):
        """Rows of this rank for an input that is zero outside [offset, offset+len) of the CLUSTER-numbered source
        vector (distributed_operator.hpp:67-78; tests/test_distributed_operator.py:105-129)."""
        x_sub = np.asarray(x_sub, dtype=self._dtype)
        if x_sub.ndim != 1:
            raise RuntimeError("Wrong dimension for DistributedOperator-vector product")
        xp = np.zeros((self.shape[1], 1), dtype=self._dtype)
        xp[offset:offset + len(x_sub), 0] = x_sub
        return self._local_rows(xp, xp_is_cluster=True)[:, 0].copy()

    # -- extensions used by the GPU-resident Krylov loop
    @property
    def local_hmatrix(self):
        return self._core.local_hmatrix if self._core is not None else None

    @property
    def comm(self):
        return self._comm

    def partition(self):
        return list(self._parts)

    @property
    def has_rccl(self):
        """True when the communicator carries a library-owned RCCL handle: matvec_device / matmat_device then exchange
        the vector slices inside the library (htool_distributed_matvec_device)."""
        return self._core is not None and self._core.has_rccl

    def exchange_kind(self, mu=1):
        """Which exchange matvec_device / matmat_device run inside the library (htool_distributed_exchange_kind): 0 none (one
        rank), 1 all-gather straight into the contiguous vector, 2 padded slices + compaction kernel, 3 / 4 the same with the
        all-gather staged through the host (communicator without RCCL handle: ranks sharing a GPU), -1 not available."""
        return self._core.exchange_kind(mu) if self.has_only_default_operator() else -1

    def matvec_device(self, x_local_ptr, y_local_ptr, stream=0):
        """GPU-resident product of the default operator: this rank's slice of x in, its rows of y out (device pointers,
        cluster numbering); the all-gather of the slices happens inside the library (RCCL, or host-staged: exchange_kind)."""
        if not self.has_only_default_operator():
            raise RuntimeError("matvec_device: only the default H-matrix operator has a device-resident product")
        self._core.matvec_device(x_local_ptr, y_local_ptr, stream)

    def matmat_device(self, x_local_ptr, ldx, y_local_ptr, ldy, mu, stream=0):
        if not self.has_only_default_operator():
            raise RuntimeError("matmat_device: only the default H-matrix operator has a device-resident product")
        self._core.matmat_device(x_local_ptr, ldx, y_local_ptr, ldy, mu, stream)

    def matmat_device_trans(self, trans, x_local_ptr, ldx, y_local_ptr, ldy, mu, stream=0):
        """The transposed ('T') or conjugate-transposed ('C') product with the same distribution (an extension: the reference's
        Python surface only passes 'N'): this rank's slice of x -- the rows it owns -- in, its slice of y (source partition
        `rank`) out, device pointers in cluster numbering; one reduce-scatter inside the library (htool_distributed_matmat_device_trans)."""
        if not self.has_only_default_operator():
            raise RuntimeError("matmat_device_trans: only the default H-matrix operator has a device-resident product")
        self._core.matmat_device_trans(trans, x_local_ptr, ldx, y_local_ptr, ldy, mu, stream)

    def has_only_default_operator(self):
        return self._core is not None and not self._g2l and not self._l2l

    # -- implementation
    def _local_rows(self, X, xp_is_cluster=False):
        """Rows of this rank (cluster order) of (sum of operators) X; X user-numbered (or cluster-numbered)."""
        off, size = self._parts[self._rank]
        mu = X.shape[1]
        Xp = X if xp_is_cluster else X[self._sperm]
        y_local = np.zeros((size, mu), dtype=self._dtype)
        if self._core is not None:
            H = self._core.local_hmatrix
            Xu = X
            if xp_is_cluster:  # the core takes user numbering on a whole-source operator
                Xu = np.empty_like(X)
                Xu[self._sperm] = X
            Yc = np.asarray(H @ np.asfortranarray(Xu)) if mu > 1 else np.asarray(H * np.ascontiguousarray(Xu[:, 0])).reshape(-1, 1)
            if self._world == 1:  # an operator built on the whole target cluster answers in user numbering
                Yc = Yc[self._tperm]
            y_local += Yc
        for op in self._g2l:
            s = op.local_source_renumbering
            inp = np.ascontiguousarray(Xp[s.offset:s.offset + s.size])
            if mu == 1:
                out = y_local[:, 0].copy()
                op.add_vector_product("N", 1.0, inp[:, 0].copy(), 1.0, out)
                y_local[:, 0] = out
            else:
                out = np.ascontiguousarray(y_local)
                op.add_matrix_product_row_major("N", 1.0, inp, 1.0, out)
                y_local[...] = out
        for op in self._l2l:
            s = op.local_source_renumbering
            inp = np.ascontiguousarray(Xp[s.offset:s.offset + s.size])
            if mu == 1:
                out = y_local[:, 0].copy()
                op.local_add_vector_product("N", 1.0, inp[:, 0].copy(), 1.0, out)
                y_local[:, 0] = out
            else:
                out = np.ascontiguousarray(y_local)
                op.local_add_matrix_product_row_major("N", 1.0, inp, 1.0, out)
                y_local[...] = out
        return y_local

    def _apply(self, X):
        if self.has_only_default_operator():  # plain default operator: the C-ABI path does everything
            return np.asarray(self._core @ np.asfortranarray(X)) if X.shape[1] > 1 else np.asarray(self._core * np.ascontiguousarray(X[:, 0])).reshape(-1, 1)
        y_local = self._local_rows(X)
        mu, es = X.shape[1], np.dtype(self._dtype).itemsize
        Yp = np.empty((self.shape[0], mu), dtype=self._dtype)
        counts = [s * es for _, s in self._parts]
        displs = [o * es for o, _ in self._parts]
        for c in range(mu):  # all-gather of the row slices
            send = np.ascontiguousarray(y_local[:, c]).view(np.uint8)
            recv = np.empty(self.shape[0] * es, dtype=np.uint8)
            self._comm._htool_allgatherv(send, recv, counts, displs)
            Yp[:, c] = recv.view(self._dtype)
        Y = np.empty_like(Yp)
        Y[self._tperm] = Yp
        return Y


def _dtype_of(builder):
    return np.complex128 if type(builder).__name__.startswith("Complex") else np.float64


class DefaultApproximationBuilder:
    """rows(partition rank) x all columns as one H-matrix (the reference's default)."""

    def __init__(self, generator, target_cluster, source_cluster, hmatrix_tree_builder, comm):
        cplx = _dtype_of(hmatrix_tree_builder) == np.complex128
        Core = _core.ComplexDefaultApproximationBuilder if cplx else _core.DefaultApproximationBuilder
        self._core = Core(generator, target_cluster, source_cluster, hmatrix_tree_builder, comm)
        self.distributed_operator = DistributedOperator(target_cluster, source_cluster, comm, self._core.distributed_operator, _dtype_of(hmatrix_tree_builder))
        self._hmatrix = self._core.hmatrix

    @property
    def hmatrix(self):
        return self._hmatrix

    @hmatrix.setter
    def hmatrix(self, value):  # utility.hpp:29 allows assigning (e.g. a recompressed copy); kept for API parity
        self._hmatrix = value

    @property
    def block_diagonal_hmatrix(self):
        return self._core.block_diagonal_hmatrix


class DefaultLocalApproximationBuilder:
    """(partition rank) x (partition rank) H-matrix as a local-to-local operator; the off-diagonal parts are added
    by the user as global-to-local operators (tests/conftest.py:352-365)."""

    def __init__(self, generator, target_cluster, source_cluster, hmatrix_tree_builder, comm):
        rank = comm.Get_rank()
        self.hmatrix = hmatrix_tree_builder.build_local(generator, target_cluster, source_cluster, rank, rank)
        self.distributed_operator = DistributedOperator(target_cluster, source_cluster, comm, None, _dtype_of(hmatrix_tree_builder))
        self._op = _HMatrixLocalToLocal(self.hmatrix, LocalRenumbering(target_cluster.get_cluster_on_partition(rank)),
                                        LocalRenumbering(source_cluster.get_cluster_on_partition(rank)))
        self.distributed_operator.add_local_to_local_operator(self._op)

    @property
    def block_diagonal_hmatrix(self):
        return self.hmatrix


class CustomApproximationBuilder:
    """Distributed operator around ONE user operator (local-to-local or global-to-local)."""

    def __init__(self, target_cluster, source_cluster, comm, operator, dtype=np.float64):
        self.distributed_operator = DistributedOperator(target_cluster, source_cluster, comm, None, dtype)
        if isinstance(operator, ILocalToLocalOperator):
            self.distributed_operator.add_local_to_local_operator(operator)
        else:
            self.distributed_operator.add_global_to_local_operator(operator)
