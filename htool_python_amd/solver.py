"""Krylov solver behind the reference's solver surface.

Reference: `Htool.DDMSolverBuilder(distributed_operator, block_diagonal_hmatrix).solver`, then
`solver.facto_one_level()`, `solver.set_hpddm_args(str)`, `solver.solve(x, b, hpddm_args)`,
`solver.get_information()` (src/htool/solver/utility.hpp:7-61, src/htool/solver/solver.hpp:16-66,
example/use_ddm_solver.py:49-69).  The reference delegates to HPDDM (absent); here the Krylov loop is the
package's own restarted GMRES on GPU-resident vectors (krylov.py) with the H-matrix product as operator.
Only the options the reference's tests use are parsed (tests/test_ddm_solver.py:550-558):
-hpddm_krylov_method gmres, -hpddm_tol, -hpddm_max_it, -hpddm_gmres_restart, -hpddm_variant right.
`facto_one_level()` sets up the one-level right preconditioner.  As in the reference it is the inverse of the rank's whole
diagonal block `block_diagonal_hmatrix` (one-level Schwarz without overlap) -- applied through the HIERARCHICAL device LU of
that block (BlockLU -> csrc/hlu_device.hip, round 4; a dense device factorisation for operators it does not cover);
otherwise (or without that block) block-Jacobi from the DENSE DIAGONAL LEAVES of this rank's H-matrix:
the leaves (t, t) of the cluster-tree leaves tile the diagonal; they are downloaded once, LU-factorised as one padded batch
(library call).  GenEO coarse spaces are out of scope.
"""
import time

import numpy as np
import torch


def _parse_hpddm(args, opts):
    tok = args.split()
    i = 0
    while i < len(tok):
        key = tok[i]
        val = tok[i + 1] if i + 1 < len(tok) and not tok[i + 1].startswith("-hpddm") else None
        if key == "-hpddm_tol" and val:
            opts["tol"] = float(val)
        elif key == "-hpddm_max_it" and val:
            opts["max_it"] = int(val)
        elif key == "-hpddm_gmres_restart" and val:
            opts["restart"] = int(val)
        elif key == "-hpddm_krylov_method" and val:
            if val != "gmres":
                raise ValueError(f"only -hpddm_krylov_method gmres is implemented (got {val})")
        i += 2 if val is not None else 1
    return opts


class DeviceOperator:
    """y_local = (shift I + A) x on GPU-resident slices in cluster numbering; A = this rank's rows."""

    def __init__(self, hmatrix, partition=None, rank=0, group=None, shift=0.0, dist_op=None):
        # dist_op: the DistributedOperator this H-matrix belongs to; the exchange + product of apply() is then ONE library call
        # (htool_distributed_matvec_device: RCCL all-gather when the communicator carries a library-owned handle, the same
        # code path staged through the host all-gather of the communicator object when ranks share a GPU)
        self.dist_op = dist_op if self._library_exchange_is_the_fast_one(dist_op) else None
        self.H = hmatrix
        self.n = hmatrix.shape[1]
        self.partition = partition or [(0, self.n)]
        self.rank, self.world = rank, len(self.partition)
        self.group, self.shift = group, shift
        self.offset, self.size = self.partition[rank]
        self.products = 0
        self._gather = None

    @staticmethod
    def _library_exchange_is_the_fast_one(dist_op):
        """The in-library exchange is taken when it runs on device buffers (kinds 0, 1, 2: one rank, or a communicator with a
        device all-gather hook -- RCCL) or when the process group is gloo anyway (ranks sharing a GPU: the rehearsal path).  Under
        an nccl process group WITHOUT a library-owned RCCL handle (nobody called comm.use_rccl() before the operator was built)
        the library would stage every product through the host -- device-to-host copy, stream synchronisation, a Python
        all-gather callback, host-to-device copy: the SliceGatherer's device all-gather over torch.distributed (also RCCL)
        stays the exchange then."""
        if dist_op is None or not hasattr(dist_op, "exchange_kind"):
            return False
        kind = dist_op.exchange_kind(1)
        if kind < 0:
            return False
        if kind <= 2 or getattr(dist_op, "has_rccl", False):
            return True
        import torch.distributed as tdist

        return not tdist.is_initialized() or tdist.get_backend() == "gloo"

    def apply(self, x_local, out=None):
        """x_local: this rank's slice, shape (size,) or (mu, size) (rows may be strided: a slot of a Krylov basis); the result goes
        to `out` (same shapes) when given."""
        if x_local.dim() == 2:
            return self._apply_multi(x_local, out)
        y = torch.empty(self.size, dtype=x_local.dtype, device=x_local.device) if out is None else out
        assert y.is_contiguous() and y.numel() == self.size
        stream = torch.cuda.current_stream().cuda_stream
        if self.dist_op is not None:
            self.dist_op.matvec_device(x_local.contiguous().data_ptr(), y.data_ptr(), stream)
        else:
            xf = x_local.contiguous() if self.world == 1 else self._gathered(x_local)
            self.H.matvec_device(xf.data_ptr(), y.data_ptr(), 1, stream)
        self.products += 1
        if self.shift != 0.0:
            y.add_(x_local, alpha=self.shift)
        return y

    apply.supports_out = True

    def _gathered(self, x_local):
        if self._gather is None:
            from .comm import SliceGatherer

            self._gather = SliceGatherer([s for _, s in self.partition], x_local.dtype, x_local.device, self.group)
        return self._gather(x_local)

    def _apply_multi(self, X, out=None):
        """mu right-hand sides in one call (one exchange, sweeps of 8 / 16 columns over the panels): X (mu, size), row c = column c."""
        mu = X.shape[0]
        if X.stride(1) != 1:
            X = X.contiguous()
        Y = torch.empty(mu, self.size, dtype=X.dtype, device=X.device) if out is None else out
        assert Y.stride(1) == 1 and Y.shape == (mu, self.size)
        stream = torch.cuda.current_stream().cuda_stream
        ldx = X.stride(0) if mu > 1 else max(self.size, 1)
        ldy = Y.stride(0) if mu > 1 else max(self.size, 1)
        if self.dist_op is not None:
            self.dist_op.matmat_device(X.data_ptr(), ldx, Y.data_ptr(), ldy, mu, stream)
        elif self.world == 1:
            self.H.matmat_device(X.data_ptr(), ldx, Y.data_ptr(), ldy, mu, 1, stream)
        else:
            for c in range(mu):
                self.H.matvec_device(self._gathered(X[c]).data_ptr(), Y[c].data_ptr(), 1, stream)
        self.products += mu
        if self.shift != 0.0:
            Y.add_(X, alpha=self.shift)
        return Y

    def reduce(self, t):
        if self.world > 1:
            from .comm import all_reduce_sum

            all_reduce_sum(t, self.group)
        return t


class BlockJacobi:
    """M^-1 = blockdiag(D_1, ..., D_q)^-1 with D_i the dense diagonal leaves (+ shift I) of the local rows."""

    def __init__(self, hmatrix, offset, size, shift=0.0):
        leaves = np.asarray(hmatrix.leaves())
        ids = [i for i, l in enumerate(leaves) if l[4] < 0 and l[0] == l[2] and l[1] == l[3] and offset <= l[0] < offset + size]
        ids.sort(key=lambda i: leaves[i][0])
        cover = sum(int(leaves[i][1]) for i in ids)
        if cover != size:
            raise RuntimeError(f"block-Jacobi: the dense diagonal leaves cover {cover} of {size} local rows")
        offs, data = hmatrix.leaf_panels_bulk(np.asarray(ids, dtype=np.int64))
        data = np.asarray(data)
        sizes = [int(leaves[i][1]) for i in ids]
        bmax = max(sizes)
        blocks = np.zeros((len(ids), bmax, bmax), dtype=data.dtype)
        index = np.empty(size, dtype=np.int64)
        for q, i in enumerate(ids):
            m = sizes[q]
            blocks[q, :m, :m] = data[offs[q, 0]: offs[q, 0] + m * m].reshape(m, m).T  # column-major block
            blocks[q, np.arange(m), np.arange(m)] += shift
            blocks[q, np.arange(m, bmax), np.arange(m, bmax)] = 1.0                  # padding: identity
            o = int(leaves[i][0]) - offset
            index[o: o + m] = q * bmax + np.arange(m)
        self.index = torch.from_numpy(index).cuda()
        self.lu, self.piv = torch.linalg.lu_factor(torch.from_numpy(blocks).cuda())
        self.nb, self.bmax = len(ids), bmax

    def __call__(self, v):
        """v: (size,) or (mu, size) -- the blocks are solved for all columns at once."""
        if v.dim() == 2:
            mu = v.shape[0]
            pad = torch.zeros(mu, self.nb * self.bmax, dtype=v.dtype, device=v.device)
            pad[:, self.index] = v
            sol = torch.linalg.lu_solve(self.lu, self.piv, pad.view(mu, self.nb, self.bmax).permute(1, 2, 0).contiguous())
            return sol.permute(2, 0, 1).reshape(mu, -1)[:, self.index]
        pad = torch.zeros(self.nb * self.bmax, dtype=v.dtype, device=v.device)
        pad[self.index] = v
        sol = torch.linalg.lu_solve(self.lu, self.piv, pad.view(self.nb, self.bmax, 1))
        return sol.reshape(-1)[self.index]


class BlockLU:
    """M^-1 = (the rank's whole diagonal block + shift I)^-1: what the reference's `facto_one_level()` applies (one-level Schwarz
    without overlap: the H-LU of `block_diagonal_hmatrix`, example/use_ddm_solver.py:48-63).  The block is factorised
    HIERARCHICALLY on the device (htool_hmatrix_lu_factorization_shifted -> csrc/hlu_device.hip; round 4) and applied by the
    factorisation's two leaf sweeps on the Krylov loop's device vectors.  Operators the hierarchical factorisation does not
    cover (complex, tolerances below 1e-7) get a dense copy factorised by the dense solver library instead (dense_device.hip),
    which raises when that copy does not fit the device."""

    def __init__(self, block_hmatrix, shift=0.0):
        n, m = block_hmatrix.shape
        if n != m:
            raise RuntimeError("BlockLU: the diagonal block is not square")
        self.H, self.n = block_hmatrix, n
        block_hmatrix.lu_factorization_shifted(float(shift))
        self.kind = block_hmatrix.factorization_info()["kind"]

    def __call__(self, v):
        """v: (size,) or (mu, size), this rank's slice in cluster numbering."""
        out = v.clone() if v.dim() == 1 or v.stride(1) == 1 else v.contiguous().clone()
        mu = 1 if out.dim() == 1 else out.shape[0]
        ldb = self.n if out.dim() == 1 else out.stride(0)
        self.H.factor_solve_device(1, "N", out.data_ptr(), max(ldb, 1), mu, torch.cuda.current_stream().cuda_stream)
        return out


DenseBlockLU = BlockLU  # (the name of round 3, when the factorisation was dense only)


class Solver:
    def __init__(self, distributed_operator=None, hmatrix=None, shift=0.0, block_diagonal_hmatrix=None):
        from .krylov import gmres

        self._gmres = gmres
        self._opts = {"tol": 1e-6, "max_it": 200, "restart": 50}
        self._info = {}
        self._precond = None
        self._block = block_diagonal_hmatrix
        if distributed_operator is not None:
            # the Krylov operator is the default H-matrix part only: an operator with extra user terms (add_global_to_local_operator /
            # add_local_to_local_operator, tests/conftest.py "ExtraDiagonal") or without an H-matrix core would be solved wrongly
            if not distributed_operator.has_only_default_operator():
                raise RuntimeError("Solver: only a DistributedOperator made of its default H-matrix part (DefaultApproximationBuilder, no extra "
                                   "global-to-local / local-to-local operators) can be solved on the HIP path")
            H = distributed_operator.local_hmatrix
            comm = distributed_operator.comm
            part = distributed_operator.partition()
            rank = comm.Get_rank()
            group = None
            if len(part) > 1:
                comm._dist()  # make sure the process group exists
            self.op = DeviceOperator(H, part, rank, group, shift, dist_op=distributed_operator)
        else:
            self.op = DeviceOperator(hmatrix, None, 0, None, shift)
        assert self.op.H.shape[1] == sum(s for _, s in self.op.partition), "GMRES needs a square operator"
        self._perm = np.asarray(self.op.H.get_source_cluster().get_permutation())
        if len(self._perm) != self.op.H.shape[1]:  # a partition-built block: vectors are its slice in cluster order already
            self._perm = np.arange(self.op.H.shape[1])

    def facto_one_level(self):
        """One-level preconditioner (src/htool/solver/solver.hpp: facto_one_level).  With the rank's `block_diagonal_hmatrix` at hand:
        the inverse of that whole block, as the reference, through its hierarchical LU on the device (BlockLU).  Without it, or when
        the block cannot be factorised: block-Jacobi on the dense diagonal leaves of the local rows."""
        import logging

        blk = self._block
        if blk is not None and blk.shape == (self.op.size, self.op.size):
            try:
                self._precond = BlockLU(blk, self.op.shift)
                self._precond_name = "one-level: %s LU of block_diagonal_hmatrix" % {"hierarchical": "hierarchical device", "dense device": "dense device"}.get(self._precond.kind, self._precond.kind)
                return
            except RuntimeError as e:  # (neither the hierarchical factorisation nor a dense copy: say so, precondition with less)
                logging.getLogger("Htool").warning("facto_one_level: block_diagonal_hmatrix (%d unknowns) could not be factorised (%s) -- block-Jacobi on the dense diagonal "
                                                   "leaves instead", self.op.size, e)
        self._precond = BlockJacobi(self.op.H, self.op.offset, self.op.size, self.op.shift)
        self._precond_name = "block-jacobi (dense diagonal leaves)"

    def _dtype(self):
        return torch.complex128 if type(self.op.H).__name__.startswith("Complex") else torch.float64

    def build_coarse_space(self, *a, **k):
        raise RuntimeError("GenEO coarse spaces are outside the MI355X hot path; not implemented")

    def set_hpddm_args(self, hpddm_args):
        _parse_hpddm(hpddm_args, self._opts)

    def solve(self, x, b, hpddm_args=""):
        """x (in/out, numpy, user numbering) <- solution of A x = b; several right-hand sides are solved in lockstep (one product
        call per iteration for all of them)."""
        _parse_hpddm(hpddm_args, self._opts)
        if b.ndim != x.ndim or (b.ndim == 2 and b.shape[1] != x.shape[1]) or b.ndim > 2:
            raise ValueError(f"Wrong dimension for right-hand side or solution\nright-hand side: {b.shape}\nsolution: {x.shape}\n")
        off, size = self.op.offset, self.op.size
        t0 = time.time()
        B2 = np.asarray(b).reshape(len(self._perm), -1)   # columns = right-hand sides
        X2 = np.asarray(x).reshape(len(self._perm), -1)
        mu = B2.shape[1]
        # all right-hand sides in lockstep: one product call per iteration for the whole block (krylov.gmres, batched form)
        bl = torch.from_numpy(np.ascontiguousarray(B2[self._perm][off:off + size].T)).cuda()
        x0 = None
        if np.any(X2 != 0):
            x0 = torch.from_numpy(np.ascontiguousarray(X2[self._perm][off:off + size].T)).cuda()
        red = self.op.reduce if self.op.world > 1 else None
        if mu == 1:
            xl, info = self._gmres(self.op.apply, bl[0], None if x0 is None else x0[0], self._opts["tol"], self._opts["restart"], self._opts["max_it"], red, precond=self._precond)
            xl = xl.unsqueeze(0)
            hist = [info["residuals"]]
        else:
            xl, info = self._gmres(self.op.apply, bl, x0, self._opts["tol"], self._opts["restart"], self._opts["max_it"], red, precond=self._precond)
            hist = info["residuals"]
        full = self._gather(xl)          # (mu, n) in cluster numbering
        out = np.empty_like(full)
        out[:, self._perm] = full
        if b.ndim == 1:
            x[...] = out[0]
        else:
            x[...] = out.T
        self._history = hist[-1]
        res = [h[-1] if h else 0.0 for h in hist]
        self._info = {"Nb_it": str(info["iterations"]), "Relative_residual": str(max(res)), "Solve_seconds": str(time.time() - t0),
                      "Products": str(self.op.products), "Krylov_method": "gmres",
                      "Preconditioner": "none" if self._precond is None else getattr(self, "_precond_name", "block-jacobi (dense diagonal leaves)")}

    def _gather(self, xl):
        """(mu, local size) device -> (mu, n) host, cluster numbering."""
        if self.op.world == 1:
            return xl.cpu().numpy()
        from .comm import SliceGatherer

        gather = SliceGatherer([s for _, s in self.op.partition], xl.dtype, xl.device, self.op.group)
        return np.stack([gather(xl[c]).cpu().numpy() for c in range(xl.shape[0])])

    def get_information(self):
        return dict(self._info)


class DDMSolverBuilder:
    """Htool.DDMSolverBuilder(distributed_operator, block_diagonal_hmatrix).solver (src/htool/solver/utility.hpp:7-61).

    Only the two-argument form of the reference (utility.hpp:14) is on the HIP path.  The reference factorises
    `block_diagonal_hmatrix` hierarchically in `facto_one_level()` (one-level Schwarz without overlap = block-Jacobi with the
    rank's whole diagonal block); so does `facto_one_level()` here (BlockLU: the hierarchical LU on the device, round 4; a dense
    device factorisation for operators it does not cover), falling back -- with a WARNING -- to block-Jacobi on the dense diagonal
    leaves of the rank's rows when neither works.  The
    overlapping forms (utility.hpp:16-40: subdomain numberings, neighbours, intersections) need HPDDM's Schwarz machinery and
    are refused."""

    def __init__(self, distributed_operator, block_diagonal_hmatrix=None, *args, **kwargs):
        if args or kwargs:
            raise RuntimeError("DDMSolverBuilder: only DDMSolverBuilder(distributed_operator, block_diagonal_hmatrix) is implemented on the HIP path; "
                               "the overlapping-subdomain constructors (src/htool/solver/utility.hpp:16-40) are outside it")
        self.solver = Solver(distributed_operator, block_diagonal_hmatrix=block_diagonal_hmatrix)
        self._local_hmatrix = block_diagonal_hmatrix
        self.local_to_global_numbering = None

    def get_local_hmatrix(self):
        return self._local_hmatrix


# (src/htool/solver/utility.hpp:46-60, solver.hpp:68-69: the variants whose local solver is dense LAPACK -- on this engine the
# one-level preconditioner is dense block-Jacobi either way, so they are the same classes)
DDMSolverWithDenseLocalSolver = DDMSolverBuilder
SolverDense = Solver
