"""`import Htool` -- the reference's module name (src/htool/main.cpp:40), served by the MI355X engine.

The compiled pybind11 shim lives in htool_python_amd/ (built in-tree by htool_python_amd.build);
this package only re-exports it and adds the pure-Python plotting helpers.  There is no CPU
fallback: if the extension is missing the import fails.
"""
from htool_python_amd.Htool import *  # noqa: F401,F403
from htool_python_amd.Htool import __doc__ as _core_doc  # noqa: F401
from htool_python_amd.io import load_hmatrix, read_cluster_from, save_cluster_to, save_hmatrix  # noqa: F401
from htool_python_amd.plotting import plot  # noqa: F401
from htool_python_amd.solver import DDMSolverBuilder, DDMSolverWithDenseLocalSolver, Solver, SolverDense  # noqa: F401,E402

# distributed-operator surface: the user-extensible pieces live in Python (htool_python_amd/distributed.py) and wrap
# the C-ABI-backed default operator; both coefficient types share the implementation
from htool_python_amd.distributed import (  # noqa: F401,E402
    CustomApproximationBuilder,
    DefaultApproximationBuilder,
    DefaultLocalApproximationBuilder,
    DistributedOperator,
    IGlobalToLocalOperator,
    ILocalToLocalOperator,
    IRestrictedGlobalToLocalOperator,
    LocalRenumbering,
    RestrictedGlobalToLocalOperator,
    VirtualLocalToLocalOperator,
)

# complex twins (src/htool/main.cpp:89-110).  The Python-level classes handle both coefficient types, so the
# `Complex`-prefixed names of the reference are the same objects.
ComplexDDMSolverBuilder = DDMSolverBuilder
ComplexDDMSolverWithDenseLocalSolver = DDMSolverWithDenseLocalSolver   # solver/utility.hpp:46 with prefix "Complex" (main.cpp:110)
ComplexSolver = Solver                                                 # main.cpp:103
ComplexSolverDense = SolverDense                                       # solver/solver.hpp:69 with className "ComplexSolver"
ComplexIGlobalToLocalOperator = IGlobalToLocalOperator                 # local_operator/local_operator.hpp:75 (main.cpp:98)
ComplexIRestrictedGlobalToLocalOperator = IRestrictedGlobalToLocalOperator  # local_operator.hpp:78
ComplexILocalToLocalOperator = ILocalToLocalOperator                   # local_operator/virtual_local_to_local_operator.hpp:92 (main.cpp:99)
ComplexVirtualPartitioning = VirtualPartitioning  # noqa: F405         # main.cpp:89 (partitioning strategies only see coordinates)
ComplexDefaultApproximationBuilder = DefaultApproximationBuilder
ComplexDefaultLocalApproximationBuilder = DefaultLocalApproximationBuilder
ComplexDistributedOperator = DistributedOperator
ComplexRestrictedGlobalToLocalOperator = RestrictedGlobalToLocalOperator
ComplexVirtualLocalToLocalOperator = VirtualLocalToLocalOperator


def ComplexCustomApproximationBuilder(target_cluster, source_cluster, comm, operator):
    import numpy as _np

    return CustomApproximationBuilder(target_cluster, source_cluster, comm, operator, _np.complex128)


def _get_distributed_information(self, comm):
    """get_distributed_information(comm) (src/htool/hmatrix/hmatrix.hpp:53-54): the local statistics reduced over
    the ranks of `comm` (sums of counts and sizes, min/max of ranks and block sizes)."""
    import numpy as _np

    local = self.get_local_information()
    if comm is None or comm.Get_size() == 1:
        return local
    leaves = _np.asarray(self.leaves()).astype(_np.int64)
    dense = leaves[:, 4] < 0
    sz = leaves[:, 1] * leaves[:, 3]
    big = _np.iinfo(_np.int64).max
    sums = _np.array([dense.sum(), (~dense).sum(), sz[dense].sum(), (leaves[~dense, 4] * (leaves[~dense, 1] + leaves[~dense, 3])).sum(),
                      leaves[~dense, 4].sum(), self.shape[0] * self.shape[1], float(local.get("HBM_bytes", 0))], dtype=_np.float64)
    maxs = _np.array([sz[dense].max(initial=0), sz[~dense].max(initial=0), leaves[~dense, 4].max(initial=0)], dtype=_np.float64)
    mins = _np.array([sz[dense].min(initial=big), sz[~dense].min(initial=big), leaves[~dense, 4].min(initial=big)], dtype=_np.float64)
    import mpi4py as _mpi

    sums = comm.allreduce(sums, op=_mpi.MPI.SUM)
    maxs = comm.allreduce(maxs, op=_mpi.MPI.MAX)
    mins = comm.allreduce(mins, op=_mpi.MPI.MIN)
    mins = _np.where(mins >= big, 0, mins)
    stored = sums[2] + sums[3]
    return {
        "Number_of_dense_blocks": str(int(sums[0])), "Number_of_low_rank_blocks": str(int(sums[1])),
        "Dense_block_size_max": str(int(maxs[0])), "Dense_block_size_min": str(int(mins[0])),
        "Low_rank_block_size_max": str(int(maxs[1])), "Low_rank_block_size_min": str(int(mins[1])),
        "Rank_max": str(int(maxs[2])), "Rank_min": str(int(mins[2])), "Rank_mean": str(sums[4] / max(sums[1], 1)),
        "Compression_ratio": str(sums[5] / stored if stored else 0), "Space_saving": str(1 - stored / sums[5] if sums[5] else 0),
        "HBM_bytes": str(int(sums[6])),
    }


HMatrix.get_distributed_information = _get_distributed_information  # noqa: F405
ComplexHMatrix.get_distributed_information = _get_distributed_information  # noqa: F405
