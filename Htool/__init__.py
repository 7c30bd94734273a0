"""`import Htool` -- the reference's module name (src/htool/main.cpp:40), served by the MI355X engine.

The compiled pybind11 shim lives in htool_python_amd/ (built in-tree by htool_python_amd.build);
this package only re-exports it and adds the pure-Python plotting helpers.  There is no CPU
fallback: if the extension is missing the import fails.
"""
from htool_python_amd.Htool import *  # noqa: F401,F403
from htool_python_amd.Htool import __doc__ as _core_doc  # noqa: F401
from htool_python_amd.plotting import plot  # noqa: F401
from htool_python_amd.solver import DDMSolverBuilder, Solver  # noqa: F401,E402

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

ComplexDefaultApproximationBuilder = DefaultApproximationBuilder
ComplexDefaultLocalApproximationBuilder = DefaultLocalApproximationBuilder
ComplexDistributedOperator = DistributedOperator
ComplexRestrictedGlobalToLocalOperator = RestrictedGlobalToLocalOperator
ComplexVirtualLocalToLocalOperator = VirtualLocalToLocalOperator


def ComplexCustomApproximationBuilder(target_cluster, source_cluster, comm, operator):
    import numpy as _np

    return CustomApproximationBuilder(target_cluster, source_cluster, comm, operator, _np.complex128)
