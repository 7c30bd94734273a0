"""Minimal mpi4py stand-in backed by torch.distributed (gloo for host buffers, RCCL for device tensors).

The reference passes `mpi4py.MPI.COMM_WORLD` to DefaultApproximationBuilder
(example/use_distributed_operator.py:18,53; caster src/htool/misc/wrapper_mpi.hpp:28-55).  Neither
mpi4py nor an MPI for this interpreter exists in the target image, so this shim provides the small
part of the API those scripts and tests use: COMM_WORLD.size/.rank/Get_size()/Get_rank()/
allreduce(op=SUM)/Barrier()/bcast(), launched with torchrun (RANK/WORLD_SIZE/MASTER_* env).
"""
from . import MPI  # noqa: F401
