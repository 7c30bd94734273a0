"""Row-distributed operator + GMRES solve (counterparts of the reference's example/use_distributed_operator.py
and example/use_ddm_solver.py for this repository).

    python examples/distributed_gmres.py                        # one rank
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 examples/distributed_gmres.py
"""
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpi4py  # noqa: E402  (the torch.distributed-backed stand-in shipped with this repository)

import Htool  # noqa: E402
from htool_python_amd.workloads import points_in_sphere  # noqa: E402


def main():
    logging.basicConfig(level=logging.INFO)
    comm = mpi4py.MPI.COMM_WORLD
    size, eta, epsilon = 4000, 10, 1e-6
    points = points_in_sphere(size)

    cluster_builder = Htool.ClusterTreeBuilder()
    cluster_builder.set_maximal_leaf_size(32)
    cluster = cluster_builder.create_cluster_tree(points, 2, size_of_partition=comm.size)

    generator = Htool.NativeGenerator("inv_delta", points, points, 0.1)
    approximation = Htool.DefaultApproximationBuilder(generator, cluster, cluster, Htool.HMatrixTreeBuilder(epsilon, eta, "S", "L"), comm)
    operator = approximation.distributed_operator
    hmatrix = approximation.hmatrix
    Htool.recompression(hmatrix)

    np.random.seed(0)
    x_ref = np.random.rand(size)
    b = operator * x_ref
    d = points[:, :, None] - points[:, None, :]
    A = 1.0 / (0.1 + np.sqrt((d * d).sum(axis=0)))
    if comm.rank == 0:
        print("shape", operator.shape, "product error", np.linalg.norm(b - A @ x_ref) / np.linalg.norm(A @ x_ref))

    solver = Htool.DDMSolverBuilder(operator, approximation.block_diagonal_hmatrix).solver
    x = np.zeros(size)
    solver.set_hpddm_args("-hpddm_krylov_method gmres -hpddm_tol 1e-8 -hpddm_max_it 500 -hpddm_gmres_restart 100")
    solver.facto_one_level()
    solver.solve(x, b)
    distributed_information = hmatrix.get_distributed_information(comm)  # collective: every rank calls it
    if comm.rank == 0:
        print("solution error", np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref))
        print(solver.get_information())
        print(distributed_information)
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-6
    comm.Barrier()


if __name__ == "__main__":
    main()
