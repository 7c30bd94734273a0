"""Worker of the multi-process tests (launched by torch.distributed.run, one process per rank).

mode "cpu": communicator shim + partition logic on the gloo backend (no GPU, no compute calls).
mode "gpu": the reference's distributed-operator assertions (tests/test_distributed_operator.py:74-103)
            with every rank building and multiplying its row block on the GPU (ranks may share one GPU).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main(mode):
    import mpi4py

    import Htool
    from oracle import oracle as O

    comm = mpi4py.MPI.COMM_WORLD
    rank, world = comm.Get_rank(), comm.Get_size()
    assert world == int(os.environ["WORLD_SIZE"]) and rank == int(os.environ["RANK"])
    assert comm.size == world and comm.rank == rank

    # ---- communicator shim
    assert comm.allreduce(rank + 1, op=mpi4py.MPI.SUM) == world * (world + 1) // 2
    counts = [3 + p for p in range(world)]
    displs = [sum(counts[:p]) for p in range(world)]
    send = np.full(counts[rank], rank + 1, dtype=np.uint8)
    recv = np.zeros(sum(counts), dtype=np.uint8)
    comm._htool_allgatherv(send, recv, counts, displs)
    assert recv.tolist() == [p + 1 for p in range(world) for _ in range(counts[p])]
    assert comm.bcast({"a": rank} if rank == 0 else None, root=0) == {"a": 0}

    # ---- partition logic (tests/test_cluster.py:33-34 with np = world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "distributed_400_d3.npz"))
    T, S = g["target"], g["source200"]
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(10)
    tcl = b.create_cluster_tree(T, 2, size_of_partition=world)
    scl = b.create_cluster_tree(S, 2, size_of_partition=world)
    local = tcl.get_cluster_on_partition(rank)
    total = sum(tcl.get_cluster_on_partition(p).get_size() for p in range(world))
    assert total == len(local.get_permutation()) == len(tcl.get_permutation()) == 400
    assert comm.allreduce(local.get_size(), op=mpi4py.MPI.SUM) == 400
    # every rank's work queues tile exactly its own rows x all columns
    adm, dns = Htool.block_tree_queues(tcl, scl, 10.0, target_partition_number=rank)
    cover = np.zeros((local.get_size(), 200), dtype=np.int32)
    for t_off, m, s_off, n in list(np.asarray(adm)) + list(np.asarray(dns)):
        cover[t_off - local.get_offset():t_off - local.get_offset() + m, s_off:s_off + n] += 1
    assert cover.min() == 1 and cover.max() == 1

    if mode == "gpu":
        from tests.helpers import NumpyGenerator

        assert Htool.device_count() > 0
        for eps in (1e-3, 1e-6):
            for native in (False, True):
                gen = Htool.NativeGenerator("inv_delta", T, S, 0.1) if native else NumpyGenerator(T, S)
                holder = Htool.DefaultApproximationBuilder(gen, tcl, scl, Htool.HMatrixTreeBuilder(eps, 10.0, "N", "N"), comm)
                op, local_h = holder.distributed_operator, holder.hmatrix
                assert op.shape == (comm.allreduce(local_h.shape[0], op=mpi4py.MPI.SUM), local_h.shape[1]) == (400, 200)
                dinfo = local_h.get_distributed_information(comm)
                nblocks = comm.allreduce(len(local_h.leaves()), op=mpi4py.MPI.SUM)
                assert int(dinfo["Number_of_dense_blocks"]) + int(dinfo["Number_of_low_rank_blocks"]) == nblocks
                y = op * g["x200"]
                assert np.linalg.norm(y - g["y200"]) / np.linalg.norm(g["y200"]) < eps
                for mu in (5, 1):
                    np.random.seed(1)
                    X = np.asfortranarray(np.random.rand(200, mu))
                    Y = op @ X
                    Ye = O.dense_matvec(O.K_INV_DELTA, T, S, X, 0.1)
                    assert Y.shape == (400, mu) and np.linalg.norm(Y - Ye) / np.linalg.norm(Ye) < eps
                # every rank returns the same replicated result
                ysum = comm.allreduce(y, op=mpi4py.MPI.SUM)
                assert np.allclose(ysum, world * y, rtol=1e-13, atol=0)
        # user-extensible operators (tests/test_distributed_operator.py rows "ExtraDiagonal" and
        # "LocalAndExtraDiagonal", fixtures tests/conftest.py:223-293,352-376)
        from tests.helpers import CustomLocalToLocalOperator, CustomRestrictedGlobalToLocalOperator

        gen = NumpyGenerator(T, S)
        t_loc, s_loc = tcl.get_cluster_on_partition(rank), scl.get_cluster_on_partition(rank)
        extra = []
        if s_loc.get_offset() > 0:
            extra.append(CustomRestrictedGlobalToLocalOperator(gen, Htool.LocalRenumbering(t_loc), Htool.LocalRenumbering(0, s_loc.get_offset(), scl.get_permutation()), False, False))
        rest = scl.get_size() - s_loc.get_size() - s_loc.get_offset()
        if rest > 0:
            extra.append(CustomRestrictedGlobalToLocalOperator(gen, Htool.LocalRenumbering(t_loc), Htool.LocalRenumbering(s_loc.get_size() + s_loc.get_offset(), rest, scl.get_permutation()), False, False))
            assert extra[-1].local_target_renumbering.size == t_loc.get_size() and extra[-1].local_source_renumbering.size == rest
        for flavour in ("ExtraDiagonal", "LocalAndExtraDiagonal"):
            if flavour == "ExtraDiagonal":
                holder = Htool.DefaultLocalApproximationBuilder(gen, tcl, scl, Htool.HMatrixTreeBuilder(1e-6, 10.0, "N", "N"), comm)
            else:
                holder = Htool.CustomApproximationBuilder(tcl, scl, comm, CustomLocalToLocalOperator(gen, Htool.LocalRenumbering(t_loc), Htool.LocalRenumbering(s_loc)))
            for op_extra in extra:
                holder.distributed_operator.add_global_to_local_operator(op_extra)
            opx = holder.distributed_operator
            assert opx.shape == (400, 200)
            y = opx * g["x200"]
            assert np.linalg.norm(y - g["y200"]) / np.linalg.norm(g["y200"]) < 1e-6
            np.random.seed(2)
            X = np.asfortranarray(np.random.rand(200, 5))
            Ye = O.dense_matvec(O.K_INV_DELTA, T, S, X, 0.1)
            assert np.linalg.norm(opx @ X - Ye) / np.linalg.norm(Ye) < 1e-6
            # sub-product on a slice of the cluster-numbered input (tests/test_distributed_operator.py:105-129)
            x = g["x200"].copy()
            o_, s_ = 20, 20
            x[:o_] = 0
            x[o_ + s_:] = 0
            x_perm = np.zeros(200)
            x_perm[np.asarray(scl.get_permutation())] = x
            y1 = opx.internal_sub_vector_product_global_to_local(x[o_:o_ + s_], o_)
            y2 = O.dense_matvec(O.K_INV_DELTA, T, S, x_perm, 0.1)[np.asarray(tcl.get_permutation())]
            assert np.linalg.norm(y1 - y2[t_loc.get_offset():t_loc.get_offset() + t_loc.get_size()]) / np.linalg.norm(y2) < 11e-6
        # same sub-product through the default operator
        holder = Htool.DefaultApproximationBuilder(gen, tcl, scl, Htool.HMatrixTreeBuilder(1e-6, 10.0, "N", "N"), comm)
        y1 = holder.distributed_operator.internal_sub_vector_product_global_to_local(x[o_:o_ + s_], o_)
        assert np.linalg.norm(y1 - y2[t_loc.get_offset():t_loc.get_offset() + t_loc.get_size()]) / np.linalg.norm(y2) < 11e-6
        # block-diagonal part of the default approximation (distributed_operator/utility.hpp:31)
        gen = NumpyGenerator(T, T)
        holder = Htool.DefaultApproximationBuilder(gen, tcl, tcl, Htool.HMatrixTreeBuilder(1e-6, 10.0, "N", "N"), comm)
        bd = holder.block_diagonal_hmatrix
        assert bd.shape == (local.get_size(), local.get_size())
        permt = np.asarray(tcl.get_permutation())
        idx = permt[local.get_offset():local.get_offset() + local.get_size()]
        Aloc = O.kernel_block(O.K_INV_DELTA, T[:, idx], T[:, idx], 0.1)
        xs = np.random.RandomState(5).rand(local.get_size())
        xin = xs
        if world == 1:  # one rank: the block-diagonal part IS the operator (user numbering on both sides)
            xin = np.zeros(400); xin[idx] = xs
            assert np.linalg.norm((bd * xin)[idx] - Aloc @ xs) <= 1e-6 * np.linalg.norm(Aloc @ xs)
        else:
            assert np.linalg.norm(bd * xin - Aloc @ xs) <= 1e-6 * np.linalg.norm(Aloc @ xs)
        # Krylov solve through the reference's solver surface (example/use_ddm_solver.py:49-69; our own GMRES,
        # no preconditioner): square symmetric operator on the target cloud, rows split over the ranks
        gen = Htool.NativeGenerator("inv_delta", T, T, 0.1)
        holder = Htool.DefaultApproximationBuilder(gen, tcl, tcl, Htool.HMatrixTreeBuilder(1e-8, 10.0, "S", "L"), comm)
        solver = Htool.DDMSolverBuilder(holder.distributed_operator, holder.block_diagonal_hmatrix).solver
        np.random.seed(3)
        x_ref = np.random.random(400)
        bb = holder.distributed_operator * x_ref
        x = np.zeros(400)
        solver.set_hpddm_args("-hpddm_compute_residual l2 ")
        solver.facto_one_level()
        solver.solve(x, bb, "-hpddm_krylov_method gmres -hpddm_variant right -hpddm_tol 1e-9 -hpddm_max_it 400 -hpddm_gmres_restart 200")
        A = O.kernel_block(O.K_INV_DELTA, T, T, 0.1)
        info = solver.get_information()
        assert float(info["Relative_residual"]) <= 1e-9, info
        assert np.linalg.norm(A @ x - bb) / np.linalg.norm(bb) < 1e-6
        assert np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref) < 1e-3
        X2, B2 = np.zeros((400, 2), order="F"), np.asfortranarray(np.stack([bb, 2 * bb], axis=1))
        solver.solve(X2, B2)
        assert np.linalg.norm(X2[:, 1] - 2 * X2[:, 0]) / np.linalg.norm(X2[:, 1]) < 1e-6
        # partitioned geometry of example/use_distributed_operator.py (local partition given), world == 2 only
        if world == 2:
            gp = np.load(os.path.join(ROOT, "tests", "golden", "partitioned_1000_w2.npz"))
            tp = b.create_cluster_tree_from_local_partition(gp["target"], 2, 2, gp["partition"])
            sp = b.create_cluster_tree(gp["source"], 2)
            gen = NumpyGenerator(gp["target"], gp["source"])
            holder = Htool.DefaultApproximationBuilder(gen, tp, sp, Htool.HMatrixTreeBuilder(1e-3, 10.0, "N", "N"), comm)
            Htool.openmp_recompression(holder.hmatrix)
            y = holder.distributed_operator * gp["x"]
            assert np.linalg.norm(y - gp["y"]) / np.linalg.norm(gp["y"]) < 1e-3
    comm.Barrier()
    print(f"rank {rank}/{world} ok ({mode})", flush=True)


if __name__ == "__main__":
    main(sys.argv[1])
