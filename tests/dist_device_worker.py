"""Worker of the multi-rank tests of the library's OWN exchange (htool_distributed_matvec_device / _matmat_device,
csrc/dist_device.hip) -- launched by torch.distributed.run with the gloo backend, all ranks sharing the box's one GPU.

The communicator object (the shipped mpi4py stand-in) has no RCCL handle here, so htool_comm.allgather_device is NULL and the
library stages its all-gather through pinned host memory and the communicator's host all-gather; everything else is the code
an 8-GPU run executes: per-rank counts / displacements from the source tree's partition, the zero-copy layout
(rank p's slice at p * pad) or padded slices + the compaction kernel, hipMemcpy2DAsync for several columns, the
cluster-numbered local product.  Mirrors the reference's assertions for the distributed product
(tests/test_distributed_operator.py:74-103: result against the exact operator, relative error < epsilon)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    import mpi4py
    import torch

    import Htool
    from oracle import oracle as O

    comm = mpi4py.MPI.COMM_WORLD
    rank, world = comm.Get_rank(), comm.Get_size()
    assert Htool.device_count() > 0
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    stream = torch.cuda.current_stream().cuda_stream
    g = np.load(os.path.join(ROOT, "tests", "golden", "distributed_400_d3.npz"))
    T, S = g["target"], g["source200"]
    b = Htool.ClusterTreeBuilder()
    b.set_maximal_leaf_size(10)
    cases = []
    # (a) the reference's own distributed case: 400 x 200 points, partition computed by the splitting rule
    cases.append(("regular", T, b.create_cluster_tree(T, 2, size_of_partition=world), S, b.create_cluster_tree(S, 2, size_of_partition=world), g["x200"], g["y200"]))
    # (b) a deliberately uneven partition given by the user (labels per point, cluster_tree_builder.hpp:32-39)
    lab_t = np.minimum((np.arange(400) * 7 // 400) * world // 7, world - 1).astype(np.int32)  # uneven pieces: sizes differ by tens
    lab_s = np.minimum((np.arange(200) * 5 // 200) * world // 5, world - 1).astype(np.int32)
    cases.append(("uneven", T, b.create_cluster_tree_from_global_partition(T, 2, world, lab_t), S,
                  b.create_cluster_tree_from_global_partition(S, 2, world, lab_s), g["x200"], g["y200"]))
    # (c) point counts that every world size of the tests divides: equal slices, the zero-copy layout
    T6, S6 = np.ascontiguousarray(T[:, :396]), np.ascontiguousarray(S[:, :198])
    cases.append(("even", T6, b.create_cluster_tree(T6, 2, size_of_partition=world), S6, b.create_cluster_tree(S6, 2, size_of_partition=world), None, None))
    checked = set()
    for name, tp, tcl, sp, scl, x_user, y_user in cases:
        s_sizes = [scl.get_cluster_on_partition(p).get_size() for p in range(world)]
        s_offs = [scl.get_cluster_on_partition(p).get_offset() for p in range(world)]
        t_loc = tcl.get_cluster_on_partition(rank)
        even = len(set(s_sizes)) == 1
        assert even == (name == "even" or (name == "regular" and world == 2)), (name, s_sizes)
        ns = sp.shape[1]
        tperm, sperm = np.asarray(tcl.get_permutation()), np.asarray(scl.get_permutation())
        for eps in (1e-3, 1e-6):
            for force_padded in ((False, True) if even else (False,)):
                os.environ["HTOOL_DIST_FORCE_PADDED"] = "1" if force_padded else "0"
                for cplx in (False, True):
                    if cplx:
                        gen = Htool.ComplexNativeGenerator("helmholtz", tp, sp, 3.0)
                        builder = Htool.ComplexHMatrixTreeBuilder(eps, 10.0, "N", "N")
                        dt, kind, par = torch.complex128, O.K_HELMHOLTZ, 3.0
                    else:
                        gen = Htool.NativeGenerator("inv_delta", tp, sp, 0.1)
                        builder = Htool.HMatrixTreeBuilder(eps, 10.0, "N", "N")
                        dt, kind, par = torch.float64, O.K_INV_DELTA, 0.1
                    holder = Htool.DefaultApproximationBuilder(gen, tcl, scl, builder, comm)
                    op, Hl = holder.distributed_operator, holder.hmatrix
                    assert not op.has_rccl
                    for mu in (1, 3):
                        kind_x = op.exchange_kind(mu)
                        assert kind_x == (3 if (even and not force_padded and mu == 1) else 4), (kind_x, even, force_padded, mu)
                        checked.add(kind_x)
                        rs = np.random.RandomState(7 + mu)
                        Xu = rs.rand(mu, ns) + (1j * rs.rand(mu, ns) if cplx else 0)  # row c = right-hand side c, USER numbering
                        if not cplx and mu == 1 and x_user is not None:
                            Xu[0] = x_user
                        Xc = Xu[:, sperm]                                                # cluster numbering
                        mine = s_sizes[rank]
                        ldx = mine + 5                                                   # a leading dimension larger than the slice
                        x_loc = torch.full((mu, ldx), float("nan"), dtype=dt, device="cuda")
                        x_loc[:, :mine] = torch.from_numpy(np.ascontiguousarray(Xc[:, s_offs[rank]: s_offs[rank] + mine])).cuda()
                        ldy = t_loc.get_size() + 3
                        y_loc = torch.zeros(mu, ldy, dtype=dt, device="cuda")
                        if mu == 1:
                            op.matvec_device(x_loc.data_ptr(), y_loc.data_ptr(), stream)
                        else:
                            op.matmat_device(x_loc.data_ptr(), ldx, y_loc.data_ptr(), ldy, mu, stream)
                        torch.cuda.synchronize()
                        got = y_loc[:, : t_loc.get_size()].cpu().numpy()
                        assert not np.isnan(got).any()
                        # (1) against the exact operator on this rank's rows: the reference's bar, relative error < epsilon
                        rows = tperm[t_loc.get_offset(): t_loc.get_offset() + t_loc.get_size()]
                        for c in range(mu):
                            ye = O.dense_matvec(kind, tp, sp, Xu[c], par, rows=rows)
                            err2 = comm.allreduce(np.array([np.linalg.norm(got[c] - ye) ** 2, np.linalg.norm(ye) ** 2]), op=mpi4py.MPI.SUM)
                            assert np.sqrt(err2[0] / err2[1]) < eps, (name, eps, force_padded, cplx, mu, c, np.sqrt(err2[0] / err2[1]))
                        if not cplx and mu == 1 and y_user is not None:  # the golden product of the reference's case, rows of all ranks together
                            e2 = comm.allreduce(np.array([np.linalg.norm(got[0] - y_user[rows]) ** 2, np.linalg.norm(y_user[rows]) ** 2]), op=mpi4py.MPI.SUM)
                            assert np.sqrt(e2[0] / e2[1]) < eps
                        # (2) bitwise equal to the one-rank product of the same rows: this rank's H-matrix applied to the whole
                        # cluster-numbered vector, no exchange involved
                        x_full = torch.from_numpy(np.ascontiguousarray(Xc)).cuda()
                        y_ref = torch.zeros(mu, t_loc.get_size(), dtype=dt, device="cuda")
                        Hl.matmat_device(x_full.data_ptr(), ns, y_ref.data_ptr(), t_loc.get_size(), mu, 1, stream)
                        torch.cuda.synchronize()
                        assert torch.equal(y_ref, y_loc[:, : t_loc.get_size()]), (name, eps, force_padded, cplx, mu)
                    # (2b) the TRANSPOSED product with the same distribution: x given by the rows each rank owns, y received by source
                    # slices; one reduce-scatter inside the library.  Against exact entries: the kernels are functions of the distance,
                    # (A^T w)[j] = sum_i K(s_j, t_i) w_i; 'C' conjugates A.
                    t_sizes = [tcl.get_cluster_on_partition(q).get_size() for q in range(world)]
                    t_offs = [tcl.get_cluster_on_partition(q).get_offset() for q in range(world)]
                    nt = tp.shape[1]
                    for mu, tr in ((1, "T"), (3, "C" if cplx else "T")):
                        rs = np.random.RandomState(11 + mu)
                        Wu = rs.rand(mu, nt) + (1j * rs.rand(mu, nt) if cplx else 0)   # USER numbering of the target points
                        Wc = Wu[:, tperm]
                        mine_t, mine_s = t_sizes[rank], s_sizes[rank]
                        ldx, ldy = mine_t + 2, mine_s + 4
                        w_loc = torch.full((mu, ldx), float("nan"), dtype=dt, device="cuda")
                        w_loc[:, :mine_t] = torch.from_numpy(np.ascontiguousarray(Wc[:, t_offs[rank]: t_offs[rank] + mine_t])).cuda()
                        z_loc = torch.zeros(mu, ldy, dtype=dt, device="cuda")
                        op.matmat_device_trans(tr, w_loc.data_ptr(), ldx, z_loc.data_ptr(), ldy, mu, stream)
                        torch.cuda.synchronize()
                        gotz = z_loc[:, :mine_s].cpu().numpy()
                        assert not np.isnan(gotz).any()
                        cols = sperm[s_offs[rank]: s_offs[rank] + mine_s]
                        for c in range(mu):
                            win = np.conj(Wu[c]) if tr == "C" else Wu[c]
                            ze = O.dense_matvec(kind, sp, tp, win, par, rows=cols)
                            if tr == "C":
                                ze = np.conj(ze)
                            e2 = comm.allreduce(np.array([np.linalg.norm(gotz[c] - ze) ** 2, np.linalg.norm(ze) ** 2]), op=mpi4py.MPI.SUM)
                            assert np.sqrt(e2[0] / e2[1]) < eps, (name, eps, force_padded, cplx, mu, tr, c, np.sqrt(e2[0] / e2[1]))
                    # (3) the replicated-vector API of the reference on the same operator agrees
                    if not cplx and x_user is not None:
                        y_rep = op * x_user
                        assert np.linalg.norm(y_rep - y_user) / np.linalg.norm(y_user) < eps
                    del holder, op, Hl
    os.environ["HTOOL_DIST_FORCE_PADDED"] = "0"
    assert checked == {3, 4}, checked

    # GMRES through the reference's solver surface: the Krylov loop's operator is the library call above
    gen = Htool.NativeGenerator("inv_delta", T, T, 0.1)
    tcl = b.create_cluster_tree(T, 2, size_of_partition=world)
    holder = Htool.DefaultApproximationBuilder(gen, tcl, tcl, Htool.HMatrixTreeBuilder(1e-8, 10.0, "N", "N"), comm)
    solver = Htool.DDMSolverBuilder(holder.distributed_operator, holder.block_diagonal_hmatrix).solver
    assert solver.op.dist_op is not None and solver.op.dist_op.exchange_kind(1) in (3, 4)
    x_ref = np.random.RandomState(3).rand(400)
    bb = holder.distributed_operator * x_ref
    x = np.zeros(400)
    solver.solve(x, bb, "-hpddm_krylov_method gmres -hpddm_tol 1e-9 -hpddm_max_it 400 -hpddm_gmres_restart 200")
    A = O.kernel_block(O.K_INV_DELTA, T, T, 0.1)
    assert np.linalg.norm(A @ x - bb) / np.linalg.norm(bb) < 1e-6
    comm.Barrier()
    print(f"rank {rank}/{world} ok (device-exchange)", flush=True)


if __name__ == "__main__":
    main()
