"""Diagnostic: hierarchical LU of partition-built diagonal blocks against the dense solve (one process)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import htool_python_amd  # noqa: F401,E402
import Htool  # noqa: E402
import torch  # noqa: E402
from htool_python_amd.workloads import points_in_sphere  # noqa: E402

n, leaf, eps = 4000, 32, 1e-6
pts = points_in_sphere(n)
b = Htool.ClusterTreeBuilder()
b.set_maximal_leaf_size(leaf)
cl = b.create_cluster_tree(pts, 2, size_of_partition=2)
gen = Htool.NativeGenerator("inv_delta", pts, pts, 0.1)
for p in (0, 1):
    Hb = Htool.HMatrixTreeBuilder(eps, 10.0, "S", "L").build_local(gen, cl, cl, p, p)
    size = Hb.shape[0]
    A = np.asarray(Hb.to_dense())
    Hb.lu_factorization()
    print("partition", p, Hb.factorization_info()["kind"], "truncations at capacity", Hb.factorization_info().get("truncations_at_capacity"))
    B = np.random.default_rng(1).normal(size=(size, 2))
    Xd = np.linalg.solve(A, B)
    os.environ["HTOOL_HLU_REFINE"] = "0"
    X = Hb.lu_solve("N", np.asfortranarray(B))
    X2 = Hb.lu_solve("N", np.asfortranarray(B))
    print("  host solve unrefined rel err", np.linalg.norm(X - Xd) / np.linalg.norm(Xd), "repeatable", np.array_equal(X, X2))
    Bt = torch.from_numpy(np.ascontiguousarray(B.T)).cuda()
    Hb.factor_solve_device(1, "N", Bt.data_ptr(), size, 2, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("  device solve rel err", np.linalg.norm(Bt.cpu().numpy().T - Xd) / np.linalg.norm(Xd))
    v = torch.from_numpy(np.ascontiguousarray(B[:, 0])).cuda()
    outs = []
    for _ in range(3):
        w = v.clone()
        Hb.factor_solve_device(1, "N", w.data_ptr(), size, 1, torch.cuda.current_stream().cuda_stream)
        outs.append(w.cpu().numpy())
    print("  one column rel err", np.linalg.norm(outs[0] - Xd[:, 0]) / np.linalg.norm(Xd[:, 0]), "repeatable", np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2]))
