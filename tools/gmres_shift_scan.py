"""Scan of the diagonal shift of the GMRES workload (BASELINE config 5): iterations of unpreconditioned GMRES(50..100) to a
relative residual of 1e-6 on (shift I + H), H the n-point Laplace H-matrix, for shift = fraction * |H|.
    python tools/gmres_shift_scan.py [n] > gpurun_out/gmres_shift_scan.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import Htool  # noqa: E402
from htool_python_amd.krylov import gmres  # noqa: E402
from htool_python_amd.solver import DeviceOperator  # noqa: E402
from htool_python_amd.workloads import gmres_shift, points_in_sphere  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
pts = points_in_sphere(n, seed=0)
b = Htool.ClusterTreeBuilder()
b.set_maximal_leaf_size(100)
cl = b.create_cluster_tree(pts, 2)
H = Htool.HMatrixTreeBuilder(1e-3, 10.0, "N", "N").build(Htool.NativeGenerator("laplace", pts, pts), cl, cl)
plain = DeviceOperator(H, None, 0, None, 0.0)
_, lam = gmres_shift(plain.apply, n, torch.float64, fraction=1.0)
out = {"n": n, "norm_estimate": lam, "scan": []}
g = torch.Generator(device="cpu").manual_seed(7)
x_ref = torch.rand(n, dtype=torch.float64, generator=g).cuda()
for frac in (3e-2, 1e-2, 5e-3, 3e-3, 2e-3, 1e-3, 5e-4):
    op = DeviceOperator(H, None, 0, None, frac * lam)
    bvec = op.apply(x_ref)
    xs, info = gmres(op.apply, bvec, tol=1e-6, restart=100, max_it=100)
    res = info["residuals"]
    out["scan"].append({"fraction": frac, "shift": frac * lam, "iterations_to_1e-6": info["iterations"] if res[-1] <= 1e-6 else None,
                        "residual_at_15": res[min(14, len(res) - 1)], "residual_at_50": res[min(49, len(res) - 1)], "last": res[-1],
                        "solution_error": float(torch.linalg.norm(xs - x_ref) / torch.linalg.norm(x_ref))})
print(json.dumps(out, indent=1))
