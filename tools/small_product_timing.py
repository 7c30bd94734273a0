"""Wall time per product for small operators (launch-bound regime)."""
import sys, time
import numpy as np, torch, Htool
from htool_python_amd.workloads import points_in_sphere
for n in (10000, 30000, 100000):
    pts = points_in_sphere(n, seed=0)
    cb = Htool.ClusterTreeBuilder(); cb.set_maximal_leaf_size(50)
    cl = cb.create_cluster_tree(pts, 2)
    for sym in ("N", "S"):
        H = Htool.HMatrixTreeBuilder(1e-3, 10.0, sym, "L" if sym == "S" else "N").build(Htool.NativeGenerator("inv_delta", pts, pts, 0.1), cl, cl)
        x = torch.rand(n, dtype=torch.float64).cuda(); y = torch.zeros(n, dtype=torch.float64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(50): H.matvec_device(x.data_ptr(), y.data_ptr(), 0, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500): H.matvec_device(x.data_ptr(), y.data_ptr(), 0, st)
        torch.cuda.synchronize()
        print(n, sym, f"{(time.perf_counter() - t0) / 500 * 1e6:.1f} us per product", flush=True)
