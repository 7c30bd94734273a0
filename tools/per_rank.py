"""What every rank of a P-GPU run has to do, measured one rank at a time on ONE GPU (same tree, same builds as
bench.py --gpus P): panels per rank, build time, product time (HIP events, cluster numbering in and out)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, Htool
from htool_python_amd.workloads import points_in_sphere, algorithmic_bytes

n = 1_000_000
pts = points_in_sphere(n, seed=0)
Htool.set_num_threads(16)
out = {}
for P in (2, 4, 8):
    cb = Htool.ClusterTreeBuilder(); cb.set_maximal_leaf_size(100)
    cl = cb.create_cluster_tree(pts, 2, size_of_partition=P)
    gen = Htool.NativeGenerator("laplace", pts, pts, 0.0)
    rows = []
    for p in range(P):
        t0 = time.time()
        H = Htool.HMatrixTreeBuilder(1e-3, 10.0, "N", "N").build(gen, cl, cl, p)
        torch.cuda.synchronize(); tb = time.time() - t0
        ab = algorithmic_bytes(H.leaves(), n, H.shape[0], 8)
        x = torch.rand(n, dtype=torch.float64).cuda(); y = torch.zeros(H.shape[0], dtype=torch.float64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(5): H.matvec_device(x.data_ptr(), y.data_ptr(), 1, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): H.matvec_device(x.data_ptr(), y.data_ptr(), 1, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        rows.append({"rank": p, "GB": ab["total"] / 1e9, "build_s": tb, "product_ms": ms})
        print(P, rows[-1], file=sys.stderr, flush=True)
        del H
    tot = sum(r["GB"] for r in rows); worst = max(r["product_ms"] for r in rows)
    out[f"P{P}"] = {"ranks": rows, "total_GB": tot, "slowest_rank_ms": worst, "aggregate_TBps_before_exchange": tot / worst}
print(json.dumps(out))
