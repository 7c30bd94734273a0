"""Hierarchical LU on the device: timings and statistics.   python tools/hlu_bench.py points [leaf] [eps] [block]
block = 1: the (partition 3 of 8) diagonal block of the operator on `points` points (BASELINE C5: 500000 -> 62 500 unknowns);
block = 0: the whole operator (symmetry 'S', one triangle stored, when the 4th argument is "S")."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import htool_python_amd  # noqa: F401,E402
import Htool  # noqa: E402
import torch  # noqa: E402
from htool_python_amd.workloads import points_in_sphere  # noqa: E402

import threading


def _heartbeat():  # (a long factorisation prints nothing by itself; the GPU pool takes seven silent minutes for a hang)
    t0 = time.time()
    while True:
        time.sleep(45)
        print("[hlu_bench] %.0f s" % (time.time() - t0), file=sys.stderr, flush=True)


threading.Thread(target=_heartbeat, daemon=True).start()
n = int(sys.argv[1])
leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 100
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
mode = sys.argv[4] if len(sys.argv) > 4 else "0"
Htool.set_device(0) if hasattr(Htool, "set_device") else None
pts = points_in_sphere(n, seed=0)
b = Htool.ClusterTreeBuilder()
b.set_maximal_leaf_size(leaf)
gen = Htool.NativeGenerator("inv_delta", pts, pts, 0.1)
t0 = time.time()
if mode in ("1", "1S"):   # "1S": the block declared symmetric -- factorised as A = L L^T on its lower triangle
    cl = b.create_cluster_tree(pts, 2, size_of_partition=8)
    H = Htool.HMatrixTreeBuilder(eps, 10.0, *(("S", "L") if mode == "1S" else ("N", "N"))).build_local(gen, cl, cl, 3, 3)
else:
    cl = b.create_cluster_tree(pts, 2, 2)
    H = Htool.HMatrixTreeBuilder(eps, 10.0, *(("S", "L") if mode == "S" else ("N", "N"))).build(gen, cl, cl)
torch.cuda.synchronize()
size = H.shape[0]
out = {"unknowns": size, "leaf": leaf, "eps": eps, "mode": mode, "build_s": round(time.time() - t0, 3), "hmatrix_GB": round(H.get_local_information().get("Device_bytes", 0) / 1e9, 3) if hasattr(H, "get_local_information") else None}
shift_rel = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0   # the bench's C5 system: (shift I + H) x = b, shift = 8e-3 |H|
shift = 0.0
if shift_rel > 0:
    v = np.random.RandomState(2).rand(size)
    for _ in range(6):
        w = H * v
        norm_h = np.linalg.norm(w) / np.linalg.norm(v)
        v = w / np.linalg.norm(w)
    shift = shift_rel * norm_h
out["shift"] = shift
x_ref = np.random.RandomState(1).rand(size)
bb = H * x_ref + shift * x_ref
for rep in range(int(os.environ.get("HLU_BENCH_REPS", "2"))):
    t0 = time.time()
    H.lu_factorization_shifted(shift) if shift > 0 else H.lu_factorization()
    torch.cuda.synchronize()
    out["lu_factorization_s_%d" % rep] = round(time.time() - t0, 3)
info = H.factorization_info()
out["info"] = {k: (round(v, 4) if isinstance(v, float) else int(v) if not isinstance(v, str) else v) for k, v in info.items()}
out["mean_rank_factors"] = round(info["rank_weight"] / max(info["rows_plus_columns"], 1), 2)
t0 = time.time()
x = H.lu_solve("N", bb)
out["lu_solve_host_s"] = round(time.time() - t0, 4)
out["solve_error"] = float(np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref))
os.environ["HTOOL_HLU_REFINE"] = "0"
x0 = H.lu_solve("N", bb)
out["solve_error_unrefined"] = float(np.linalg.norm(x0 - x_ref) / np.linalg.norm(x_ref))
# the raw application of the factors on a device vector (the preconditioner of a Krylov loop)
v = torch.rand(size, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mu in (1, 8):
    w = torch.rand(mu, size, dtype=torch.float64, device="cuda")
    H.factor_solve_device(1, "N", w.data_ptr(), size, mu, st)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        H.factor_solve_device(1, "N", w.data_ptr(), size, mu, st)
    torch.cuda.synchronize()
    out["apply_ms_mu%d" % mu] = round((time.time() - t0) / 5 * 1e3, 3)
t0 = time.time()
for _ in range(5):
    y = H * x_ref
out["product_host_ms"] = round((time.time() - t0) / 5 * 1e3, 3)
print(json.dumps(out))
