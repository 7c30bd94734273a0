"""The 62 500-unknown diagonal block of BASELINE config C5 on 8 GPUs through the dense device factorisation, step by step with the
library's log on stderr (standalone form of tests/test_gpu_boundary.py::test_one_level_preconditioner_at_the_per_gpu_block_of_c5)."""
import logging, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.basicConfig(level=logging.DEBUG, stream=sys.stderr)
import numpy as np, torch, Htool
from htool_python_amd.workloads import points_in_sphere
n, world, p = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000, 8, 3
pts = points_in_sphere(n, seed=0)
b = Htool.ClusterTreeBuilder(); b.set_maximal_leaf_size(100)
cl = b.create_cluster_tree(pts, 2, size_of_partition=world)
gen = Htool.NativeGenerator("inv_delta", pts, pts, 0.1)
sym = sys.argv[2] if len(sys.argv) > 2 else "N"   # "S": the block of a symmetric operator (lu_factorization then tries Cholesky first)
Hb = Htool.HMatrixTreeBuilder(1e-6, 10.0, sym, "L" if sym == "S" else "N").build_local(gen, cl, cl, p, p)
size = Hb.shape[0]
print("block", Hb.shape, flush=True)
x_ref = np.random.RandomState(1).rand(size)
bb = Hb * x_ref
t = time.time(); Hb.lu_factorization(); torch.cuda.synchronize(); print("lu_factorization", time.time() - t, flush=True)
t = time.time(); x = Hb.lu_solve("N", bb); print("lu_solve", time.time() - t, "error", np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref), flush=True)
