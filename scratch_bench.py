import sys, time
sys.path.insert(0, '.')
import numpy as np, logging, torch
logging.basicConfig(level=logging.DEBUG)
import Htool
from oracle import oracle as O
n = int(sys.argv[1]); leaf = int(sys.argv[2]); eps = float(sys.argv[3])
np.random.seed(0)
pts = O.points_in_sphere(n)
t0 = time.time()
b = Htool.ClusterTreeBuilder(); b.set_maximal_leaf_size(leaf)
cl = b.create_cluster_tree(pts, 2)
t1 = time.time()
gen = Htool.NativeGenerator("laplace", pts, pts)
H = Htool.HMatrixTreeBuilder(eps, 10.0, "N", "N").build(gen, cl, cl)
t2 = time.time()
st = H.stats()
print("cluster s", t1 - t0, "build s", t2 - t1, st)
elems = st["dense_elements"] + st["low_rank_elements"]
B = 8 * (elems + 2 * n)
x = np.random.rand(n)
y = H * x
import torch
xd = torch.from_numpy(x).cuda(); yd = torch.zeros(n, dtype=torch.float64, device='cuda')
s = torch.cuda.current_stream().cuda_stream
for _ in range(3): H.matvec_device(xd.data_ptr(), yd.data_ptr(), 0, s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 20
e0.record()
for _ in range(K): H.matvec_device(xd.data_ptr(), yd.data_ptr(), 0, s)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
print(f"matvec {ms:.3f} ms  algorithmic {B/1e9:.2f} GB  -> {B/ms/1e6:.1f} GB/s ; hbm resident {st['hbm_bytes']/1e9:.2f} GB")
print("max diff host/device path", np.abs(yd.cpu().numpy() - y).max())
rows = np.arange(0, n, max(1, n // 200))
ye = O.dense_matvec(O.K_LAPLACE, pts, pts, x, rows=rows)
print("rel err sampled", np.linalg.norm(y[rows] - ye) / np.linalg.norm(ye))
