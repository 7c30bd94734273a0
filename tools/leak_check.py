"""Device-memory leak check: build / multiply / recompress / copy / destroy in a loop; free memory must come back."""
import copy, gc, sys
import numpy as np, torch, Htool
from htool_python_amd.workloads import points_in_sphere

def free_gb():
    torch.cuda.synchronize()
    f, t = torch.cuda.mem_get_info()
    return f / 1e9

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
pts = points_in_sphere(n, seed=0)
cb = Htool.ClusterTreeBuilder(); cb.set_maximal_leaf_size(100)
cl = cb.create_cluster_tree(pts, 2)
x = np.random.rand(n)
base = None
for it in range(4):
    for sym, uplo in (("N", "N"), ("S", "L")):
        H = Htool.HMatrixTreeBuilder(1e-3, 10.0, sym, uplo).build(Htool.NativeGenerator("laplace", pts, pts, 0.0), cl, cl)
        y = H * x
        Y = H @ np.asfortranarray(np.random.rand(n, 3))
        H2 = copy.deepcopy(H)
        Htool.recompression(H2)
        y2 = H2 * x
        del H, H2
        gc.collect()
    Htool.release_workspace()
    f = free_gb()
    if base is None:
        base = f
    print(f"cycle {it}: free {f:.3f} GB (first cycle {base:.3f})", flush=True)
assert abs(f - base) < 0.05, "device memory is leaking"
print("no leak")
