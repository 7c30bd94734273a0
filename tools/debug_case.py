"""Diagnose a fuzz case: which rows of y are wrong, and how they relate to the tiles / leaves."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import Htool
from tools.fuzz import exact_rows

def cloud(rng, shape, dim, m):
    if shape == "ball":
        p = rng.randn(dim, m); p /= np.linalg.norm(p, axis=0); return p * rng.rand(m) ** (1.0 / dim)
    if shape == "cube":
        return rng.rand(dim, m)
    raise SystemExit("shape")

# parameters of seed 44 / case 538, with a fresh cloud (the bug should not depend on the exact points)
n, ns, leaf, eta, eps, children, strategy = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
rng = np.random.RandomState(5)
pt = np.asfortranarray(cloud(rng, "ball", 3, n)); ps = np.asfortranarray(cloud(rng, "ball", 3, ns) + 0.3)
cb = Htool.ClusterTreeBuilder(); cb.set_maximal_leaf_size(leaf); cb.set_partitioning_strategy(getattr(Htool, strategy)())
ct = cb.create_cluster_tree(pt, children, size_of_partition=1); cs = cb.create_cluster_tree(ps, children, size_of_partition=1)
H = Htool.HMatrixTreeBuilder(eps, eta, "N", "N").build(Htool.NativeGenerator("laplace", pt, ps, 0.0), ct, cs)
st = H.stats(); print(st)
x = rng.rand(ns)
y = H * x
rows = rng.choice(n, 2000, replace=False)
ye = exact_rows("laplace", pt, ps, x, 0.0, rows)
rel = np.abs(y[rows] - ye) / np.abs(ye)
print("overall err", np.linalg.norm(y[rows] - ye) / np.linalg.norm(ye), "bad rows", int((rel > 100 * eps).sum()), "of", len(rows))
perm = np.asarray(ct.get_permutation()); inv = np.empty(n, dtype=np.int64); inv[perm] = np.arange(n)
bad = rows[rel > 100 * eps]
if len(bad):
    pos = np.sort(inv[bad]); print("cluster positions of bad rows: min", pos.min(), "max", pos.max(), "first", pos[:20])
    L = np.asarray(H.leaves())
    tiles = np.asarray(Htool.cluster_tiles(ct))
    # tile of each bad position
    tstart = tiles[:, 0]; idx = np.searchsorted(tstart, pos, side="right") - 1
    print("tiles of bad rows (offset,size):", [tuple(tiles[i]) for i in np.unique(idx)[:15]])
    # columns (dense + lr) per bad tile
    for i in np.unique(idx)[:5]:
        o, sz = tiles[i]
        sel = L[(L[:, 0] <= o) & (L[:, 0] + L[:, 1] >= o + sz)]
        print(" tile", o, sz, "leaves", len(sel), "dense cols", int(sel[sel[:, 4] < 0][:, 3].sum()), "lr cols", int(sel[sel[:, 4] > 0][:, 4].sum()))
