"""Task / level / memory statistics of the hierarchical-LU plan (host only: runs without a GPU).

    python tools/hlu_plan_stats.py [points] [leaf] [rank] [eta]
"""
import ctypes
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import htool_python_amd  # noqa: F401,E402  (loads the library)
import Htool  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 8
eta = float(sys.argv[4]) if len(sys.argv) > 4 else 10.0

rng = np.random.default_rng(0)
pts = rng.normal(size=(3, n))
pts /= np.linalg.norm(pts, axis=0)
pts *= rng.random(n) ** (1 / 3)
pts = np.asfortranarray(pts)
b = Htool.ClusterTreeBuilder()
b.set_maximal_leaf_size(leaf)
t0 = time.time()
cl = b.create_cluster_tree(pts, 2, 2)
adm, dns = Htool.block_tree_queues(cl, cl, eta, 0, 0, -1)
print("cluster + block tree %.2f s: %d admissible, %d dense" % (time.time() - t0, len(adm), len(dns)))
rects = np.zeros((len(adm) + len(dns), 5), dtype=np.int32)
rects[: len(adm), :4] = adm
rects[: len(adm), 4] = rank
rects[len(adm):, :4] = dns
rects[len(adm):, 4] = -1
t0 = time.time()
plan = Htool.HLUPlan(cl, rects, 1e-3)
v = plan.info()
names = ["n", "leaves", "diag leaves", "factor elems", "diag elems", "scratch elems", "rank slots", "windows", "factor tasks", "factor launches", "factor levels",
         "solve tasks", "solve launches", "solve levels", "plan us", "solveT tasks", "FILL", "APPLY_DENSE", "APPLY_LR", "ADDLR", "FINAL", "DDPROD", "GETRF"]
print("plan %.2f s" % (time.time() - t0))
for k, x in zip(names, v):
    print("%-16s %d" % (k, x))
hm_elems = sum(int(r[1]) * int(r[3]) if r[4] < 0 else rank * (int(r[1]) + int(r[3])) for r in rects)
print("H-matrix %.3f GB, factor arena %.3f GB (%.2f x), diag %.3f GB, scratch %.3f GB" % (hm_elems * 8e-9, v[3] * 8e-9, v[3] / hm_elems, v[4] * 8e-9, v[5] * 8e-9))
