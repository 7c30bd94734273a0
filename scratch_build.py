import sys, time
sys.path.insert(0, '.')
import numpy as np, logging, torch
logging.basicConfig(level=logging.DEBUG)
import Htool
from htool_python_amd.workloads import points_in_sphere
n = int(sys.argv[1]); leaf = int(sys.argv[2]); eps = float(sys.argv[3])
pts = points_in_sphere(n)
b = Htool.ClusterTreeBuilder(); b.set_maximal_leaf_size(leaf)
t0 = time.time(); cl = b.create_cluster_tree(pts, 2); print("cluster", time.time() - t0)
gen = Htool.NativeGenerator("laplace", pts, pts)
for rep in range(2):
    t0 = time.time()
    H = Htool.HMatrixTreeBuilder(eps, 10.0, "N", "N").build(gen, cl, cl)
    print("build wall", time.time() - t0, H.stats()["build_seconds"])
    del H
