import logging, os, sys
logging.basicConfig(level=logging.DEBUG, stream=sys.stderr)
import numpy as np, torch, Htool
from htool_python_amd.workloads import points_in_sphere
pts = points_in_sphere(12000, seed=0)
cb = Htool.ClusterTreeBuilder(); cb.set_maximal_leaf_size(40)
cl = cb.create_cluster_tree(pts, 2)
for mb in (2, 12, 40):
    os.environ["HTOOL_BUILD_ARENA_MB"] = str(mb)
    print("arena MB", mb, file=sys.stderr)
    H = Htool.HMatrixTreeBuilder(1e-5, 10.0, "N", "N").build(Htool.NativeGenerator("laplace", pts, pts, 0.0), cl, cl)
