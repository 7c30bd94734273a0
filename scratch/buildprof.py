import logging, time, sys
logging.basicConfig(level=logging.DEBUG, stream=sys.stderr)
import numpy as np, torch, Htool
from htool_python_amd.workloads import points_in_sphere
n=1_000_000
pts=points_in_sphere(n, seed=0)
Htool.set_num_threads(16)
for rep in range(3):
    t0=time.time()
    cb=Htool.ClusterTreeBuilder(); cb.set_maximal_leaf_size(100)
    cl=cb.create_cluster_tree(pts,2,size_of_partition=1)
    t1=time.time()
    gen=Htool.NativeGenerator("laplace",pts,pts,0.0)
    b=Htool.HMatrixTreeBuilder(1e-3,10.0,"N","N")
    torch.cuda.synchronize(); t2=time.time()
    H=b.build(gen,cl,cl)
    torch.cuda.synchronize(); t3=time.time()
    print(f"rep {rep}: cluster {t1-t0:.3f} gen {t2-t1:.3f} build {t3-t2:.3f}", file=sys.stderr)
    del H
    time.sleep(3)
