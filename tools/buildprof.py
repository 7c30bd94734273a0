import logging, os, time, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.basicConfig(level=logging.DEBUG, stream=sys.stderr)
import numpy as np, torch, Htool
from htool_python_amd.workloads import points_in_sphere
kernel = sys.argv[1] if len(sys.argv) > 1 else "laplace"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
eps = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
leaf = int(sys.argv[5]) if len(sys.argv) > 5 else 100
pts=points_in_sphere(n, seed=0)
Htool.set_num_threads(16)
for rep in range(reps):
    t0=time.time()
    cb=Htool.ClusterTreeBuilder(); cb.set_maximal_leaf_size(leaf)
    cl=cb.create_cluster_tree(pts,2,size_of_partition=1)
    t1=time.time()
    if kernel == "helmholtz":
        gen=Htool.ComplexNativeGenerator("helmholtz",pts,pts,10.0)
        b=Htool.ComplexHMatrixTreeBuilder(eps,10.0,"N","N")
    else:
        gen=Htool.NativeGenerator("laplace",pts,pts,0.0)
        b=Htool.HMatrixTreeBuilder(eps,10.0,"N","N")
    torch.cuda.synchronize(); t2=time.time()
    H=b.build(gen,cl,cl)
    torch.cuda.synchronize(); t3=time.time()
    print(f"rep {rep}: cluster {t1-t0:.3f} gen {t2-t1:.3f} build {t3-t2:.3f}", file=sys.stderr)
    del H
    time.sleep(3)
