"""Diagnostic: device kernels of the hierarchical LU against the CPU checker, window by window; prints the first window whose
leaves differ and the tasks that touch the worst leaf.   python tools/hlu_debug_windows.py n leaf eta children [window_tasks]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import htool_python_amd  # noqa: F401,E402
import Htool  # noqa: E402
from oracle import hlu as ohlu  # noqa: E402
from tests.test_gpu_hlu import device_window  # noqa: E402
from tests.test_hlu_cpu import make_case  # noqa: E402

n, leaf, eta, children = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
wt = int(sys.argv[5]) if len(sys.argv) > 5 else 200
eps, eps_lu = 1e-3, 1e-4
H, cl = make_case(n, leaf, eps, eta, children)
plan = Htool.HLUPlan(cl, H.leaves, eps_lu, window_tasks=wt, cap_factor=2 * np.log(eps_lu) / np.log(eps))
host = ohlu.HostLU(plan, H.leaf_data, eps_lu, run=False)
lr = host.leaves["kind"][:host.n_leaves] == 1
names = ["FILL", "APPLY_DENSE", "APPLY_LR", "ADDLR", "FINAL", "DDPROD", "GETRF"]
for w in range(host.info["windows"]):
    rank_before = host.rank.copy()
    dev = device_window(plan, host, w)
    host.run_window(w)
    errs = np.zeros(host.n_leaves)
    for i in range(host.n_leaves):
        a, b = host.leaf_dense(i), dev.leaf_dense(i)
        errs[i] = np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-300)
    i = int(np.argmax(errs))
    print("window %d: worst leaf %d err %.2e kind %d ranks cpu %d gpu %d; slots differ: %d" % (w, i, errs[i], host.leaves["kind"][i], host.rank[i], dev.rank[i],
                                                                                              int((host.rank != dev.rank).sum())))
    if errs[i] > 1e-3:
        t, b, g, _, _aux = plan.program(w)
        T = np.ascontiguousarray(t).view(ohlu.TASK).ravel()
        print("leaf", host.leaves[i], "rank before", rank_before[i])
        bad = np.where(errs > 1e-3)[0]
        print("bad leaves", bad[:20], errs[bad[:20]])
        for x in T:
            if len(T) < 40:
                print(names[x["type"]], "lev", x["level"], "fl", x["flags"], "leaf", x["leaf"], "kref", x["kref"], "kc", x["kconst"], "m n r0 c0", x["m"], x["n"], x["r0"], x["c0"], "lds", x["a_ld"], x["b_ld"], x["x_ld"], x["y_ld"],
                      "spaces a b x y", x["a"] >> 60, x["b"] >> 60, x["x"] >> 60, x["y"] >> 60, "offs", x["a"] & (2**60 - 1), x["b"] & (2**60 - 1), x["x"] & (2**60 - 1), x["y"] & (2**60 - 1))
            elif x["leaf"] == i and x["type"] in (3, 4, 5):
                k = x["kconst"] if x["kref"] == -1 else rank_before[x["kref"]] if x["kref"] < len(host.leaves) else -1
                print(names[x["type"]], "level", x["level"], "flags", x["flags"], "kref", x["kref"], "k(before)", k, "m n r0 c0", x["m"], x["n"], x["r0"], x["c0"], "x>>60", x["x"] >> 60, "y>>60", x["y"] >> 60,
                      "slot cpu/gpu", (host.rank[x["kref"]], dev.rank[x["kref"]]) if x["kref"] >= 0 else None)
        print("task kinds in the window", {names[k]: int((T["type"] == k).sum()) for k in range(7)})
        break
