"""Diagnostic: inside one window of the hierarchical LU, find the first LAUNCH after which the device differs from the CPU checker
(leaves, diagonal inverses, scratch).   python tools/hlu_debug_buckets.py n leaf eta children window_tasks window"""
import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import htool_python_amd  # noqa: F401,E402
import Htool  # noqa: E402
from oracle import hlu as ohlu  # noqa: E402
from tests.test_hlu_cpu import make_case  # noqa: E402

n, leaf, eta, children, wt, win = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
eps, eps_lu = 1e-3, 1e-4
H, cl = make_case(n, leaf, eps, eta, children)
plan = Htool.HLUPlan(cl, H.leaves, eps_lu, window_tasks=wt, cap_factor=2 * np.log(eps_lu) / np.log(eps))
host = ohlu.HostLU(plan, H.leaf_data, eps_lu, run=False)
for w in range(win):
    host.run_window(w)
t, b, g, _, _aux = plan.program(win)
T = np.ascontiguousarray(t).view(ohlu.TASK).ravel()
B = np.ascontiguousarray(b)
names = ["FILL", "APPLY_DENSE", "APPLY_LR", "ADDLR", "FINAL", "DDPROD", "GETRF"]


def snapshot(x):
    c = copy.copy(x)
    for name in ("factor", "diag", "rank", "norm0", "norm2", "counters", "scratch"):
        setattr(c, name, getattr(x, name).copy())
    return c


def differs(a, d):
    out = []
    for i in range(a.n_leaves):
        x, y = a.leaf_dense(i), d.leaf_dense(i)
        e = np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-300)
        if e > 1e-6:
            out.append(("leaf", i, e, a.leaves[i], int(a.rank[i]), int(d.rank[i])))
    for name in ("diag",):
        x, y = getattr(a, name), getattr(d, name)
        bad = np.where(~np.isclose(x, y, rtol=1e-6, atol=1e-9 * max(1.0, np.abs(x).max())))[0]
        if len(bad):
            out.append((name, len(bad), int(bad[0]), int(bad[-1]), float(x[bad[0]]), float(y[bad[0]])))
    if not np.array_equal(a.rank, d.rank):
        out.append(("rank", np.where(a.rank != d.rank)[0][:10]))
    return out


base = snapshot(host)
lo, hi = 0, len(B)
# the first prefix of launches whose result differs
for nb in range(1, len(B) + 1):
    cpu = snapshot(base)
    cpu.run_window(win, max_buckets=nb)
    dev = snapshot(base)
    os.environ["HTOOL_HLU_DEBUG_BUCKETS"] = str(nb)
    plan.debug_execute(win, win, dev.factor, dev.diag, dev.rank, dev.norm0, dev.norm2, dev.counters, scratch=dev.scratch)
    d = differs(cpu, dev)
    if d:
        bk = B[nb - 1]
        print("first differing launch %d of %d: type %s level %d tasks %d..%d" % (nb, len(B), names[bk[0] & 0xffffffff], bk[0] >> 32, bk[1], bk[2]))
        print(d)
        badl = set(q[1] for q in d if q[0] == "leaf")
        for x in T[bk[1]:bk[2]]:
            if badl and x["leaf"] not in badl:
                continue
            print(names[x["type"]], "lev", x["level"], "fl", x["flags"], "leaf", x["leaf"], "kref", x["kref"], "kc", x["kconst"], "m n r0 c0", x["m"], x["n"], x["r0"], x["c0"], "lds", x["a_ld"], x["b_ld"], x["x_ld"], x["y_ld"],
                  "spaces", x["a"] >> 60, x["b"] >> 60, x["x"] >> 60, x["y"] >> 60, "offs", x["a"] & (2**60 - 1), x["b"] & (2**60 - 1), x["x"] & (2**60 - 1), x["y"] & (2**60 - 1),
                  "k now", (cpu.rank[x["kref"]] if x["kref"] >= 0 else None), "leafrec", host.leaves[x["leaf"]] if x["leaf"] >= 0 else None)
        break
else:
    print("no difference in window", win)
