"""Where does the error of the `shape = sheet` fuzz cases come from?  (VERDICT round 2, weak spot 5a.)

    FUZZ_DUMP=/tmp/case.npz FUZZ_MAX_LOG10_N=5.5 PYTHONPATH=. python tools/fuzz.py 100000 61 322    # inputs of the case (no GPU needed)
    PYTHONPATH=. python tools/sheet_analysis.py /tmp/case.npz > profiles/r03_fuzz_sheet_case_61_322.txt

CPU only.  Builds the operator of the dumped case with the C++ oracle, finds the admissible leaves that carry the error of
the sampled rows (exact block minus U V applied to x, leaf by leaf), and puts next to each of the worst ones: the
explicit-residual numpy ACA of oracle/independent.py (a formulation that shares no recurrence with the engines), the SVD
epsilon-rank, and what a residual re-test on unused rows / columns after convergence would have seen."""
import ast
import sys

import numpy as np

from oracle import independent as I
from oracle import oracle as O


def main(path, n_worst=12):
    d = np.load(path, allow_pickle=True)
    label = ast.literal_eval(str(d["label"]))
    pt, ps, x, rows, p0 = d["pt"], d["ps"], d["x"], d["rows"], float(d["p0"])
    kind = {"inv_delta": 0, "laplace": 1, "helmholtz": 2}[label["kind"]]
    eps, eta, leaf = label["eps"], label["eta"], label["leaf"]
    sid = {"PCARegular": 0, "PCAGeometric": 1, "BoundingBoxRegular": 2, "BoundingBoxGeometric": 3}[label["strategy"]]
    print("case:", label)
    O.build()
    O.set_num_threads(O.usable_cpus())
    oc = O.Cluster(pt, n_children=label["children"], size_of_partition=1, max_leaf=leaf, strategy=sid)
    ocs = oc if ps is pt or (ps.shape == pt.shape and np.array_equal(ps, pt)) else O.Cluster(ps, n_children=label["children"], size_of_partition=1, max_leaf=leaf, strategy=sid)
    OH = O.HMatrix(oc, ocs, kind, p0, is_complex=kind == 2, eps=eps, eta=eta, symmetry=label["sym"], uplo=label["uplo"])
    y = OH.matvec(x)
    ye = O.dense_matvec(kind, pt, ps, x, p0, rows=rows)
    err = np.linalg.norm(y[rows] - ye) / np.linalg.norm(ye)
    print(f"CPU oracle on the sampled rows: relative error {err:.3e} = {err / eps:.1f} epsilon  (leaves: {len(OH.leaves)})")
    # error per admissible leaf on ALL its rows: |(A_leaf - U V) x_s| -- ranked
    tperm, sperm = np.asarray(oc.perm), np.asarray(ocs.perm)
    L = OH.leaves
    adm = [i for i in range(len(L)) if L[i][4] >= 0]
    # leaves that touch the sampled rows
    pos_of_user = np.empty(len(tperm), dtype=np.int64)
    pos_of_user[tperm] = np.arange(len(tperm))
    row_pos = set(int(p) for p in pos_of_user[rows])
    touched = [i for i in adm if any(L[i][0] <= p < L[i][0] + L[i][1] for p in row_pos)]
    contrib = []
    for i in touched:
        t_off, m, s_off, n, r = (int(v) for v in L[i])
        if m * n > 40_000_000:
            continue
        A = O.kernel_block(kind, pt[:, tperm[t_off:t_off + m]], ps[:, sperm[s_off:s_off + n]], p0)
        U, V = OH.leaf_data(i)
        E = A - np.asarray(U) @ np.asarray(V)
        contrib.append((np.linalg.norm(E @ x[sperm[s_off:s_off + n]]), np.linalg.norm(E) / np.linalg.norm(A), i))
    contrib.sort(reverse=True)
    total = np.sqrt(sum(c[0] ** 2 for c in contrib))
    print(f"admissible leaves touching the sampled rows: {len(touched)}; the {n_worst} worst carry "
          f"{np.sqrt(sum(c[0] ** 2 for c in contrib[:n_worst])) / max(total, 1e-300):.3f} of the low-rank error (2-norm of the per-leaf error vectors)")
    print("leaf (t_off m s_off n) | engine rank, |A-UV|_F/|A|_F in eps | explicit-residual ACA: rank, error in eps | SVD eps-rank | eps/10-rank | max |residual| on 8 unused rows+cols / (eps |A|_F / sqrt(mn))")
    rng = np.random.RandomState(0)
    for _, rel, i in contrib[:n_worst]:
        t_off, m, s_off, n, r = (int(v) for v in L[i])
        A = O.kernel_block(kind, pt[:, tperm[t_off:t_off + m]], ps[:, sperm[s_off:s_off + n]], p0)
        U, V = (np.asarray(a) for a in OH.leaf_data(i))
        res = I.aca_full_residual(A, eps, transpose_role=t_off > s_off)
        if res is None:
            ind = "rejected"
        else:
            ind = f"{res[0].shape[1]:3d}, {np.linalg.norm(A - res[0] @ res[1]) / np.linalg.norm(A) / eps:7.2f}"
        E = A - U @ V
        # what a post-convergence re-test would see: residual on a few rows / columns that were never pivots
        rr = rng.choice(m, size=min(8, m), replace=False)
        cc = rng.choice(n, size=min(8, n), replace=False)
        probe = max(np.abs(E[rr]).max(), np.abs(E[:, cc]).max())
        scale = eps * np.linalg.norm(A) / np.sqrt(m * n)
        print(f"({t_off:6d} {m:5d} {s_off:6d} {n:5d}) | {r:3d}, {rel / eps:7.2f} | {ind} | {I.svd_rank(A, eps):3d} | {I.svd_rank(A, eps / 10):3d} | {probe / scale:8.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
