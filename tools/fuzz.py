"""Randomised stress of build + product (+ recompression, copy, multi-RHS, small arenas) against sampled exact rows.

    PYTHONPATH=. python tools/fuzz.py [seconds] [seed]

Every case prints one line; a failing case prints FAIL with its parameters (they reproduce it)."""
import copy
import os
import sys
import time

import numpy as np


def exact_rows(kind, pts_t, pts_s, x, p0, rows):
    d = np.sqrt(((pts_t[:, rows][:, :, None] - pts_s[:, None, :]) ** 2).sum(0))
    with np.errstate(divide="ignore", invalid="ignore"):
        if kind == "inv_delta":
            A = 1.0 / (p0 + d)
        elif kind == "laplace":
            A = np.where(d > 0, 1.0 / (4 * np.pi * d), 0.0)
        else:
            A = np.where(d > 0, np.exp(1j * p0 * d) / (4 * np.pi * d), 0.0)
    return A @ x


def main(budget, seed, only_case=None, with_oracle=False):
    import Htool

    if os.environ.get("FUZZ_DEBUG_LOG"):
        import logging

        logging.basicConfig(level=logging.DEBUG, stream=sys.stderr)

    rng = np.random.RandomState(seed)
    t_end = time.time() + budget
    n_case = n_fail = 0
    while time.time() < t_end:
        n_case += 1
        dim = int(rng.choice([2, 3]))
        n = int(10 ** rng.uniform(2.5, float(os.environ.get("FUZZ_MAX_LOG10_N", "5.2"))))
        square = rng.rand() < 0.75
        ns = n if square else int(n * rng.uniform(0.3, 1.5))
        leaf = int(rng.choice([3, 5, 10, 16, 33, 50, 64, 100, 130, 257, 600]))
        eta = float(rng.choice([0.7, 3.0, 10.0, 100.0]))
        eps = float(10 ** rng.uniform(-8, -2))
        kind = str(rng.choice(["inv_delta", "laplace", "helmholtz"]))
        cplx = kind == "helmholtz"
        sym = "N" if not square else str(rng.choice(["N", "S"]))
        uplo = "N" if sym == "N" else str(rng.choice(["L", "U"]))
        one_tri = bool(rng.rand() < 0.7)
        children = int(rng.choice([2, 2, 2, 3, 4]))
        strategy = str(rng.choice(["PCARegular", "PCAGeometric", "BoundingBoxRegular", "BoundingBoxGeometric"]))
        arena = None if rng.rand() < 0.6 else int(10 ** rng.uniform(1.3, 2.7))
        if arena is not None and not os.environ.get("FUZZ_TINY_ARENAS"):
            # the forced arena is there to exercise multi-round builds, not thousands of rounds: seed 42 case 43 (0.49 M x 0.54 M points,
            # complex, eps 1e-7, 76 MB) needed 3848 rounds and 143 s for the build alone (gpurun_out/r03e/fuzz_42_43.log)
            arena = max(arena, int(max(n, ns) / 400))
        recompress = rng.rand() < 0.3 and eps <= 1e-4
        shape = str(rng.choice(["ball", "cube", "sheet", "clustered"]))
        part = int(rng.choice([2, 3, 4, 8])) if (square and sym == "N" and rng.rand() < 0.35) else 1   # rows of one partition (a rank's share)
        which = int(rng.randint(0, part))
        label = dict(part=part, which=which, dim=dim, n=n, ns=ns, leaf=leaf, eta=eta, eps=float(f"{eps:.2e}"), kind=kind, sym=sym, uplo=uplo, one_tri=one_tri, children=children,
                     strategy=strategy, arena_mb=arena, recompress=bool(recompress), shape=shape, seed=seed, case=n_case)

        def cloud(m):
            if shape == "ball":
                p = rng.randn(dim, m); p /= np.linalg.norm(p, axis=0); return p * rng.rand(m) ** (1.0 / dim)
            if shape == "cube":
                return rng.rand(dim, m)
            if shape == "sheet":
                p = rng.rand(dim, m); p[-1] *= 1e-3; return p
            centres = rng.rand(dim, 8)
            return centres[:, rng.randint(0, 8, m)] + 0.02 * rng.randn(dim, m)

        try:
            pt = np.asfortranarray(cloud(n))
            ps = pt if square else np.asfortranarray(cloud(ns) + (0.3 if rng.rand() < 0.5 else 0.0))
            p0 = 0.1 if kind == "inv_delta" else (float(rng.uniform(1, 8)) if cplx else 0.0)
            x = rng.rand(ns) + (1j * rng.rand(ns) if cplx else 0)
            rows = rng.choice(n, min(n, 64), replace=False)
            pick_rng = np.random.RandomState(rng.randint(0, 2 ** 31 - 1))
            case_cols = pick_rng.randint(0, 5)   # 3 .. 7 right-hand sides: sweeps of 8 / 4 / 2 / 1 columns   # (drawn here so that a replay consumes the same stream)
            if only_case is not None and n_case != only_case:
                if os.environ.get("FUZZ_LIST"):  # list the parameters of the cases a replay passes over (no build: works without a GPU)
                    print("skip", label, flush=True)
                if n_case > only_case:
                    break
                continue  # replay: the random stream has been advanced exactly as in the original run
            if os.environ.get("FUZZ_DUMP"):  # the inputs of the replayed case for an analysis elsewhere (tools/sheet_analysis.py); no build
                np.savez(os.environ["FUZZ_DUMP"], pt=pt, ps=ps, x=x, rows=rows, p0=p0, label=repr(label))
                print("dumped", label, flush=True)
                return 0
            if arena is not None:
                os.environ["HTOOL_BUILD_ARENA_MB"] = str(arena)
            else:
                os.environ.pop("HTOOL_BUILD_ARENA_MB", None)
            cb = Htool.ClusterTreeBuilder()
            cb.set_maximal_leaf_size(leaf)
            cb.set_partitioning_strategy(getattr(Htool, strategy)())
            ct = cb.create_cluster_tree(pt, children, size_of_partition=part)
            cs = ct if square else cb.create_cluster_tree(ps, children, size_of_partition=1)
            Builder = Htool.ComplexHMatrixTreeBuilder if cplx else Htool.HMatrixTreeBuilder
            Gen = Htool.ComplexNativeGenerator if cplx else Htool.NativeGenerator
            b = Builder(eps, eta, sym, uplo)
            b.set_symmetric_storage(one_tri)
            confirm = int(os.environ.get("FUZZ_CONFIRM", "0"))  # confirmation steps of the ACA stopping test (0 = the reference's rule)
            b.set_aca_confirmation_steps(confirm)
            t0 = time.time()
            if part > 1:   # the per-rank operator of a row split: local rows in cluster order, x in user numbering
                H = b.build(Gen(kind, pt, ps, p0), ct, cs, which)
                sub = ct.get_cluster_on_partition(which)
                perm = np.asarray(ct.get_permutation())
                local_users = perm[sub.get_offset(): sub.get_offset() + sub.get_size()]
                pick = pick_rng.choice(len(local_users), min(len(local_users), 64), replace=False)
                y_local = H * x
                assert y_local.shape == (sub.get_size(),), y_local.shape
                y = np.zeros(n, dtype=y_local.dtype)
                y[local_users] = y_local
                rows = local_users[pick]
            else:
                H = b.build(Gen(kind, pt, ps, p0), ct, cs)
                y = H * x
            ye = exact_rows(kind, pt, ps, x, p0, rows)
            scale = np.linalg.norm(ye) + 1e-300
            err = np.linalg.norm(y[rows] - ye) / scale
            tol = 20 * eps   # sampled rows: the Frobenius-type bound of the operator is looser row by row
            ok = np.all(np.isfinite(y)) and err < tol
            X = np.asfortranarray(np.stack([x, -x, 0.5 * x, x, 2 * x, x, -x][: int(3 + case_cols)], axis=1))
            Y = H @ X
            y_ref = y if part == 1 else y[local_users]
            ok = ok and np.allclose(Y[:, 0], y_ref, rtol=1e-11, atol=1e-13 * scale) and np.allclose(Y[:, 1], -y_ref, rtol=1e-11, atol=1e-13 * scale)
            # more than 8 columns: the 16-wide sweep on the matrix cores (operators that store both triangles), real and complex
            X16 = np.asfortranarray(np.stack([x * (1.0 + 0.1 * c) for c in range(11)], axis=1))
            Y16 = H @ X16
            ok = ok and np.allclose(Y16[:, 0], y_ref, rtol=1e-10, atol=1e-12 * scale) and np.allclose(Y16[:, 10], 2.0 * y_ref, rtol=1e-10, atol=1e-12 * scale)
            # transposed product (tables made on first use; one-triangle storage answers with the stored operator) against exact
            # entries -- the kernels are functions of the distance, so (A^T w)[j] = sum_i K(s_j, t_i) w_i
            t_pts = pt if part == 1 else np.asfortranarray(pt[:, local_users])
            w = pick_rng.rand(t_pts.shape[1]) + (1j * pick_rng.rand(t_pts.shape[1]) if cplx else 0)
            cols = pick_rng.choice(ns, min(ns, 48), replace=False)
            z = H.transposed_mul(w, "T")
            ze = exact_rows(kind, ps, t_pts, w, p0, cols)
            zscale = np.linalg.norm(ze) + 1e-300
            errT = np.linalg.norm(z[cols] - ze) / zscale
            ok = ok and z.shape == (ns,) and np.all(np.isfinite(z)) and errT < tol
            ok = ok and np.array_equal(H * x, y_ref)  # the direct product after the tables were written again
            # ... and eleven columns of it: the 16-wide transposed sweep on the matrix cores
            W16 = np.asfortranarray(np.stack([w * (1.0 + 0.1 * c) for c in range(11)], axis=1))
            Z16 = np.asarray(H.transposed_mul(W16, "T"))
            ok = ok and np.allclose(Z16[:, 0], z, rtol=1e-10, atol=1e-12 * zscale) and np.allclose(Z16[:, 10], 2.0 * z, rtol=1e-10, atol=1e-12 * zscale)
            lhs, rhs = np.sum(w * y_ref), np.sum(z * x)
            ok = ok and abs(lhs - rhs) <= 1e-9 * (abs(lhs) + np.linalg.norm(w) * np.linalg.norm(y_ref) * 1e-3)
            if recompress:
                H2 = copy.deepcopy(H)
                Htool.recompression(H2, max(eps * 10, 1e-6))
                y2 = H2 * x
                if part > 1:
                    full2 = np.zeros(n, dtype=y2.dtype)
                    full2[local_users] = y2
                    y2 = full2
                err2 = np.linalg.norm(y2[rows] - ye) / scale
                ok = ok and np.all(np.isfinite(y2)) and err2 < 20 * max(eps * 10, 1e-6) + tol
                ok = ok and np.array_equal(H * x, y_ref)  # the copy was recompressed, not the original
            if with_oracle and part == 1 and not os.environ.get("FUZZ_NO_ORACLE"):  # the CPU restatement on the same input: is the error the algorithm's or the engine's?
                from oracle import oracle as O
                sid = {"PCARegular": 0, "PCAGeometric": 1, "BoundingBoxRegular": 2, "BoundingBoxGeometric": 3}[strategy]
                oc = O.Cluster(pt, n_children=children, size_of_partition=1, max_leaf=leaf, strategy=sid)
                ocs = oc if square else O.Cluster(ps, n_children=children, size_of_partition=1, max_leaf=leaf, strategy=sid)
                OH = O.HMatrix(oc, ocs, {"inv_delta": 0, "laplace": 1, "helmholtz": 2}[kind], p0, is_complex=cplx, eps=eps, eta=eta, symmetry=sym, uplo=uplo, confirm=confirm)
                yo = OH.matvec(x)
                print(f"   oracle: err {np.linalg.norm(yo[rows] - ye) / scale:.2e}, |y - y_oracle|/|y| {np.linalg.norm(y - yo) / np.linalg.norm(yo):.2e}", flush=True)
            print(("ok  " if ok else "FAIL"), f"{time.time() - t0:6.2f}s err {err:.2e} errT {errT:.2e}", label, flush=True)
            if not ok:
                n_fail += 1
        except Exception as e:  # noqa: BLE001
            n_fail += 1
            print("FAIL (exception)", repr(e)[:300], label, flush=True)
    print(f"cases {n_case}, failures {n_fail}", flush=True)
    return n_fail


if __name__ == "__main__":
    # fuzz.py seconds seed [case]  -- with a case number: replay that case only, next to the CPU oracle
    case = int(sys.argv[3]) if len(sys.argv) > 3 else None
    sys.exit(1 if main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0, case, case is not None) else 0)
