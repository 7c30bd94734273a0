"""Robustness sweep over degenerate inputs (tiny, identical, collinear, duplicated points; one dense block; leaf size 1;
every storage mode; the three native kernels): build + product + two-column product against the exact dense operator.
`run()` returns the list of failing cases; used by tests/test_gpu_hmatrix.py::test_degenerate_inputs_sweep."""
import itertools
import sys

import numpy as np


def _cases(rng):
    yield "n1", rng.rand(3, 1)
    yield "n2", rng.rand(3, 2)
    yield "n7", rng.rand(3, 7)
    yield "n33", rng.rand(3, 33)
    yield "identical50", np.tile(rng.rand(3, 1), (1, 50))
    yield "collinear300", np.vstack([np.linspace(0, 1, 300), np.zeros(300), np.zeros(300)])
    yield "dup200", np.repeat(rng.rand(3, 100), 2, axis=1)
    yield "plane2d_500", rng.rand(2, 500)
    yield "n1000", rng.rand(3, 1000)


def _exact(kind, pts):
    d = np.sqrt(((pts[:, :, None] - pts[:, None, :]) ** 2).sum(0))
    with np.errstate(divide="ignore", invalid="ignore"):
        if kind == "inv_delta":
            return 1.0 / (0.1 + d)
        if kind == "laplace":
            return np.where(d > 0, 1.0 / (4 * np.pi * d), 0.0)
        return np.where(d > 0, np.exp(1j * 3.0 * d) / (4 * np.pi * d), 0.0)


def run(verbose=True):
    import Htool

    rng = np.random.RandomState(0)
    failures = []
    modes = (("N", "N"), ("S", "L"), ("S", "U"))
    kinds = ("inv_delta", "laplace", "helmholtz")
    for (name, pts), leaf, (sym, uplo), kind in itertools.product(_cases(rng), (1, 10, 2000), modes, kinds):
        pts = np.asfortranarray(pts)
        n = pts.shape[1]
        label = (name, leaf, sym, uplo, kind)
        try:
            cb = Htool.ClusterTreeBuilder()
            cb.set_maximal_leaf_size(leaf)
            cl = cb.create_cluster_tree(pts, 2)
            cplx = kind == "helmholtz"
            if cplx:
                H = Htool.ComplexHMatrixTreeBuilder(1e-6, 10.0, sym, uplo).build(Htool.ComplexNativeGenerator(kind, pts, pts, 3.0), cl, cl)
            else:
                H = Htool.HMatrixTreeBuilder(1e-6, 10.0, sym, uplo).build(Htool.NativeGenerator(kind, pts, pts, 0.1 if kind == "inv_delta" else 0.0), cl, cl)
            x = rng.rand(n) + (1j * rng.rand(n) if cplx else 0)
            y = H * x
            ye = _exact(kind, pts) @ x
            err = np.linalg.norm(y - ye) / max(np.linalg.norm(ye), 1e-300)
            Y = H @ np.asfortranarray(np.stack([x, 2 * x], axis=1))
            ok = err < 1e-5 and np.all(np.isfinite(y)) and np.allclose(Y[:, 1], 2 * Y[:, 0], rtol=1e-12, atol=1e-300)
            if not ok:
                failures.append(label + (float(err),))
        except Exception as e:  # noqa: BLE001 -- the sweep reports, the caller asserts
            failures.append(label + (repr(e)[:200],))
        if verbose and failures and failures[-1][:5] == label:
            print("FAIL", failures[-1], flush=True)
    return failures


if __name__ == "__main__":
    bad = run()
    print("done, failures =", len(bad))
    sys.exit(1 if bad else 0)
