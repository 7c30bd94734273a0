"""Generates tests/golden/*.npz.  Run HERE (the build container), where /root/reference exists:

    python tests/golden/make_golden.py

What is pinned: the reference has no golden vectors for the H-matrix path; its tests
compare against the exact dense operator computed from the generator's kernel.  The only reference
code that is importable in this container is example/create_geometry.py (pure numpy); it is imported
here to produce the exact input geometries of the reference's tests, and the expected outputs are the
exact dense products y = A x with A(i,j) = 1/(0.1 + |x_i - y_j|) (the formula of
example/define_generators.py:14-17, evaluated with numpy -- Htool itself cannot be imported).
Only data (inputs and expected outputs) is stored; no reference source text.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True  # the reference tree is read-only for this project: no __pycache__ there
sys.path.insert(0, "/root/reference")
from example.create_geometry import create_partitionned_geometries, create_random_geometries  # noqa: E402


def dense_inv_delta(T, S):
    diff = T[:, :, None] - S[:, None, :]
    return 1.0 / (0.1 + np.sqrt((diff * diff).sum(axis=0)))


def main():
    # tests/test_hmatrix.py:28-38,76-79 -- 500 x 500, d=3, seed 0, x = rand(500), X = rand(500, 2)
    T, S = create_random_geometries(3, 500, 500)
    assert hashlib.sha256(T.tobytes()).hexdigest().startswith("59c1e4eb97438e21")
    assert hashlib.sha256(S.tobytes()).hexdigest().startswith("51cd65409ca0714a")
    np.random.seed(0)
    x = np.random.rand(500)
    np.random.seed(0)
    X = np.random.rand(500, 2)
    A = dense_inv_delta(T, S)
    Asym = dense_inv_delta(T, T)
    np.savez_compressed(os.path.join(HERE, "hmatrix_500.npz"), target=T, source=S, x=x, X=X, y=A @ x, Y=A @ X, y_sym=Asym @ x, Y_sym=Asym @ X)

    # 2-D variant of the same generator (tests/test_cluster.py geometry family)
    T2, S2 = create_random_geometries(2, 500, 500)
    np.savez_compressed(os.path.join(HERE, "geometry_2d_500.npz"), target=T2, source=S2)

    # example/use_distributed_operator.py:13-18 with 2 ranks: partitioned geometry 1000 x 1000
    Tp, Sp, part = create_partitionned_geometries(3, 1000, 1000, 2)
    assert part.tolist() == [[0, 500], [500, 500]]
    np.random.seed(0)
    xp = np.random.rand(1000)
    np.savez_compressed(os.path.join(HERE, "partitioned_1000_w2.npz"), target=Tp, source=Sp, partition=part, x=xp, y=dense_inv_delta(Tp, Sp) @ xp)

    # tests/conftest.py:103-138 geometry (partition_type None): 400 x {400, 200}, seed 0, uniform cube
    for d in (2, 3):
        np.random.seed(0)
        Tt = np.random.random((d, 400))
        S400 = np.random.random((d, 400))
        np.random.seed(0)
        _ = np.random.random((d, 400))
        S200 = np.random.random((d, 200))
        np.random.seed(0)
        x400, x200 = np.random.rand(400), None
        np.random.seed(0)
        x200 = np.random.rand(200)
        np.savez_compressed(os.path.join(HERE, f"distributed_400_d{d}.npz"), target=Tt, source400=S400, source200=S200, x400=x400, x200=x200,
                            y400=dense_inv_delta(Tt, S400) @ x400, y200=dense_inv_delta(Tt, S200) @ x200, y_sym=dense_inv_delta(Tt, Tt) @ x400)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
