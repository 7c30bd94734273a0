"""Build an H-matrix and multiply with it (counterpart of the reference's example/use_hmatrix.py, written
for this repository: same API calls, numpy-vectorised callback generator, optional native generator).

    python examples/hmatrix_quickstart.py [--native] [--size 1000] [--plot out.png]
"""
import argparse
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import Htool  # noqa: E402
from htool_python_amd.workloads import points_in_sphere  # noqa: E402


class Generator(Htool.VirtualGenerator):
    """A(i, j) = 1 / (0.1 + |x_i - y_j|), filled block-wise in user numbering."""

    def __init__(self, target_points, source_points):
        super().__init__()
        self.target_points, self.source_points = target_points, source_points

    def build_submatrix(self, J, K, mat):
        d = self.target_points[:, J][:, :, None] - self.source_points[:, K][:, None, :]
        mat[:, :] = 1.0 / (0.1 + np.sqrt((d * d).sum(axis=0)))

    def mat_vec(self, x):
        out = np.empty((self.target_points.shape[1],) + x.shape[1:])
        for a in range(0, len(out), 1024):
            d = self.target_points[:, a:a + 1024][:, :, None] - self.source_points[:, None, :]
            out[a:a + 1024] = (1.0 / (0.1 + np.sqrt((d * d).sum(axis=0)))) @ x
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1000)
    ap.add_argument("--native", action="store_true", help="evaluate the kernel on the GPU (Htool.NativeGenerator)")
    ap.add_argument("--plot", default="")
    args = ap.parse_args()
    logging.basicConfig(level=logging.INFO)

    size, eta, epsilon = args.size, 10, 0.1
    coordinates = points_in_sphere(size)
    cluster_tree_builder = Htool.ClusterTreeBuilder()
    cluster_tree_builder.set_maximal_leaf_size(50)
    target_cluster = cluster_tree_builder.create_cluster_tree(coordinates, 2)
    source_cluster = cluster_tree_builder.create_cluster_tree(coordinates, 2)

    generator = Generator(coordinates, coordinates)
    device_generator = Htool.NativeGenerator("inv_delta", coordinates, coordinates, 0.1) if args.native else generator
    hmatrix = Htool.HMatrixTreeBuilder(epsilon, eta, "S", "L").build(device_generator, target_cluster, source_cluster)

    np.random.seed(0)
    x = np.random.rand(size)
    y = hmatrix * x
    y_dense = generator.mat_vec(x)
    print("matvec error", np.linalg.norm(y - y_dense) / np.linalg.norm(y_dense), "epsilon", epsilon)
    X = np.random.rand(size, 2)
    Y = hmatrix @ X
    Y_dense = generator.mat_vec(X)
    print("matmat error", np.linalg.norm(Y - Y_dense) / np.linalg.norm(Y_dense), "epsilon", epsilon)
    print(hmatrix.shape)
    print(hmatrix.get_tree_parameters())
    print(hmatrix.get_local_information())

    if args.plot:
        import matplotlib

        matplotlib.use("Agg")
        import matplotlib.pyplot as plt

        fig = plt.figure()
        ax1 = fig.add_subplot(2, 2, 1, projection="3d")
        ax2 = fig.add_subplot(2, 2, 2, projection="3d")
        ax4 = fig.add_subplot(2, 2, 4)
        Htool.plot(ax1, target_cluster, coordinates, 1)
        Htool.plot(ax2, target_cluster, coordinates, 2)
        Htool.plot(ax4, hmatrix)
        fig.savefig(args.plot)
        print("figure written to", args.plot)
    assert np.linalg.norm(y - y_dense) / np.linalg.norm(y_dense) < epsilon


if __name__ == "__main__":
    main()
