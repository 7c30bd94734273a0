"""Htool.plot(ax, cluster, coordinates, depth) and Htool.plot(ax, hmatrix)
(reference: src/htool/matplotlib/cluster.hpp:9-72, src/htool/matplotlib/hmatrix.hpp:10-89).
Pure host-side visualisation on top of the introspection entry points; not part of the hot path.
"""
import numpy as np


def _plot_cluster(ax, cluster, coordinates, depth):
    ints, _ = cluster._nodes()
    perm = np.asarray(cluster.get_permutation())
    root = cluster._node_id()
    root_depth = ints[root, 2]
    coordinates = np.asarray(coordinates)
    # nodes of the sub-tree at the requested depth (leaves above it are kept)
    sel, stack = [], [root]
    while stack:
        i = stack.pop()
        if ints[i, 2] - root_depth == depth or ints[i, 5] == 0:
            sel.append(i)
        else:
            stack.extend(range(ints[i, 4], ints[i, 4] + ints[i, 5]))
    colors = np.zeros(coordinates.shape[1])
    mask = np.zeros(coordinates.shape[1], dtype=bool)
    for c, i in enumerate(sorted(sel)):
        idx = perm[ints[i, 0]: ints[i, 0] + ints[i, 1]]
        colors[idx] = c
        mask[idx] = True
    pts = coordinates[:, mask]
    # colour table chosen by the number of clusters shown, as the reference does (cluster.hpp:50-62); the points are
    # handed over positionally (x, y[, z]) without a size argument, so that 3-D points on a plain 2-D Axes behave as
    # they do there (tests/test_cluster.py:39)
    import matplotlib.pyplot as plt
    from matplotlib.colors import Normalize

    n_shown = len(sel)
    name = "Dark2" if n_shown < 9 else "Set1" if n_shown == 9 else "tab10" if n_shown == 10 else "tab20"
    rgba = plt.get_cmap(name)(Normalize(vmin=colors[mask].min(), vmax=colors[mask].max())(colors[mask]))
    ax.scatter(*[pts[k] for k in range(coordinates.shape[0])], c=rgba, marker="o")


def _plot_hmatrix(ax, hmatrix):
    import matplotlib.patches as patches

    leaves = np.asarray(hmatrix.leaves())
    nr, nc = hmatrix.shape
    if len(leaves) == 0:
        return
    r0, c0 = leaves[:, 0].min(), leaves[:, 2].min()
    max_rank = max(int(leaves[:, 4].max()), 1)
    ax.set_xlim(0, nc)
    ax.set_ylim(nr, 0)
    for t_off, m, s_off, n, rank in leaves:
        if rank < 0:
            color = (1.0, 0.0, 0.0)
        else:
            g = 1.0 - 0.7 * rank / max_rank
            color = (0.2, g, 0.2)
        ax.add_patch(patches.Rectangle((s_off - c0, t_off - r0), n, m, facecolor=color, edgecolor="k", linewidth=0.2))
        if rank >= 0 and m * n > 0.002 * nr * nc:
            ax.text(s_off - c0 + n / 2, t_off - r0 + m / 2, str(rank), ha="center", va="center", fontsize=6)


def plot(ax, obj, *args):
    if hasattr(obj, "leaves"):
        return _plot_hmatrix(ax, obj)
    coordinates, depth = args
    return _plot_cluster(ax, obj, coordinates, depth)
