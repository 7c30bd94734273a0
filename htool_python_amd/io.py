"""Checkpoint / resume of a built H-matrix ("an .npz save/load of the flattened H-matrix").

    Htool.save_hmatrix(path, hmatrix)                      # leaves, panels and the parameters, one .npz file
    H = Htool.load_hmatrix(path, target_cluster, source_cluster=None)

The file holds the leaf table, every leaf's panels as `leaf_panels_bulk` returns them (downloaded from HBM in one go),
the build parameters and the cluster permutations; loading needs the SAME cluster trees (rebuilt from the same points
and options -- the permutations and every leaf's (offset, size) are checked) and packs the panels straight into the
tile-major layout on the device: no generator call, no ACA.  Useful when the entries come from a slow Python generator.
Sizes: the file is as large as the operator in HBM (a 1 M-point operator is ~85 GB -- this is for what fits the host).
"""
import numpy as np


def _info(hmatrix):
    return dict(hmatrix.get_tree_parameters())


def save_hmatrix(path, hmatrix):
    import Htool

    leaves = np.asarray(hmatrix.leaves(), dtype=np.int32)
    offsets, data = hmatrix.leaf_panels_bulk(np.arange(len(leaves), dtype=np.int64))
    info = _info(hmatrix)
    target, source = hmatrix.get_target_cluster(), hmatrix.get_source_cluster()
    np.savez(path,
             leaves=leaves, offsets=np.asarray(offsets, dtype=np.int64), data=np.asarray(data),
             epsilon=float(info["Epsilon"]), eta=float(info["Eta"]), symmetry=info["Symmetry"], uplo=info["UPLO"],
             one_triangle=bool(hmatrix.is_one_triangle()), is_complex=isinstance(hmatrix, Htool.ComplexHMatrix),
             shape=np.asarray(hmatrix.shape, dtype=np.int64), n_leaves=len(leaves),
             target_offset=int(target.get_offset()), target_size=int(target.get_size()),
             target_permutation=np.asarray(target.get_permutation(), dtype=np.int32),
             source_permutation=np.asarray(source.get_permutation(), dtype=np.int32))


def load_hmatrix(path, target_cluster, source_cluster=None, target_partition_number=-1):
    """target_cluster / source_cluster: the ROOT clusters (as passed to HMatrixTreeBuilder.build); for an operator that was
    built on one partition of the target tree pass the same target_partition_number as then."""
    from . import Htool as _core

    source_cluster = target_cluster if source_cluster is None else source_cluster
    with np.load(path if str(path).endswith(".npz") else str(path) + ".npz", allow_pickle=False) as f:
        same_t = np.array_equal(f["target_permutation"], np.asarray(target_cluster.get_permutation()))
        same_s = np.array_equal(f["source_permutation"], np.asarray(source_cluster.get_permutation()))
        if not (same_t and same_s):
            raise RuntimeError("load_hmatrix: the cluster trees are not the ones the H-matrix was saved with (permutations differ)")
        if "n_leaves" in f and int(f["n_leaves"]) != len(f["leaves"]):
            raise RuntimeError("load_hmatrix: the file is truncated (leaf count does not match the leaf table)")
        part = target_cluster if target_partition_number < 0 else target_cluster.get_cluster_on_partition(target_partition_number)
        if (int(f["target_offset"]), int(f["target_size"])) != (part.get_offset(), part.get_size()):
            raise RuntimeError(f"load_hmatrix: the operator was saved for rows [{int(f['target_offset'])}, +{int(f['target_size'])}) but target_partition_number="
                               f"{target_partition_number} selects [{part.get_offset()}, +{part.get_size()})")
        make = _core._complex_hmatrix_from_leaves if bool(f["is_complex"]) else _core._hmatrix_from_leaves
        H = make(target_cluster, source_cluster, float(f["epsilon"]), float(f["eta"]), str(f["symmetry"]), str(f["uplo"]), bool(f["one_triangle"]),
                 int(target_partition_number), f["leaves"], f["offsets"], f["data"])
        if tuple(int(v) for v in f["shape"]) != tuple(H.shape):
            raise RuntimeError(f"load_hmatrix: saved shape {tuple(f['shape'])} but the clusters give {tuple(H.shape)} (wrong target_partition_number?)")
    return H


# ----------------------------------------------------------------------------------------------
# cluster trees on disk (read_cluster_from, src/htool/clustering/utility.hpp:10; used at tests/conftest.py:446-449)
# ----------------------------------------------------------------------------------------------
# The reference reads two CSV files written by lib/htool's save_cluster_tree ("<prefix>_cluster_tree_properties.csv" and
# "<prefix>_cluster_tree.csv").  lib/htool and its data files are not in the container, so that format cannot be
# reproduced verifiably; this pair of functions uses the package's own, documented layout (round trip tested):
#   properties file:  "key: value" lines -- space_dimension, number_of_points, maximal_leaf_size, number_of_children,
#                     number_of_nodes, and "permutation: i0,i1,..." (cluster position -> user index)
#   tree file:        one line per node, parents before children:
#                     offset,size,depth,parent,first_child,n_children,partition,cx,cy,cz,radius
def save_cluster_to(cluster, properties_file, tree_file):
    ints, dbl = cluster._nodes()
    perm = np.asarray(cluster.get_permutation())
    with open(properties_file, "w") as f:
        f.write(f"space_dimension: {cluster._dimension()}\nnumber_of_points: {len(perm)}\nmaximal_leaf_size: {cluster.get_maximal_leaf_size()}\n")
        f.write(f"number_of_children: {cluster._number_of_children()}\nnumber_of_nodes: {len(ints)}\n")
        f.write("permutation: " + ",".join(str(int(v)) for v in perm) + "\n")
    with open(tree_file, "w") as f:
        for r, d in zip(np.asarray(ints), np.asarray(dbl)):
            f.write(",".join(str(int(v)) for v in r) + "," + ",".join(repr(float(v)) for v in d) + "\n")


def read_cluster_from(properties_file, tree_file):
    """Htool.read_cluster_from(properties_csv, tree_csv) -> Cluster (root).  See save_cluster_to for the file layout.

    NOT interchangeable with upstream files: the reference's function of this name reads the CSV pair lib/htool's
    save_cluster_tree writes (src/htool/clustering/utility.hpp:10); that writer and every sample of its output are absent from
    the container, so this function reads the pair `Htool.save_cluster_to` writes and refuses anything else with a message that
    says so ("Cluster files")."""
    from . import Htool as _core

    props = {}
    with open(properties_file) as f:
        for line in f:
            if ":" in line:
                key, val = line.split(":", 1)
                props[key.strip()] = val.strip()
    try:
        perm = np.array([int(v) for v in props["permutation"].split(",")], dtype=np.int32)
        dim, max_leaf, nch = int(props["space_dimension"]), int(props["maximal_leaf_size"]), int(props["number_of_children"])
    except (KeyError, ValueError) as e:
        raise RuntimeError(f"read_cluster_from: {properties_file} is not a cluster properties file written by this package's Htool.save_cluster_to ({e}); "
                           "cluster files written by upstream htool (save_cluster_tree) are not supported -- rebuild the tree with ClusterTreeBuilder "
                           "and save it with Htool.save_cluster_to")
    rows = np.loadtxt(tree_file, delimiter=",", ndmin=2)
    if rows.shape[1] != 11 or len(perm) != int(props.get("number_of_points", -1)) or len(rows) != int(props.get("number_of_nodes", -1)):
        raise RuntimeError("read_cluster_from: the tree file does not match the properties file")
    return _core._cluster_from_tables(dim, max_leaf, nch, perm, rows[:, :7].astype(np.int32), np.ascontiguousarray(rows[:, 7:]))
