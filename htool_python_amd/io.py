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
             shape=np.asarray(hmatrix.shape, dtype=np.int64),
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
        make = _core._complex_hmatrix_from_leaves if bool(f["is_complex"]) else _core._hmatrix_from_leaves
        H = make(target_cluster, source_cluster, float(f["epsilon"]), float(f["eta"]), str(f["symmetry"]), str(f["uplo"]), bool(f["one_triangle"]),
                 int(target_partition_number), f["leaves"], f["offsets"], f["data"])
        if tuple(int(v) for v in f["shape"]) != tuple(H.shape):
            raise RuntimeError(f"load_hmatrix: saved shape {tuple(f['shape'])} but the clusters give {tuple(H.shape)} (wrong target_partition_number?)")
    return H
