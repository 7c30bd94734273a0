// cluster.hpp -- host cluster tree (SoA node table over one shared permutation).
// Mirrors what the reference exposes through Cluster / ClusterTreeBuilder
// (src/htool/clustering/cluster_node.hpp:18-26, cluster_tree_builder.hpp:19-67).
#pragma once
#include <memory>
#include <vector>

#include "common.hpp"

namespace hm {

struct ClusterTree;

// what a C-ABI htool_cluster* points at: (tree, node id)
struct ClusterHandle {
    ClusterTree *tree;
    int node;
};

struct ClusterTree {
    int n_points = 0, dim = 0, max_leaf = 10, n_partition = 1, n_children = 2;
    std::vector<double> coords; // point-major copy (dim doubles per point), user numbering
    std::vector<int> perm;      // perm[i] = user index at cluster position i
    // node table (SoA); node 0 is the root, children of a node are consecutive ids
    std::vector<int> offset, size, depth, parent, first_child, n_child, partition;
    std::vector<double> cx, cy, cz, radius;
    std::vector<int> part_nodes; // node id of partition p
    // stable C-ABI handles, one per node, created on demand by handle()
    std::vector<std::unique_ptr<ClusterHandle>> handles;

    int node_count() const { return (int)offset.size(); }
    bool is_leaf(int id) const { return n_child[id] == 0; }
    ClusterHandle *handle(int node);
};

struct ClusterBuildArgs {
    const double *coords;
    int n_points, dim;
    const double *radii, *weights;
    int n_children, size_of_partition;
    const int *partition;
    bool partition_is_local;
    int max_leaf, strategy;
};

ClusterTree *build_cluster_tree(const ClusterBuildArgs &a);
// the same tree, bit for bit, built on the GPU (cluster_device.hip); needs a HIP device and maximal_leaf_size >= 1
ClusterTree *build_cluster_tree_device(const ClusterBuildArgs &a);
size_t cluster_device_release_workspace();
ClusterTree *cluster_tree_from_tables(int n_points, int dim, int max_leaf, int n_children, const int *perm, int n_nodes, const int *ints7, const double *doubles4);

} // namespace hm
