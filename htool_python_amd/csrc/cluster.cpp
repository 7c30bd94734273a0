// cluster.cpp -- geometric cluster tree, built level by level (OpenMP over the nodes of a level).
//
// Replaces htool::ClusterTreeBuilder<double>::create_cluster_tree as called from
// src/htool/clustering/cluster_tree_builder.hpp:23,39,56 (lib/htool itself is not vendored;
// algorithm per SURVEY.md Appendix A.2):
//   centre = weighted mean, radius = max(|p - c| + radii), split direction = principal axis of
//   the weighted covariance (PCA*) or longest bounding-box edge (BoundingBox*), points sorted
//   along it and cut in equal counts (Regular) or equal widths (Geometric); a node stays a leaf
//   when a child would be smaller than maximal_leaf_size; depth-1 children are the partition.
#include "cluster.hpp"

#include <algorithm>
#include <parallel/algorithm>
#include <omp.h>
#include <cmath>
#include <numeric>

namespace hm {

ClusterHandle *ClusterTree::handle(int node) {
    if (handles.size() < offset.size()) handles.resize(offset.size());
    if (!handles[node]) handles[node].reset(new ClusterHandle{this, node});
    return handles[node].get();
}

namespace {

struct Geometry {
    double c[3];
    double radius;
};

// Sums over the points of a node (weighted mean, covariance) are formed in BLOCKS of SUM_BLOCK consecutive points of the
// cluster order: every block is summed from zero, the block sums are added in block order.  A node of at most SUM_BLOCK points
// is one block, i.e. the plain running sum; the large nodes at the top of the tree -- where a level has fewer nodes than threads
// and these passes used to run on one thread -- have their blocks summed by all threads, with the same result whatever their number.
constexpr int SUM_BLOCK = 4096;

Geometry compute_geometry(const ClusterTree &T, const double *radii, const double *weights, int off, int sz, bool parallel = false) {
    const int d = T.dim;
    Geometry g{{0, 0, 0}, 0};
    const int nb = (sz + SUM_BLOCK - 1) / SUM_BLOCK;
    std::vector<double> part((size_t)std::max(nb, 1) * 4, 0.0); // per block: sum w x, sum w y, sum w z, sum w
    const bool par = parallel && nb > 1;
#pragma omp parallel for schedule(static) if (par)
    for (int b = 0; b < nb; b++) {
        double c[3] = {0, 0, 0}, wsum = 0;
        const int i1 = std::min(sz, (b + 1) * SUM_BLOCK);
        for (int i = b * SUM_BLOCK; i < i1; i++) {
            int u = T.perm[off + i];
            double w = weights ? weights[u] : 1.0;
            wsum += w;
            for (int k = 0; k < d; k++) c[k] += w * T.coords[(size_t)u * d + k];
        }
        for (int k = 0; k < 3; k++) part[(size_t)b * 4 + k] = c[k];
        part[(size_t)b * 4 + 3] = wsum;
    }
    double wsum = 0;
    for (int b = 0; b < nb; b++) {
        for (int k = 0; k < d; k++) g.c[k] += part[(size_t)b * 4 + k];
        wsum += part[(size_t)b * 4 + 3];
    }
    if (wsum != 0)
        for (int k = 0; k < d; k++) g.c[k] /= wsum;
    double radius = 0; // (a maximum: the same in any order)
#pragma omp parallel for schedule(static) reduction(max : radius) if (par)
    for (int i = 0; i < sz; i++) {
        int u = T.perm[off + i];
        double s = 0;
        for (int k = 0; k < d; k++) {
            double t = T.coords[(size_t)u * d + k] - g.c[k];
            s += t * t;
        }
        double r = std::sqrt(s) + (radii ? radii[u] : 0.0);
        if (r > radius) radius = r;
    }
    g.radius = radius;
    return g;
}

// dominant eigenvector of a symmetric d x d matrix (d <= 3), cyclic Jacobi
void dominant_axis(double a[3][3], int d, double dir[3]) {
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 50; sweep++) {
        double off = 0;
        for (int i = 0; i < d; i++)
            for (int j = i + 1; j < d; j++) off += a[i][j] * a[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < d; p++)
            for (int q = p + 1; q < d; q++) {
                if (std::fabs(a[p][q]) < 1e-300) continue;
                double theta = (a[q][q] - a[p][p]) / (2 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                double cs = 1 / std::sqrt(t * t + 1), sn = t * cs;
                for (int k = 0; k < d; k++) {
                    double x = a[k][p], y = a[k][q];
                    a[k][p] = cs * x - sn * y;
                    a[k][q] = sn * x + cs * y;
                }
                for (int k = 0; k < d; k++) {
                    double x = a[p][k], y = a[q][k];
                    a[p][k] = cs * x - sn * y;
                    a[q][k] = sn * x + cs * y;
                }
                for (int k = 0; k < d; k++) {
                    double x = v[k][p], y = v[k][q];
                    v[k][p] = cs * x - sn * y;
                    v[k][q] = sn * x + cs * y;
                }
            }
    }
    int best = 0;
    for (int i = 1; i < d; i++)
        if (a[i][i] > a[best][best]) best = i;
    for (int k = 0; k < 3; k++) dir[k] = k < d ? v[k][best] : 0.0;
    for (int k = 0; k < d; k++)
        if (std::fabs(dir[k]) > 1e-14) {
            if (dir[k] < 0)
                for (int q = 0; q < d; q++) dir[q] = -dir[q];
            break;
        }
}

// sorts perm[off, off+sz) along the split direction, returns the piece sizes
// parallel_sort: the caller is not inside a parallel region (top levels of the tree: few, large nodes) -- the sort then
// uses all threads; a stable sort has a unique result, so the permutation does not depend on the thread count
std::vector<int> split_range(ClusterTree &T, const double *weights, int off, int sz, const double centre[3], int pieces, int strategy, bool parallel_sort = false) {
    const int d = T.dim;
    double dir[3] = {1, 0, 0};
    const bool pca = strategy == 0 || strategy == 1, regular = strategy == 0 || strategy == 2;
    if (pca) {
        double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        const int nb = (sz + SUM_BLOCK - 1) / SUM_BLOCK; // blocked sums, see compute_geometry
        std::vector<double> part((size_t)std::max(nb, 1) * 9, 0.0);
        const bool par_sum = parallel_sort && nb > 1;
#pragma omp parallel for schedule(static) if (par_sum)
        for (int b = 0; b < nb; b++) {
            double cb[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            const int i1 = std::min(sz, (b + 1) * SUM_BLOCK);
            for (int i = b * SUM_BLOCK; i < i1; i++) {
                int u = T.perm[off + i];
                double w = weights ? weights[u] : 1.0, t[3];
                for (int k = 0; k < d; k++) t[k] = T.coords[(size_t)u * d + k] - centre[k];
                for (int p = 0; p < d; p++)
                    for (int q = 0; q < d; q++) cb[p][q] += w * t[p] * t[q];
            }
            for (int p = 0; p < 3; p++)
                for (int q = 0; q < 3; q++) part[(size_t)b * 9 + p * 3 + q] = cb[p][q];
        }
        for (int b = 0; b < nb; b++)
            for (int p = 0; p < d; p++)
                for (int q = 0; q < d; q++) cov[p][q] += part[(size_t)b * 9 + p * 3 + q];
        dominant_axis(cov, d, dir);
    } else {
        double lo[3], hi[3];
        for (int k = 0; k < 3; k++) lo[k] = 1e300, hi[k] = -1e300;
        double lo0 = 1e300, lo1 = 1e300, lo2 = 1e300, hi0 = -1e300, hi1 = -1e300, hi2 = -1e300; // (extrema: the same in any order)
#pragma omp parallel for schedule(static) reduction(min : lo0, lo1, lo2) reduction(max : hi0, hi1, hi2) if (parallel_sort && sz >= 65536)
        for (int i = 0; i < sz; i++) {
            int u = T.perm[off + i];
            const double *x = &T.coords[(size_t)u * d];
            lo0 = std::min(lo0, x[0]); hi0 = std::max(hi0, x[0]);
            if (d > 1) { lo1 = std::min(lo1, x[1]); hi1 = std::max(hi1, x[1]); }
            if (d > 2) { lo2 = std::min(lo2, x[2]); hi2 = std::max(hi2, x[2]); }
        }
        lo[0] = lo0; lo[1] = lo1; lo[2] = lo2; hi[0] = hi0; hi[1] = hi1; hi[2] = hi2;
        int best = 0;
        for (int k = 1; k < d; k++)
            if (hi[k] - lo[k] > hi[best] - lo[best]) best = k;
        for (int k = 0; k < 3; k++) dir[k] = k == best;
    }
    std::vector<std::pair<double, int>> key(sz);
    const bool par = parallel_sort && sz >= 65536;
#pragma omp parallel for schedule(static) if (par)
    for (int i = 0; i < sz; i++) {
        int u = T.perm[off + i];
        double s = 0;
        for (int k = 0; k < d; k++) s += (T.coords[(size_t)u * d + k] - centre[k]) * dir[k];
        key[i] = std::make_pair(s, u);
    }
    auto by_projection = [](const std::pair<double, int> &a, const std::pair<double, int> &b) { return a.first < b.first; };
    if (par) __gnu_parallel::stable_sort(key.begin(), key.end(), by_projection);
    else std::stable_sort(key.begin(), key.end(), by_projection);
#pragma omp parallel for schedule(static) if (par)
    for (int i = 0; i < sz; i++) T.perm[off + i] = key[i].second;
    std::vector<int> sizes(pieces, 0);
    if (regular) {
        int base = sz / pieces;
        for (int p = 0; p < pieces; p++) sizes[p] = base;
        sizes[pieces - 1] = sz - base * (pieces - 1);
    } else {
        double lo = key.front().first, hi = key.back().first, w = (hi - lo) / pieces;
        int pos = 0;
        for (int p = 0; p < pieces; p++) {
            double cut = lo + w * (p + 1);
            int start = pos;
            if (p == pieces - 1)
                pos = sz;
            else
                while (pos < sz && key[pos].first < cut) pos++;
            sizes[p] = pos - start;
        }
    }
    return sizes;
}

int push_node(ClusterTree &T, int off, int sz, int depth, int parent, int part, const Geometry &g) {
    T.offset.push_back(off);
    T.size.push_back(sz);
    T.depth.push_back(depth);
    T.parent.push_back(parent);
    T.first_child.push_back(-1);
    T.n_child.push_back(0);
    T.partition.push_back(part);
    T.cx.push_back(g.c[0]);
    T.cy.push_back(g.c[1]);
    T.cz.push_back(g.c[2]);
    T.radius.push_back(g.radius);
    return (int)T.offset.size() - 1;
}

} // namespace

ClusterTree *build_cluster_tree(const ClusterBuildArgs &a) {
    HM_CHECK(a.dim >= 1 && a.dim <= 3, "cluster tree: spatial dimension must be 1, 2 or 3");
    HM_CHECK(a.n_points > 0, "cluster tree: no points");
    HM_CHECK(a.n_children >= 2, "cluster tree: number_of_children must be >= 2");
    std::unique_ptr<ClusterTree> Tp(new ClusterTree);
    ClusterTree &T = *Tp;
    T.n_points = a.n_points;
    T.dim = a.dim;
    T.max_leaf = a.max_leaf;
    T.n_children = a.n_children;
    T.n_partition = a.size_of_partition < 1 ? 1 : a.size_of_partition;
    T.coords.assign(a.coords, a.coords + (size_t)a.n_points * a.dim);
    T.perm.resize(a.n_points);
    std::iota(T.perm.begin(), T.perm.end(), 0);
    const int P = T.n_partition;

    Geometry g0 = compute_geometry(T, a.radii, a.weights, 0, a.n_points, true);
    push_node(T, 0, a.n_points, 0, -1, -1, g0);
    std::vector<int> level; // nodes to try to split next
    if (P == 1) {
        T.partition[0] = 0;
        T.part_nodes.push_back(0);
        level.push_back(0);
    } else {
        std::vector<int> sizes;
        if (a.partition && a.partition_is_local) {
            int total = 0;
            for (int p = 0; p < P; p++) {
                HM_CHECK(a.partition[2 * p] == total, "Wrong format for partition");
                sizes.push_back(a.partition[2 * p + 1]);
                total += a.partition[2 * p + 1];
            }
            HM_CHECK(total == a.n_points, "Wrong format for partition");
        } else if (a.partition) {
            sizes.assign(P, 0);
            for (int i = 0; i < a.n_points; i++) {
                HM_CHECK(a.partition[i] >= 0 && a.partition[i] < P, "Wrong format for partition");
                sizes[a.partition[i]]++;
            }
            std::vector<int> start(P, 0), np(a.n_points);
            for (int p = 1; p < P; p++) start[p] = start[p - 1] + sizes[p - 1];
            for (int i = 0; i < a.n_points; i++) np[start[a.partition[i]]++] = i;
            T.perm.swap(np);
        } else {
            sizes = split_range(T, a.weights, 0, a.n_points, g0.c, P, a.strategy, true);
        }
        T.first_child[0] = 1;
        T.n_child[0] = P;
        int off = 0;
        for (int p = 0; p < P; p++) {
            Geometry g = compute_geometry(T, a.radii, a.weights, off, sizes[p], true);
            int id = push_node(T, off, sizes[p], 1, 0, p, g);
            T.part_nodes.push_back(id);
            level.push_back(id);
            off += sizes[p];
        }
    }

    struct Split {
        bool ok;
        std::vector<int> sizes;
        std::vector<Geometry> geo;
    };
    const int nc = a.n_children;
    while (!level.empty()) {
        std::vector<Split> res(level.size());
        // few, large nodes (top of the tree): one node after the other, each sorted by all threads; later levels: the
        // nodes in parallel, each sorted by one thread
        const bool wide_level = (int)level.size() >= 2 * omp_get_max_threads();
        auto split_node = [&](long q) {
            int id = level[q], off = T.offset[id], sz = T.size[id];
            Split &s = res[q];
            s.ok = false;
            if (sz / nc < T.max_leaf) return;
            std::vector<int> saved(T.perm.begin() + off, T.perm.begin() + off + sz);
            double c[3] = {T.cx[id], T.cy[id], T.cz[id]};
            s.sizes = split_range(T, a.weights, off, sz, c, nc, a.strategy, !wide_level);
            bool small = false;
            for (int v : s.sizes)
                if (v < T.max_leaf) small = true;
            if (small) {
                std::copy(saved.begin(), saved.end(), T.perm.begin() + off);
                return;
            }
            s.ok = true;
            int o = off;
            for (int v : s.sizes) {
                s.geo.push_back(compute_geometry(T, a.radii, a.weights, o, v, !wide_level));
                o += v;
            }
        };
        // (an `omp parallel for if(false)` would still count as an active level and serialise the sorts inside)
        if (wide_level) {
#pragma omp parallel for schedule(dynamic, 1)
            for (long q = 0; q < (long)level.size(); q++) split_node(q);
        } else {
            for (long q = 0; q < (long)level.size(); q++) split_node(q);
        }
        std::vector<int> next;
        for (size_t q = 0; q < level.size(); q++) {
            if (!res[q].ok) continue;
            int id = level[q], o = T.offset[id];
            T.first_child[id] = (int)T.offset.size();
            T.n_child[id] = nc;
            for (int p = 0; p < nc; p++) {
                int ch = push_node(T, o, res[q].sizes[p], T.depth[id] + 1, id, T.partition[id], res[q].geo[p]);
                next.push_back(ch);
                o += res[q].sizes[p];
            }
        }
        level.swap(next);
    }
    return Tp.release();
}

// A cluster tree from its node table and permutation (read_cluster_from, src/htool/clustering/utility.hpp:10): the
// counterpart of htool_cluster_nodes / htool_cluster_permutation.  Everything a build relies on is validated.
ClusterTree *cluster_tree_from_tables(int n_points, int dim, int max_leaf, int n_children, const int *perm, int n_nodes, const int *ints7, const double *doubles4) {
    HM_CHECK(n_points > 0 && n_nodes > 0 && perm && ints7 && doubles4, "cluster tree tables: null or empty argument");
    HM_CHECK(dim >= 1 && dim <= 3, "cluster tree: spatial dimension must be 1, 2 or 3");
    HM_CHECK(max_leaf >= 1 && n_children >= 2, "cluster tree tables: maximal_leaf_size must be >= 1 and number_of_children >= 2");
    std::unique_ptr<ClusterTree> Tp(new ClusterTree);
    ClusterTree &T = *Tp;
    T.n_points = n_points; T.dim = dim; T.max_leaf = max_leaf; T.n_children = n_children;
    T.perm.assign(perm, perm + n_points);
    std::vector<char> seen((size_t)n_points, 0);
    for (int i = 0; i < n_points; i++) {
        HM_CHECK(perm[i] >= 0 && perm[i] < n_points && !seen[perm[i]], "cluster tree tables: the permutation is not a permutation");
        seen[perm[i]] = 1;
    }
    for (int i = 0; i < n_nodes; i++) {
        const int *r = ints7 + 7 * i;
        const double *d = doubles4 + 4 * i;
        T.offset.push_back(r[0]); T.size.push_back(r[1]); T.depth.push_back(r[2]); T.parent.push_back(r[3]);
        T.first_child.push_back(r[4]); T.n_child.push_back(r[5]); T.partition.push_back(r[6]);
        T.cx.push_back(d[0]); T.cy.push_back(d[1]); T.cz.push_back(d[2]); T.radius.push_back(d[3]);
    }
    HM_CHECK(T.offset[0] == 0 && T.size[0] == n_points && T.parent[0] == -1, "cluster tree tables: node 0 is not the root of all points");
    for (int i = 0; i < n_nodes; i++) {
        HM_CHECK(T.size[i] > 0 && T.offset[i] >= 0 && T.offset[i] + T.size[i] <= n_points, "cluster tree tables: node range outside the points");
        if (T.n_child[i] == 0) continue;
        const int f = T.first_child[i], nc = T.n_child[i];
        HM_CHECK(f > i && f + nc <= n_nodes, "cluster tree tables: children must follow their parent in the node table");
        int pos = T.offset[i];
        for (int c = f; c < f + nc; c++) {
            HM_CHECK(T.parent[c] == i && T.depth[c] == T.depth[i] + 1 && T.offset[c] == pos, "cluster tree tables: children do not tile their parent");
            pos += T.size[c];
        }
        HM_CHECK(pos == T.offset[i] + T.size[i], "cluster tree tables: children do not tile their parent");
    }
    // the partition: the depth-1 children when the root carries partition -1, the root itself otherwise
    if (T.partition[0] >= 0 || T.n_child[0] == 0) { T.n_partition = 1; T.part_nodes.push_back(0); }
    else {
        T.n_partition = T.n_child[0];
        for (int c = 0; c < T.n_child[0]; c++) T.part_nodes.push_back(T.first_child[0] + c);
    }
    return Tp.release();
}

} // namespace hm
