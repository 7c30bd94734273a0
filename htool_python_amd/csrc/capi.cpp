// capi.cpp -- extern "C" surface declared in include/htool_mi355x.h
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <cmath>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../include/htool_mi355x.h"
#include "hmatrix.hpp"

using namespace hm;

#include "capi_internal.hpp"

static thread_local std::string g_err;
std::string &htool_error_slot() { return g_err; }

static inline const ClusterHandle *CH(const htool_cluster *c) { return reinterpret_cast<const ClusterHandle *>(c); }

// Where the cluster tree is built.  Both builders give the same tree bit for bit (tests/test_gpu_cluster_tree.py), so this is a
// question of time only: the GPU for everything but small clouds (its fixed cost is some thirty launches per level),
// HTOOL_CLUSTER_TREE=host|device overrides, HTOOL_CLUSTER_DEVICE_MIN moves the threshold.
static bool cluster_tree_on_device(int n_points, int max_leaf) {
    const char *e = getenv("HTOOL_CLUSTER_TREE");
    if (e && !std::strcmp(e, "host")) return false;
    if (max_leaf < 1 || device_count() == 0) return false;
    if (e && !std::strcmp(e, "device")) return true;
    const char *m = getenv("HTOOL_CLUSTER_DEVICE_MIN");
    return n_points >= (m ? atoi(m) : 32768);
}

static double g_warm_up_s = 0.0;

extern "C" {

double htool_last_warm_up_seconds(void) { return g_warm_up_s; }
const char *htool_last_error(void) { return g_err.c_str(); }
int htool_device_count(void) { return device_count(); }
int htool_set_device(int device) {
    API_BEGIN
    device_select(device);
    if (!(getenv("HTOOL_WARM_UP") && std::string(getenv("HTOOL_WARM_UP")) == "0")) g_warm_up_s = device_warm_up();
    API_END
}
const char *htool_device_name(void) {
    static thread_local std::string s;
    s = device_name();
    return s.c_str();
}

int64_t htool_release_workspace(void) { return (int64_t)device_release_workspace(); }
void htool_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
void htool_set_log_sink(htool_log_sink sink) { set_log_sink(sink); }
void htool_test_logger(void) {
    log_message(LOG_CRITICAL, "Critical message");
    log_message(LOG_ERROR, "Error message");
    log_message(LOG_WARNING, "Warning message");
    log_message(LOG_DEBUG, "Debug message");
    log_message(LOG_INFO, "Info message");
}

// ---- cluster ---------------------------------------------------------------------------------
int htool_cluster_create(const double *coordinates, int n_points, int dim, const double *radii, const double *weights, int number_of_children,
                         int size_of_partition, const int *partition, int partition_is_local, int maximal_leaf_size, int strategy, htool_cluster **out) {
    API_BEGIN
    ClusterBuildArgs a{coordinates, n_points, dim, radii, weights, number_of_children, size_of_partition, partition, partition_is_local != 0, maximal_leaf_size, strategy};
    ClusterTree *T = cluster_tree_on_device(n_points, maximal_leaf_size) ? build_cluster_tree_device(a) : build_cluster_tree(a);
    *out = reinterpret_cast<htool_cluster *>(T->handle(0));
    API_END
}
void htool_cluster_destroy(htool_cluster *root) {
    if (root) delete CH(root)->tree;
}
int htool_cluster_size(const htool_cluster *c) { return CH(c)->tree->size[CH(c)->node]; }
int htool_cluster_offset(const htool_cluster *c) { return CH(c)->tree->offset[CH(c)->node]; }
int htool_cluster_maximal_leaf_size(const htool_cluster *c) { return CH(c)->tree->max_leaf; }
const int *htool_cluster_permutation(const htool_cluster *c, int *n) {
    if (n) *n = CH(c)->tree->n_points;
    return CH(c)->tree->perm.data();
}
const htool_cluster *htool_cluster_on_partition(const htool_cluster *c, int p) {
    ClusterTree *T = CH(c)->tree;
    if (p < 0 || p >= (int)T->part_nodes.size()) {
        g_err = "partition index out of range";
        return nullptr;
    }
    return reinterpret_cast<const htool_cluster *>(T->handle(T->part_nodes[p]));
}
int htool_cluster_dimension(const htool_cluster *c) { return CH(c)->tree->dim; }
int htool_cluster_number_of_children(const htool_cluster *c) { return CH(c)->tree->n_children; }
int htool_cluster_create_from_tables(int n_points, int dim, int maximal_leaf_size, int number_of_children, const int *permutation, int n_nodes,
                                     const int *ints7, const double *doubles4, htool_cluster **out) {
    API_BEGIN
    ClusterTree *T = cluster_tree_from_tables(n_points, dim, maximal_leaf_size, number_of_children, permutation, n_nodes, ints7, doubles4);
    *out = reinterpret_cast<htool_cluster *>(T->handle(0));
    API_END
}
int htool_cluster_node_count(const htool_cluster *c) { return CH(c)->tree->node_count(); }
int htool_cluster_node_id(const htool_cluster *c) { return CH(c)->node; }
void htool_cluster_nodes(const htool_cluster *c, int *ints7, double *doubles4) {
    const ClusterTree &T = *CH(c)->tree;
    for (int i = 0; i < T.node_count(); i++) {
        int *p = ints7 + 7 * i;
        p[0] = T.offset[i]; p[1] = T.size[i]; p[2] = T.depth[i]; p[3] = T.parent[i]; p[4] = T.first_child[i]; p[5] = T.n_child[i]; p[6] = T.partition[i];
        double *q = doubles4 + 4 * i;
        q[0] = T.cx[i]; q[1] = T.cy[i]; q[2] = T.cz[i]; q[3] = T.radius[i];
    }
}

// ---- generators --------------------------------------------------------------------------------
int htool_generator_create_callback(int is_complex, htool_copy_submatrix_fn fn, void *ctx, htool_generator **out) {
    API_BEGIN
    HM_CHECK(fn != nullptr, "generator callback is null");
    htool_generator *g = new htool_generator;
    g->g.is_complex = is_complex != 0;
    g->g.native = false;
    g->g.fn = fn;
    g->g.ctx = ctx;
    *out = g;
    API_END
}
int htool_generator_create_native(int kind, int dim, const double *target_points, int n_target, const double *source_points, int n_source,
                                  double param, htool_generator **out) {
    API_BEGIN
    HM_CHECK(kind >= 0 && kind <= 2, "unknown native kernel kind");
    HM_CHECK(dim >= 1 && dim <= 3, "native generator: dimension must be 1, 2 or 3");
    htool_generator *g = new htool_generator;
    g->g.native = true;
    g->g.kind = kind;
    g->g.dim = dim;
    g->g.param = param;
    g->g.is_complex = kind == HTOOL_KERNEL_HELMHOLTZ;
    g->g.n_target = n_target;
    g->g.n_source = n_source;
    g->g.tpts.assign(target_points, target_points + (size_t)n_target * dim);
    g->g.spts.assign(source_points, source_points + (size_t)n_source * dim);
    *out = g;
    API_END
}
int htool_generator_is_complex(const htool_generator *g) { return g->g.is_complex ? 1 : 0; }
void htool_generator_destroy(htool_generator *g) { delete g; }

void htool_build_params_default(htool_build_params *p) {
    std::memset(p, 0, sizeof(*p));
    p->epsilon = 1e-3;
    p->eta = 10;
    p->symmetry = 'N';
    p->uplo = 'N';
    p->reqrank = -1;
    p->block_tree_consistency = 1;
    p->store_one_triangle = 1;
}

// ---- H-matrix ----------------------------------------------------------------------------------
// the same cluster tree, or two trees built alike on the same points (the reference's example builds the target and the
// source cluster separately, example/use_hmatrix.py:27-28): same permutation and same node table
static bool same_tree(const ClusterTree *T, const ClusterTree *S) {
    if (T == S) return true;
    return T->n_points == S->n_points && T->perm == S->perm && T->offset == S->offset && T->size == S->size && T->first_child == S->first_child &&
           T->n_child == S->n_child && T->cx == S->cx && T->cy == S->cy && T->cz == S->cz && T->radius == S->radius;
}

// leaves handed over by the caller instead of a generator (htool_hmatrix_build_from_leaves)
struct LeafPreset {
    int is_complex;
    int64_t n_leaves;
    const int *leaves5;
    const int64_t *offsets2;
    const void *data;
    int64_t n_elements;
};

static void fill_from_preset(HMatrix &H, const LeafPreset &ps) {
    const ClusterTree &T = *H.tc, &S = *H.sc;
    auto index_nodes = [](const ClusterTree &C) {
        std::map<std::pair<int, int>, int> m; // (offset, size) -> node; the first (shallowest) node of a range wins
        for (int id = 0; id < C.node_count(); id++) m.emplace(std::make_pair(C.offset[id], C.size[id]), id);
        return m;
    };
    const auto tn = index_nodes(T), sn = (&T == &S) ? tn : index_nodes(S);
    H.blocks().clear();
    H.blocks().reserve((size_t)ps.n_leaves);
    for (int64_t i = 0; i < ps.n_leaves; i++) {
        const int *l = ps.leaves5 + 5 * i;
        auto it = tn.find(std::make_pair(l[0], l[1]));
        auto is = sn.find(std::make_pair(l[2], l[3]));
        HM_CHECK(it != tn.end() && is != sn.end(), strprintf("leaf %lld (%d,%d,%d,%d) does not match a pair of cluster nodes", (long long)i, l[0], l[1], l[2], l[3]));
        HM_CHECK(l[0] >= H.row_off && l[0] + l[1] <= H.row_off + H.row_size && l[2] >= H.col_off && l[2] + l[3] <= H.col_off + H.col_size, "leaf outside the operator");
        BlockRec b;
        b.t_node = it->second; b.s_node = is->second;
        b.t_off = l[0]; b.m = l[1]; b.s_off = l[2]; b.n = l[3];
        b.rank = l[4]; b.cap = std::max(l[4], 0); b.batch = -1;
        b.tmp_u = ps.offsets2[2 * i]; b.tmp_v = ps.offsets2[2 * i + 1];
        b.ucol = b.vcol = 0; b.tpos = 0; b.v_obase = 0; b.v_ostride = 0; b.status = 0;
        const int64_t lu = (int64_t)(b.rank >= 0 ? b.rank : b.n) * b.m, lv = b.rank > 0 ? (int64_t)b.rank * b.n : 0;
        HM_CHECK(b.rank >= -1 && b.rank <= std::min(b.m, b.n), "leaf with an impossible rank");
        HM_CHECK(b.tmp_u >= 0 && b.tmp_u + lu <= ps.n_elements && (lv == 0 || (b.tmp_v >= 0 && b.tmp_v + lv <= ps.n_elements)), "leaf data outside the buffer");
        if (H.one_triangle) {
            const bool lower = H.params.uplo == 'L';
            HM_CHECK(lower ? b.s_off < b.t_off + b.m : b.t_off < b.s_off + b.n, "one-triangle storage: a leaf lies in the triangle that is not stored");
        }
        H.blocks().push_back(b);
    }
    // The leaves have to tile the operator: no gaps, no overlaps (a truncated or mismatched file would otherwise give a wrong
    // operator without any error).  Every leaf is a pair of cluster nodes (checked above), so its rows are a union of whole row
    // tiles: for every row tile the column ranges of the leaves covering it must be disjoint and -- when both triangles are
    // stored -- follow each other from the first to the last column.  One-triangle storage: disjointness per row tile, plus
    // the total area with every off-diagonal leaf counted twice.
    struct Piece { int tile, s_off, n; };
    std::vector<Piece> pieces;
    long double area = 0;
    for (const BlockRec &b : H.blocks()) {
        for (int r = H.rtiles.node_tile_begin[b.t_node]; r < H.rtiles.node_tile_end[b.t_node]; r++) pieces.push_back({r, b.s_off, b.n});
        area += (long double)b.m * b.n * ((H.one_triangle && b.t_off != b.s_off) ? 2 : 1);
    }
    std::sort(pieces.begin(), pieces.end(), [](const Piece &x, const Piece &y) { return x.tile != y.tile ? x.tile < y.tile : x.s_off < y.s_off; });
    const int col_end = H.col_off + H.col_size;
    size_t i = 0;
    for (int r = 0; r < H.rtiles.count(); r++) {
        int pos = H.col_off;
        for (; i < pieces.size() && pieces[i].tile == r; i++) {
            HM_CHECK(pieces[i].s_off >= pos, strprintf("leaf table: two leaves overlap (row tile %d, column %d)", r, pieces[i].s_off));
            HM_CHECK(H.one_triangle || pieces[i].s_off == pos, strprintf("leaf table: columns [%d, %d) of row tile %d are not covered by any leaf", pos, pieces[i].s_off, r));
            pos = pieces[i].s_off + pieces[i].n;
        }
        HM_CHECK(H.one_triangle || pos == col_end, strprintf("leaf table: columns [%d, %d) of row tile %d are not covered by any leaf", pos, col_end, r));
    }
    const long double want = (long double)H.row_size * H.col_size;
    HM_CHECK(area == want, strprintf("the leaves cover %.0Lf entries but the operator has %.0Lf: a truncated or mismatched leaf table", area, want));
}

static htool_hmatrix *build_hmatrix(const htool_generator *g, const htool_cluster *target_root, const htool_cluster *source_root,
                                    const htool_build_params *params, int target_partition, int source_partition = -1, const LeafPreset *preset = nullptr) {
    HM_CHECK((g || preset) && target_root && source_root && params, "htool_hmatrix_build: null argument");
    ClusterTree *T = CH(target_root)->tree, *S = CH(source_root)->tree;
    HM_CHECK(params->symmetry == 'N' || params->symmetry == 'S' || params->symmetry == 'H', "symmetry must be 'N', 'S' or 'H'");
    HM_CHECK(params->uplo == 'N' || params->uplo == 'L' || params->uplo == 'U', "UPLO must be 'N', 'L' or 'U'");
    if (g && g->g.native) {
        HM_CHECK(g->g.n_target == T->n_points && g->g.n_source == S->n_points, "native generator: point counts do not match the clusters");
        HM_CHECK(g->g.dim == T->dim && g->g.dim == S->dim, "native generator: dimension does not match the clusters");
    }
    HM_CHECK(device_count() > 0, "no HIP device available: libhtool_mi355x has no CPU fallback (HIP path required)");
    std::unique_ptr<htool_hmatrix> h(new htool_hmatrix);
    HMatrix &H = h->H;
    H.tc = T;
    H.sc = S;
    H.is_complex = preset ? preset->is_complex != 0 : g->g.is_complex;
    H.params.epsilon = params->epsilon;
    H.params.eta = params->eta;
    H.params.symmetry = params->symmetry;
    H.params.uplo = params->uplo;
    H.params.reqrank = params->reqrank;
    HM_CHECK(params->aca_confirm_steps >= 0 && params->aca_confirm_steps <= 8, "aca_confirm_steps must be between 0 and 8");
    H.params.aca_confirm_steps = params->aca_confirm_steps;
    H.params.min_target_depth = params->minimal_target_depth;
    H.params.min_source_depth = params->minimal_source_depth;
    H.params.block_tree_consistency = params->block_tree_consistency;
    if (!params->block_tree_consistency)
        log_message(LOG_DEBUG, "set_block_tree_consistency(False): accepted for compatibility -- upstream the flag governs how the POINTER tree of blocks is "
                               "kept refinable for H-LU; this engine stores the leaves as two flat queues (no parent / child links) and the leaves are the same either way");
    H.params.compress = params->compress;
    H.params.compress_ctx = params->compress_ctx;
    H.params.compress_borrows = params->compress_borrows;
    H.params.dense_blocks = params->dense_blocks;
    H.params.dense_blocks_ctx = params->dense_blocks_ctx;
    if (params->store_one_triangle) {
        const bool eligible = (params->symmetry == 'S' || params->symmetry == 'H') && (params->uplo == 'L' || params->uplo == 'U') && same_tree(T, S) && target_partition < 0 && source_partition < 0;
        if (eligible) { H.params.store_one_triangle = 1; H.one_triangle = true; }
        else if (preset && params->symmetry != 'N' && params->uplo != 'N')
            throw Error("htool_hmatrix_build_from_leaves: the leaves were saved as ONE triangle of a symmetric operator, but this configuration (two cluster "
                        "trees, or a partition) cannot use one-triangle storage -- the other triangle would be lost");
        else if (params->symmetry != 'N') log_message(LOG_DEBUG, "symmetric build restricted to a partition or on two cluster trees: both triangles of the requested rows are stored");
    }
    if (params->transposed_products && !H.one_triangle) H.transposable = true; // (the layout of every batch then carries the slots of the transposed use)
    if (target_partition >= 0) {
        HM_CHECK(target_partition < (int)T->part_nodes.size(), "target_partition_number out of range");
        H.t_root = T->part_nodes[target_partition];
    } else {
        H.t_root = 0;
    }
    H.row_off = T->offset[H.t_root];
    H.row_size = T->size[H.t_root];
    H.s_root = 0;
    if (source_partition >= 0) {
        HM_CHECK(source_partition < (int)S->part_nodes.size(), "source partition number out of range");
        H.s_root = S->part_nodes[source_partition];
    }
    H.col_off = S->offset[H.s_root];
    H.col_size = S->size[H.s_root];
    H.tile_max = H.is_complex ? 64 : 128;
    H.rtiles = make_tiles(*T, H.t_root, H.tile_max);
    H.ctiles = make_tiles(*S, H.s_root, H.tile_max);
    {
        int gp = 512; // source positions per phase-A group (HTOOL_PHASE_A_GROUP=1: one tile per group, the round-1 scheme)
        if (const char *e = getenv("HTOOL_PHASE_A_GROUP")) gp = std::max(1, atoi(e));
        assign_tile_groups(*S, H.s_root, H.ctiles, gp, H.ctile_group);
    }
    h->tch = T->handle(H.t_root);
    h->sch = S->handle(H.s_root);
    double t0 = wall_seconds();
    if (preset) {
        fill_from_preset(H, *preset);
        device_build_from_host(H, preset->data, preset->n_elements);
    } else if (g->g.native) {
        device_build_native(H, g->g);
    } else if (H.is_complex) {
        std::vector<cplx> arena;
        host_fill_blocks<cplx>(g->g, H, arena);
        device_build_from_host(H, arena.data(), (int64_t)arena.size());
    } else {
        std::vector<double> arena;
        host_fill_blocks<double>(g->g, H, arena);
        device_build_from_host(H, arena.data(), (int64_t)arena.size());
    }
    H.build_seconds = wall_seconds() - t0;
    log_message(LOG_INFO, strprintf("H-matrix built: %zu leaves, %.3f s", H.leaf_count(), H.build_seconds));
    return h.release();
}

int htool_hmatrix_build(const htool_generator *g, const htool_cluster *target_root, const htool_cluster *source_root, const htool_build_params *params,
                        int target_partition_number, int partition_number_for_symmetry, htool_hmatrix **out) {
    API_BEGIN
    if (partition_number_for_symmetry >= 0 && partition_number_for_symmetry != target_partition_number)
        log_message(LOG_WARNING, "partition_number_for_symmetry differs from target_partition_number: ignored (a build restricted to a partition stores both triangles of its rows)");
    *out = build_hmatrix(g, target_root, source_root, params, target_partition_number);
    API_END
}
int htool_hmatrix_build_local(const htool_generator *g, const htool_cluster *target_root, const htool_cluster *source_root, const htool_build_params *params,
                              int target_partition_number, int source_partition_number, htool_hmatrix **out) {
    API_BEGIN
    htool_hmatrix *h = build_hmatrix(g, target_root, source_root, params, target_partition_number, source_partition_number);
    h->H.local_numbering = true;
    *out = h;
    API_END
}
int htool_hmatrix_build_from_leaves(const htool_cluster *target_root, const htool_cluster *source_root, const htool_build_params *params,
                                    int is_complex, int target_partition_number, int64_t n_leaves, const int *leaves5,
                                    const int64_t *offsets2, const void *data, int64_t n_elements, htool_hmatrix **out) {
    API_BEGIN
    HM_CHECK(n_leaves >= 0 && (n_leaves == 0 || (leaves5 && offsets2)) && (n_elements == 0 || data), "htool_hmatrix_build_from_leaves: null argument");
    LeafPreset ps{is_complex, n_leaves, leaves5, offsets2, data, n_elements};
    *out = build_hmatrix(nullptr, target_root, source_root, params, target_partition_number, -1, &ps);
    API_END
}
int htool_debug_scan_positions(const int *counts_dev, int64_t n, int64_t *positions_dev, int64_t *totals2) {
    API_BEGIN
    HM_CHECK(n >= 0 && (n == 0 || (counts_dev && positions_dev)) && totals2, "htool_debug_scan_positions: bad argument");
    long long t[2] = {0, 0};
    device_debug_positions(counts_dev, (long long)n, (long long *)positions_dev, t);
    totals2[0] = t[0]; totals2[1] = t[1];
    API_END
}
int htool_debug_sort_pairs(uint32_t *keys_dev, uint32_t *values_dev, int64_t n, int key_bits) {
    API_BEGIN
    HM_CHECK(n >= 0 && (n == 0 || (keys_dev && values_dev)) && key_bits >= 1 && key_bits <= 32, "htool_debug_sort_pairs: bad argument");
    device_debug_sort_pairs(keys_dev, values_dev, (long long)n, key_bits);
    API_END
}
int htool_hmatrix_is_one_triangle(const htool_hmatrix *h) { return h->H.one_triangle ? 1 : 0; }
void htool_hmatrix_destroy(htool_hmatrix *h) { delete h; }
int htool_hmatrix_clone(const htool_hmatrix *h, htool_hmatrix **out) {
    API_BEGIN
    std::unique_ptr<htool_hmatrix> c(new htool_hmatrix);
    const HMatrix &s = h->H;
    HMatrix &d = c->H;
    d.tc = s.tc; d.sc = s.sc; d.t_root = s.t_root; d.row_off = s.row_off; d.row_size = s.row_size; d.is_complex = s.is_complex;
    d.s_root = s.s_root; d.col_off = s.col_off; d.col_size = s.col_size; d.local_numbering = s.local_numbering; d.one_triangle = s.one_triangle;
    d.params = s.params; d.tile_max = s.tile_max; d.rtiles = s.rtiles; d.ctiles = s.ctiles; d.ctile_group = s.ctile_group; d.blocks() = s.blocks(); d.r_elems = s.r_elems;
    d.build_seconds = s.build_seconds; d.n_batches = s.n_batches; d.transposable = s.transposable;
    c->tch = h->tch; c->sch = h->sch;
    device_clone(s, d);
    *out = c.release();
    API_END
}
int htool_hmatrix_recompress(htool_hmatrix *h, double epsilon, int64_t *n_reduced) {
    API_BEGIN
    int64_t c = device_recompress(h->H, epsilon);
    if (n_reduced) *n_reduced = c;
    API_END
}
int htool_hmatrix_is_complex(const htool_hmatrix *h) { return h->H.is_complex ? 1 : 0; }
int htool_hmatrix_nb_rows(const htool_hmatrix *h) { return h->H.row_size; }
int htool_hmatrix_nb_cols(const htool_hmatrix *h) { return h->H.col_size; }
const htool_cluster *htool_hmatrix_target_cluster(const htool_hmatrix *h) { return reinterpret_cast<const htool_cluster *>(h->tch); }
const htool_cluster *htool_hmatrix_source_cluster(const htool_hmatrix *h) { return reinterpret_cast<const htool_cluster *>(h->sch); }

} // extern "C"

template <typename T>
static void axpby(size_t n, T alpha, const T *t, T beta, T *y) {
    for (size_t i = 0; i < n; i++) y[i] = alpha * t[i] + (beta == T(0) ? T(0) : beta * y[i]);
}

static void check_numbering(const HMatrix &H, int numbering) {
    HM_CHECK(numbering >= 0 && numbering <= 3, "numbering must be 0 (user/user), 1 (cluster/cluster), 2 (user in, cluster out) or 3 (cluster in, user out)");
    const bool in_user = numbering == 0 || numbering == 2, out_user = numbering == 0 || numbering == 3;
    HM_CHECK(!out_user || H.t_root == 0, "user-numbered output needs an H-matrix built on the whole target cluster");
    HM_CHECK(!in_user || H.s_root == 0, "user-numbered input needs an H-matrix built on the whole source cluster");
}

static void check_trans(char trans) { HM_CHECK(trans == 'N' || trans == 'T' || trans == 'C', "H-matrix product: trans must be 'N', 'T' or 'C'"); }

static void matvec_scaled(const HMatrix &H, char trans, const void *alpha, const void *x, const void *beta, void *y) {
    // entries of the result: rows of H (trans = 'N') or its columns
    const size_t n = trans == 'N' ? (size_t)(H.t_root == 0 ? H.tc->n_points : H.row_size) : (size_t)H.col_size;
    if (H.is_complex) {
        cplx a = alpha ? *(const cplx *)alpha : cplx(1), b = beta ? *(const cplx *)beta : cplx(0);
        if (a == cplx(1) && b == cplx(0)) { device_matvec_host(H, x, y, trans); return; }
        std::vector<cplx> t(n);
        device_matvec_host(H, x, t.data(), trans);
        axpby<cplx>(n, a, t.data(), b, (cplx *)y);
    } else {
        double a = alpha ? *(const double *)alpha : 1.0, b = beta ? *(const double *)beta : 0.0;
        if (a == 1.0 && b == 0.0) { device_matvec_host(H, x, y, trans); return; }
        std::vector<double> t(n);
        device_matvec_host(H, x, t.data(), trans);
        axpby<double>(n, a, t.data(), b, (double *)y);
    }
}

extern "C" {

int htool_hmatrix_matvec(const htool_hmatrix *h, char trans, const void *alpha, const void *x, const void *beta, void *y) {
    API_BEGIN
    check_trans(trans);
    matvec_scaled(h->H, trans, alpha, x, beta, y);
    API_END
}
int htool_hmatrix_matmat(const htool_hmatrix *h, char trans, const void *alpha, const void *X, int mu, const void *beta, void *Y) {
    API_BEGIN
    check_trans(trans);
    const HMatrix &H = h->H;
    const size_t es = H.is_complex ? 16 : 8;
    size_t nin = (size_t)H.col_size, nout = (size_t)(H.t_root == 0 ? H.tc->n_points : H.row_size);
    if (trans != 'N') std::swap(nin, nout);
    bool plain;
    if (H.is_complex) plain = (!alpha || *(const cplx *)alpha == cplx(1)) && (!beta || *(const cplx *)beta == cplx(0));
    else plain = (!alpha || *(const double *)alpha == 1.0) && (!beta || *(const double *)beta == 0.0);
    if (plain) device_matmat_host(H, X, mu, Y, trans); // all right-hand sides in one sweep of the panels
    else for (int c = 0; c < mu; c++) matvec_scaled(H, trans, alpha, (const char *)X + c * nin * es, beta, (char *)Y + c * nout * es);
    API_END
}
int htool_hmatrix_matmat_device(const htool_hmatrix *h, const void *X_dev, int64_t ldx, void *Y_dev, int64_t ldy, int mu, int numbering, void *stream) {
    API_BEGIN
    check_numbering(h->H, numbering);
    HM_CHECK(mu >= 1, "mu must be >= 1");
    device_matmat_device(h->H, X_dev, (long long)ldx, Y_dev, (long long)ldy, mu, numbering, stream);
    API_END
}
int htool_hmatrix_matmat_device_trans(const htool_hmatrix *h, char trans, const void *X_dev, int64_t ldx, void *Y_dev, int64_t ldy, int mu, int numbering, void *stream) {
    API_BEGIN
    check_trans(trans);
    HM_CHECK(numbering >= 0 && numbering <= 3, "numbering must be 0 (user/user), 1 (cluster/cluster), 2 (user in, cluster out) or 3 (cluster in, user out)");
    if (trans == 'N') check_numbering(h->H, numbering);
    else { // x lives on the target side, y on the source side
        const bool in_user = numbering == 0 || numbering == 2, out_user = numbering == 0 || numbering == 3;
        HM_CHECK(!in_user || h->H.t_root == 0, "user-numbered input of a transposed product needs an H-matrix built on the whole target cluster");
        HM_CHECK(!out_user || h->H.s_root == 0, "user-numbered output of a transposed product needs an H-matrix built on the whole source cluster");
    }
    HM_CHECK(mu >= 1, "mu must be >= 1");
    device_matmat_device(h->H, X_dev, (long long)ldx, Y_dev, (long long)ldy, mu, numbering, stream, trans);
    API_END
}
int htool_hmatrix_matvec_device(const htool_hmatrix *h, const void *x_dev, void *y_dev, int numbering, void *stream) {
    API_BEGIN
    check_numbering(h->H, numbering);
    device_matvec_device(h->H, x_dev, y_dev, numbering, stream);
    API_END
}

} // extern "C"

static void densify(const HMatrix &H, void *out, int user_numbering) {
    const size_t es = H.is_complex ? 16 : 8;
    const int ns = H.col_size;
    const size_t nr = (size_t)(H.t_root == 0 ? H.tc->n_points : H.row_size);
    const int BS = 64;
    const bool local = H.s_root != 0 || H.local_numbering; // local block: both sides are slices in cluster order already
    // leaf by leaf on the device (device_expand.inc) when the dense copy fits there: every panel is read once; the permutation to
    // the caller's numbering (columns; rows too when the operator covers the whole target cluster) is applied here
    static const bool by_products = getenv("HTOOL_DENSE_EXPANSION") && std::string(getenv("HTOOL_DENSE_EXPANSION")) == "products";
    if (!by_products && nr == (size_t)H.row_size) {
        const bool permute_cols = user_numbering && !local, permute_rows = permute_cols && H.t_root == 0;
        std::vector<char> tmp;
        void *dst = out;
        if (permute_cols) { tmp.resize(nr * (size_t)ns * es); dst = tmp.data(); }
        if (device_to_dense_host(H, dst)) {
            if (permute_cols) {
                const long long ncol = ns;
#pragma omp parallel for schedule(static)
                for (long long j = 0; j < ncol; j++) {
                    const char *src = tmp.data() + (size_t)j * nr * es;
                    char *col = (char *)out + (size_t)H.sc->perm[H.col_off + j] * nr * es;
                    if (!permute_rows) std::memcpy(col, src, nr * es);
                    else for (size_t i = 0; i < nr; i++) std::memcpy(col + (size_t)H.tc->perm[H.row_off + i] * es, src + i * es, es);
                }
            }
            return;
        }
    }
    if (local) user_numbering = 1;
    // dense(H) = H * I, 64 unit vectors per call (8 sweeps of 8 right-hand sides)
    std::vector<char> e((size_t)ns * BS * es, 0), col(nr * BS * es);
    const double one = 1.0, zero = 0.0;
    for (int j0 = 0; j0 < ns; j0 += BS) {
        const int nb = std::min(BS, ns - j0);
        for (int j = 0; j < nb; j++) { // unit vector in user numbering selecting cluster (or user) column j0+j
            int uj = user_numbering ? j0 + j : H.sc->perm[j0 + j];
            std::memcpy(&e[((size_t)j * ns + uj) * es], &one, sizeof(double));
        }
        device_matmat_host(H, e.data(), nb, col.data());
        for (int j = 0; j < nb; j++) {
            int uj = user_numbering ? j0 + j : H.sc->perm[j0 + j];
            std::memcpy(&e[((size_t)j * ns + uj) * es], &zero, sizeof(double));
            char *dst = (char *)out + (size_t)(j0 + j) * nr * es;
            const char *src = &col[(size_t)j * nr * es];
            if (user_numbering || H.t_root != 0) std::memcpy(dst, src, nr * es);
            else for (size_t i = 0; i < nr; i++) std::memcpy(dst + i * es, src + (size_t)H.tc->perm[i] * es, es);
        }
    }
}

// ---- dense host fallbacks for H-LU / H-Cholesky ------------------------------------
// The reference factorises the H-matrix hierarchically (src/htool/hmatrix/hmatrix.hpp:58-94 ->
// htool::lu_factorization / cholesky_factorization / lu_solve / cholesky_solve).  That is outside the
// accelerated path; so that code written against the reference still runs, small operators are
// densified (GPU products with unit vectors) and factorised on the host with partial pivoting.
struct DenseFactor {
    int kind = 0; // 1 LU, 2 Cholesky
    char uplo = 'L';
    int n = 0;
    std::vector<double> ar; // real storage (column-major, user numbering)
    std::vector<cplx> ac;
    std::vector<int> piv;
};

template <typename T>
static void lu_factor(int n, std::vector<T> &a, std::vector<int> &piv) {
    piv.resize(n);
    for (int k = 0; k < n; k++) {
        int p = k;
        double best = std::abs(a[(size_t)k * n + k]);
        for (int i = k + 1; i < n; i++) { double v = std::abs(a[(size_t)k * n + i]); if (v > best) best = v, p = i; }
        HM_CHECK(best > 0, "lu_factorization: singular matrix");
        piv[k] = p;
        if (p != k) for (int j = 0; j < n; j++) std::swap(a[(size_t)j * n + k], a[(size_t)j * n + p]);
        const T inv = T(1) / a[(size_t)k * n + k];
        for (int i = k + 1; i < n; i++) a[(size_t)k * n + i] *= inv;
#pragma omp parallel for schedule(static)
        for (int j = k + 1; j < n; j++) {
            const T akj = a[(size_t)j * n + k];
            if (akj == T(0)) continue;
            T *cj = &a[(size_t)j * n];
            const T *ck = &a[(size_t)k * n];
            for (int i = k + 1; i < n; i++) cj[i] -= ck[i] * akj;
        }
    }
}
template <typename T>
static void lu_solve_cols(int n, const std::vector<T> &a, const std::vector<int> &piv, char trans, T *b, int mu) {
    for (int c = 0; c < mu; c++) {
        T *x = b + (size_t)c * n;
        if (trans == 'N') {
            for (int k = 0; k < n; k++) if (piv[k] != k) std::swap(x[k], x[piv[k]]);
            for (int k = 0; k < n; k++) for (int i = k + 1; i < n; i++) x[i] -= a[(size_t)k * n + i] * x[k];
            for (int k = n - 1; k >= 0; k--) { x[k] /= a[(size_t)k * n + k]; for (int i = 0; i < k; i++) x[i] -= a[(size_t)k * n + i] * x[k]; }
        } else { // A^T x = b: U^T, then L^T, then the inverse permutation
            for (int k = 0; k < n; k++) { T s = x[k]; for (int i = 0; i < k; i++) s -= a[(size_t)k * n + i] * x[i]; x[k] = s / a[(size_t)k * n + k]; }
            for (int k = n - 1; k >= 0; k--) { T s = x[k]; for (int i = k + 1; i < n; i++) s -= a[(size_t)k * n + i] * x[i]; x[k] = s; }
            for (int k = n - 1; k >= 0; k--) if (piv[k] != k) std::swap(x[k], x[piv[k]]);
        }
    }
}
template <typename T>
static void chol_factor(int n, std::vector<T> &a, char uplo) { // lower factor L stored in the lower triangle (A = L L^H)
    if (uplo == 'U') // use the upper triangle as the data: mirror it down
        for (int j = 0; j < n; j++) for (int i = j + 1; i < n; i++) a[(size_t)j * n + i] = a[(size_t)i * n + j];
    for (int k = 0; k < n; k++) {
        double d = std::real(cplx(a[(size_t)k * n + k]));
        HM_CHECK(d > 0, "cholesky_factorization: matrix is not positive definite");
        const double l = std::sqrt(d);
        a[(size_t)k * n + k] = l;
        for (int i = k + 1; i < n; i++) a[(size_t)k * n + i] /= l;
#pragma omp parallel for schedule(static)
        for (int j = k + 1; j < n; j++) {
            const T ljk = a[(size_t)k * n + j];
            T *cjp = &a[(size_t)j * n];
            const T *ck = &a[(size_t)k * n];
            for (int i = j; i < n; i++) cjp[i] -= ck[i] * ljk; // real symmetric / complex symmetric-as-stored
        }
    }
}
template <typename T>
static void chol_solve_cols(int n, const std::vector<T> &a, T *b, int mu) {
    for (int c = 0; c < mu; c++) {
        T *x = b + (size_t)c * n;
        for (int k = 0; k < n; k++) { x[k] /= a[(size_t)k * n + k]; for (int i = k + 1; i < n; i++) x[i] -= a[(size_t)k * n + i] * x[k]; }
        for (int k = n - 1; k >= 0; k--) { T s = x[k]; for (int i = k + 1; i < n; i++) s -= a[(size_t)k * n + i] * x[i]; x[k] = s / a[(size_t)k * n + k]; }
    }
}

extern "C" {
int htool_hmatrix_to_dense(const htool_hmatrix *h, void *out, int user_numbering) {
    API_BEGIN
    densify(h->H, out, user_numbering);
    API_END
}

// Where a factorisation runs.  Whole-cluster operators of at most 20 000 unknowns: the host fallback below (round 1).  Larger
// ones and partition-built blocks (the block_diagonal_hmatrix of a distributed operator, in its local numbering): a dense copy on
// the device factorised by the dense solver library (dense_device.hip).  HTOOL_DENSE_FACTOR=device / host forces one (tests).
static bool factor_on_device(const HMatrix &H) {
    if (const char *v = getenv("HTOOL_DENSE_FACTOR")) {
        if (std::string(v) == "device") return true;
        if (std::string(v) == "host") return false;
    }
    return H.t_root != 0 || H.s_root != 0 || H.local_numbering || H.tc->n_points > 20000;
}
static DenseFactor *factor_of(htool_hmatrix *h, int kind, char uplo) {
    const HMatrix &H = h->H;
    HM_CHECK(H.t_root == 0 && H.tc->n_points == H.sc->n_points, "factorization needs a square H-matrix built on the whole clusters");
    const int n = H.tc->n_points;
    HM_CHECK(n <= 20000, "factorization: the dense host fallback is limited to 20000 unknowns (hierarchical LU is outside the MI355X hot path)");
    log_message(LOG_WARNING, strprintf("%s: outside the accelerated path -- densifying %d x %d on the GPU and factorising on the host (O(N^3))", kind == 1 ? "lu_factorization" : "cholesky_factorization", n, n));
    std::unique_ptr<DenseFactor> f(new DenseFactor);
    f->kind = kind; f->uplo = uplo; f->n = n;
    if (H.is_complex) { f->ac.resize((size_t)n * n); densify(H, f->ac.data(), 1); if (kind == 1) lu_factor(n, f->ac, f->piv); else chol_factor(n, f->ac, uplo); }
    else { f->ar.resize((size_t)n * n); densify(H, f->ar.data(), 1); if (kind == 1) lu_factor(n, f->ar, f->piv); else chol_factor(n, f->ar, uplo); }
    return f.release();
}
// Round 4: the factorisation is HIERARCHICAL on the device (hlu_device.hip) whenever the operator is one it covers -- real, square on
// one cluster (sub)tree, tolerance >= 1e-7; the dense factorisations above / in dense_device.hip are the fallback for the others
// (complex operators, tighter tolerances) and for HTOOL_FACTOR=dense (or HTOOL_DENSE_FACTOR=host|device, which names a dense
// path).  HTOOL_FACTOR=hlu refuses to fall back.
static void drop_factors(htool_hmatrix *h) {
    delete (DenseFactor *)h->factor;
    h->factor = nullptr;
    device_dense_factor_free(h->dfactor);
    h->dfactor = nullptr;
    device_hlu_free(h->hfactor);
    h->hfactor = nullptr;
}
static void factorise(htool_hmatrix *h, int kind, char uplo, double shift, bool device_only = false) {
    const char *mode_env = getenv("HTOOL_FACTOR");
    const std::string mode = mode_env ? mode_env : "";
    const bool dense_forced = mode == "dense" || (mode != "hlu" && getenv("HTOOL_DENSE_FACTOR") != nullptr);
    if (!dense_forced) {
        try {
            DeviceHLU *f = device_hlu_factor(h->H, kind, shift, 0.0);
            drop_factors(h);
            h->hfactor = f;
            return;
        } catch (const Error &e) {
            if (mode == "hlu") throw;
            log_message(LOG_WARNING, strprintf("%s: the hierarchical factorisation does not cover this operator (%s) -- dense fallback", kind == 1 ? "lu_factorization" : "cholesky_factorization", e.what()));
        }
    }
    if (device_only || factor_on_device(h->H)) {
        log_message(LOG_WARNING, strprintf("%s: dense fallback on the device (a dense copy of the operator is factorised by the dense solver library)",
                                           kind == 1 ? "lu_factorization" : "cholesky_factorization"));
        DeviceDenseFactor *f = device_dense_factor(h->H, kind, uplo, shift);
        drop_factors(h);
        h->dfactor = f;
        return;
    }
    HM_CHECK(shift == 0.0, "factorization with a diagonal shift is implemented on the device paths only");
    DenseFactor *f = factor_of(h, kind, uplo);
    drop_factors(h);
    h->factor = f;
}
int htool_hmatrix_lu_factorization(htool_hmatrix *h) {
    API_BEGIN
    factorise(h, 1, 'N', 0.0);
    API_END
}
int htool_hmatrix_lu_factorization_shifted(htool_hmatrix *h, double shift) {
    API_BEGIN
    factorise(h, 1, 'N', shift, true); // (an extension of the device path: always there, whatever the size)
    API_END
}
int htool_hmatrix_cholesky_factorization(htool_hmatrix *h, char uplo) {
    API_BEGIN
    HM_CHECK(uplo == 'L' || uplo == 'U', "UPLO must be 'L' or 'U'");
    HM_CHECK(!h->H.is_complex, "cholesky_factorization: complex operators are not supported by the dense fallback");
    factorise(h, 2, uplo, 0.0);
    API_END
}
int htool_hmatrix_factor_solve_device(const htool_hmatrix *h, int kind, char trans, void *B_dev, int64_t ldb, int mu, void *stream) {
    API_BEGIN
    if (h->hfactor) {
        HM_CHECK(device_hlu_kind(h->hfactor) == kind, kind == 1 ? "lu_solve: call lu_factorization first" : "cholesky_solve: call cholesky_factorization first");
        device_hlu_solve(h->hfactor, trans, B_dev, (long long)ldb, mu, stream);
        return 0;
    }
    HM_CHECK(h->dfactor != nullptr && device_dense_factor_kind(h->dfactor) == kind, "factor_solve_device: no device factorisation of that kind (call lu_factorization / cholesky_factorization first; "
                                                                                     "operators of at most 20000 unknowns are factorised on the host unless HTOOL_DENSE_FACTOR=device)");
    device_dense_solve(h->dfactor, trans, B_dev, (long long)ldb, mu, stream ? stream : (void *)nullptr);
    API_END
}
int htool_hmatrix_to_dense_device(const htool_hmatrix *h, void *out_dev, int64_t ld, void *stream) {
    API_BEGIN
    HM_CHECK(out_dev != nullptr, "to_dense_device: null output");
    device_to_dense_device(h->H, out_dev, (long long)ld, stream);
    API_END
}
int htool_hmatrix_factor_solve(const htool_hmatrix *h, int kind, char trans, void *B, int mu) {
    API_BEGIN
    if (h->hfactor) {
        HM_CHECK(device_hlu_kind(h->hfactor) == kind, kind == 1 ? "lu_solve: call lu_factorization first" : "cholesky_solve: call cholesky_factorization first");
        HM_CHECK(trans == 'N' || trans == 'T', "factor solve: trans must be 'N' or 'T'");
        device_hlu_solve_host(h->H, h->hfactor, trans, B, mu);
        return 0;
    }
    if (h->dfactor) {
        HM_CHECK(device_dense_factor_kind(h->dfactor) == kind, kind == 1 ? "lu_solve: call lu_factorization first" : "cholesky_solve: call cholesky_factorization first");
        HM_CHECK(trans == 'N' || trans == 'T', "factor solve: trans must be 'N' or 'T'");
        device_dense_solve_host(h->H, h->dfactor, trans, B, mu);
        return 0;
    }
    const DenseFactor *f = (const DenseFactor *)h->factor;
    HM_CHECK(f != nullptr && f->kind == kind, kind == 1 ? "lu_solve: call lu_factorization first" : "cholesky_solve: call cholesky_factorization first");
    HM_CHECK(trans == 'N' || trans == 'T', "factor solve: trans must be 'N' or 'T'");
    if (kind == 1) { if (h->H.is_complex) lu_solve_cols<cplx>(f->n, f->ac, f->piv, trans, (cplx *)B, mu); else lu_solve_cols<double>(f->n, f->ar, f->piv, trans, (double *)B, mu); }
    else chol_solve_cols<double>(f->n, f->ar, (double *)B, mu);
    API_END
}
} // extern "C"

extern "C" {
int htool_block_tree_queues(const htool_cluster *target_root, const htool_cluster *source_root, const htool_build_params *params,
                            int target_partition_number, int64_t *n_admissible, int64_t *n_dense, int *admissible4, int *dense4) {
    API_BEGIN
    HM_CHECK(target_root && source_root && params, "htool_block_tree_queues: null argument");
    ClusterTree *T = CH(target_root)->tree, *S = CH(source_root)->tree;
    BuildParams P;
    P.eta = params->eta;
    P.symmetry = params->symmetry;
    P.uplo = params->uplo;
    P.min_target_depth = params->minimal_target_depth;
    P.min_source_depth = params->minimal_source_depth;
    if (params->store_one_triangle && (params->symmetry == 'S' || params->symmetry == 'H') && (params->uplo == 'L' || params->uplo == 'U') && same_tree(T, S) && target_partition_number < 0)
        P.store_one_triangle = 1;
    int t_root = 0;
    if (target_partition_number >= 0) {
        HM_CHECK(target_partition_number < (int)T->part_nodes.size(), "target_partition_number out of range");
        t_root = T->part_nodes[target_partition_number];
    }
    std::vector<BlockRec> adm, dns;
    build_block_tree(*T, *S, P, t_root, 0, adm, dns);
    if (n_admissible) *n_admissible = (int64_t)adm.size();
    if (n_dense) *n_dense = (int64_t)dns.size();
    auto fill = [](const std::vector<BlockRec> &v, int *out) {
        if (!out) return;
        for (size_t i = 0; i < v.size(); i++) { out[4 * i] = v[i].t_off; out[4 * i + 1] = v[i].m; out[4 * i + 2] = v[i].s_off; out[4 * i + 3] = v[i].n; }
    };
    fill(adm, admissible4);
    fill(dns, dense4);
    API_END
}
int htool_cluster_tiles(const htool_cluster *root, int partition_number, int tile_max, int *out2, int cap) {
    const ClusterTree &T = *CH(root)->tree;
    int node = 0;
    if (partition_number >= 0 && partition_number < (int)T.part_nodes.size()) node = T.part_nodes[partition_number];
    TileSet ts = make_tiles(T, node, tile_max);
    for (int i = 0; i < ts.count() && i < cap && out2; i++) { out2[2 * i] = ts.off[i]; out2[2 * i + 1] = ts.size[i]; }
    return ts.count();
}

} // extern "C"
htool_hmatrix::~htool_hmatrix() {
    delete (DenseFactor *)factor;
    device_dense_factor_free(dfactor);
    device_hlu_free(hfactor);
}
extern "C" int htool_hmatrix_factorization_info(const htool_hmatrix *h, int64_t *out17, double *seconds4) {
    API_BEGIN
    HM_CHECK(h && out17, "htool_hmatrix_factorization_info: null argument");
    for (int i = 0; i < 17; i++) out17[i] = 0;
    if (seconds4) for (int i = 0; i < 4; i++) seconds4[i] = 0;
    if (h->hfactor) { out17[0] = 3; device_hlu_stats(h->hfactor, out17 + 1, seconds4); }
    else if (h->dfactor) out17[0] = 2;
    else if (h->factor) out17[0] = 1;
    API_END
}
extern "C" {
int64_t htool_hmatrix_leaf_count(const htool_hmatrix *h) { return (int64_t)h->H.leaf_count(); }
void htool_hmatrix_leaves(const htool_hmatrix *h, int *out5) {
    for (size_t i = 0; i < h->H.blocks().size(); i++) {
        const BlockRec &b = h->H.blocks()[i];
        int *p = out5 + 5 * i;
        p[0] = b.t_off; p[1] = b.m; p[2] = b.s_off; p[3] = b.n; p[4] = b.rank;
    }
}
int htool_hmatrix_leaf_panels(const htool_hmatrix *h, int64_t leaf, void *A, void *B) {
    API_BEGIN
    device_leaf_panels(h->H, leaf, A, B);
    API_END
}

int htool_hmatrix_leaf_panels_bulk(const htool_hmatrix *h, int64_t n, const int64_t *leaf_ids, int64_t *offsets2, void *out, int64_t *n_elements) {
    API_BEGIN
    int64_t e = device_leaf_panels_bulk(h->H, n, leaf_ids, offsets2, out);
    if (n_elements) *n_elements = e;
    API_END
}

void htool_hmatrix_stats(const htool_hmatrix *h, int64_t *out8) {
    const HMatrix &H = h->H;
    int64_t dense = 0, lr = 0, nd = 0, nl = 0, sr = 0, maxr = 0;
    for (const BlockRec &b : H.blocks()) {
        if (b.rank < 0) { dense += (int64_t)b.m * b.n; nd++; }
        else { lr += (int64_t)b.rank * (b.m + b.n); nl++; sr += b.rank; maxr = std::max<int64_t>(maxr, b.rank); }
    }
    out8[0] = dense; out8[1] = lr; out8[2] = nd; out8[3] = nl; out8[4] = sr;
    out8[5] = device_resident_bytes(H);
    out8[6] = (int64_t)(H.build_seconds * 1e6);
    out8[7] = maxr;
}
double htool_hmatrix_last_product_us(const htool_hmatrix *h) { return device_last_product_us(h->H); }
int htool_hmatrix_phase_times(const htool_hmatrix *h, double *out4) { return device_phase_times(h->H, out4); }
int htool_hmatrix_set_phase_timing(htool_hmatrix *h, int on) {
    API_BEGIN
    device_set_phase_timing(h->H, on != 0);
    API_END
}

int htool_hmatrix_info(const htool_hmatrix *h, int which, char *buf, int cap) {
    const HMatrix &H = h->H;
    std::ostringstream o;
    if (which == 0) {
        o << "Eta=" << H.params.eta << "\nEpsilon=" << H.params.epsilon << "\nTarget_size=" << H.row_size << "\nSource_size=" << H.sc->n_points
          << "\nDimension=" << H.tc->dim << "\nTarget_minclustersize=" << H.tc->max_leaf << "\nSource_minclustersize=" << H.sc->max_leaf
          << "\nSymmetry=" << H.params.symmetry << "\nUPLO=" << H.params.uplo << "\nBackend=HIP gfx950\nTile_size=" << H.tile_max << "\n";
    } else {
        int64_t st[8];
        htool_hmatrix_stats(h, st);
        int64_t dmin = -1, dmax = 0, lmin = -1, lmax = 0, rmin = -1;
        double rmean = 0;
        for (const BlockRec &b : H.blocks()) {
            int64_t sz = (int64_t)b.m * b.n;
            if (b.rank < 0) { dmin = dmin < 0 ? sz : std::min(dmin, sz); dmax = std::max(dmax, sz); }
            else { lmin = lmin < 0 ? sz : std::min(lmin, sz); lmax = std::max(lmax, sz); rmin = rmin < 0 ? b.rank : std::min<int64_t>(rmin, b.rank); rmean += b.rank; }
        }
        if (st[3]) rmean /= (double)st[3];
        double full = (double)H.row_size * (double)H.col_size;
        double cr = full > 0 ? (double)(st[0] + st[1]) / full : 0;
        o << "Number_of_dense_blocks=" << st[2] << "\nNumber_of_low_rank_blocks=" << st[3] << "\nDense_block_size_max=" << dmax << "\nDense_block_size_min=" << std::max<int64_t>(dmin, 0)
          << "\nLow_rank_block_size_max=" << lmax << "\nLow_rank_block_size_min=" << std::max<int64_t>(lmin, 0) << "\nRank_max=" << st[7] << "\nRank_min=" << std::max<int64_t>(rmin, 0)
          << "\nRank_mean=" << rmean << "\nCompression_ratio=" << (cr > 0 ? 1.0 / cr : 0) << "\nSpace_saving=" << 1 - cr << "\nHBM_bytes=" << st[5]
          << "\nBuild_seconds=" << H.build_seconds << "\nNumber_of_batches=" << H.n_batches << "\n";
    }
    std::string s = o.str();
    if (buf && cap > 0) {
        int n = std::min<int>((int)s.size(), cap - 1);
        std::memcpy(buf, s.data(), (size_t)n);
        buf[n] = 0;
    }
    return (int)s.size() + 1;
}

// ---- distributed operator ----------------------------------------------------------------------
int htool_distributed_create_default(const htool_generator *g, const htool_cluster *target_root, const htool_cluster *source_root,
                                     const htool_build_params *params, const htool_comm *comm, htool_distributed **out) {
    API_BEGIN
    HM_CHECK(comm != nullptr, "communicator is null");
    ClusterTree *T = CH(target_root)->tree;
    HM_CHECK((int)T->part_nodes.size() == comm->size, strprintf("target cluster has %zu partitions but the communicator has %d ranks", T->part_nodes.size(), comm->size));
    std::unique_ptr<htool_distributed> d(new htool_distributed);
    d->comm = *comm;
    d->tc = T;
    d->sc = CH(source_root)->tree;
    for (int p = 0; p < comm->size; p++) {
        d->counts.push_back(T->size[T->part_nodes[p]]);
        d->displs.push_back(T->offset[T->part_nodes[p]]);
    }
    if ((int)d->sc->part_nodes.size() == comm->size) // the source tree carries a partition of the same size: the device path gathers x by it
        for (int p = 0; p < comm->size; p++) {
            d->s_counts.push_back(d->sc->size[d->sc->part_nodes[p]]);
            d->s_displs.push_back(d->sc->offset[d->sc->part_nodes[p]]);
        }
    const bool single = comm->size == 1 && T->part_nodes[0] == 0;
    d->hmat = build_hmatrix(g, target_root, source_root, params, single ? -1 : comm->rank);
    // block-diagonal part (distributed_operator/utility.hpp:31): needs the source tree to carry the same partition
    const ClusterTree *S = CH(source_root)->tree;
    // It is built when first asked for (htool_distributed_block_diagonal_hmatrix): a solver without preconditioner, or a
    // bare product loop, never needs this second copy of the diagonal block.  The generator and the clusters have to
    // outlive the distributed operator, as they do in the reference (which stores references to them).
    if (single) d->block_diag = d->hmat;
    else if ((int)S->part_nodes.size() == comm->size) {
        d->gen = g; d->t_root = target_root; d->s_root = source_root; d->params = *params;
        d->block_diag_pending = true;
    }
    *out = d.release();
    API_END
}
void htool_distributed_destroy(htool_distributed *d) {
    if (d) {
        dist_device_free(d->dev);
        if (d->block_diag != d->hmat) delete d->block_diag;
        delete d->hmat;
        delete d;
    }
}
htool_hmatrix *htool_distributed_hmatrix(htool_distributed *d) { return d->hmat; }
htool_hmatrix *htool_distributed_block_diagonal_hmatrix(htool_distributed *d) {
    if (d->block_diag_pending) {
        d->block_diag_pending = false;
        try {
            d->block_diag = build_hmatrix(d->gen, d->t_root, d->s_root, &d->params, d->comm.rank, d->comm.rank);
            d->block_diag->H.local_numbering = true;
        } catch (const std::exception &e) {
            g_err = e.what();
            d->block_diag = nullptr;
        }
    }
    return d->block_diag;
}
void htool_distributed_shape(const htool_distributed *d, int *rows, int *cols) {
    *rows = d->tc->n_points;
    *cols = d->sc->n_points;
}

} // extern "C"
static void distributed_product(const htool_distributed *d, const void *X, int mu, void *Y) {
    const HMatrix &H = d->hmat->H;
    const size_t es = H.is_complex ? 16 : 8;
    const int nt = d->tc->n_points, ns = d->sc->n_points;
    if (H.t_root == 0) { // one rank owning everything: plain user-numbered product
        if (mu == 1) device_matvec_host(H, X, Y);
        else device_matmat_host(H, X, mu, Y);
        return;
    }
    (void)ns;
    std::vector<char> local((size_t)H.row_size * es * mu), full((size_t)nt * es);
    if (mu == 1) device_matvec_host(H, X, local.data()); // local rows, cluster order
    else device_matmat_host(H, X, mu, local.data());
    std::vector<int64_t> cb(d->counts.size()), db(d->displs.size());
    for (size_t p = 0; p < cb.size(); p++) { cb[p] = d->counts[p] * (int64_t)es; db[p] = d->displs[p] * (int64_t)es; }
    HM_CHECK(d->comm.allgatherv != nullptr, "communicator has no allgatherv");
    for (int c = 0; c < mu; c++) {
        int rc = d->comm.allgatherv(d->comm.ctx, local.data() + (size_t)c * H.row_size * es, (int64_t)H.row_size * es, full.data(), cb.data(), db.data());
        HM_CHECK(rc == 0, "allgatherv failed");
        char *y = (char *)Y + (size_t)c * nt * es;
        for (int i = 0; i < nt; i++) std::memcpy(y + (size_t)d->tc->perm[i] * es, &full[(size_t)i * es], es);
    }
}

extern "C" {
int htool_distributed_partition(const htool_distributed *d, int p, int *offset, int *size) {
    API_BEGIN
    HM_CHECK(p >= 0 && p < (int)d->counts.size(), "partition index out of range");
    *offset = (int)d->displs[p];
    *size = (int)d->counts[p];
    API_END
}
int htool_distributed_matvec(const htool_distributed *d, const void *x, void *y) {
    API_BEGIN
    distributed_product(d, x, 1, y);
    API_END
}
int htool_distributed_matmat(const htool_distributed *d, const void *X, int mu, void *Y) {
    API_BEGIN
    distributed_product(d, X, mu, Y);
    API_END
}
}
