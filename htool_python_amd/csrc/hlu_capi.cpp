// hlu_capi.cpp -- C ABI of the hierarchical LU's PLAN (host only; the factorisation itself is entered through
// htool_hmatrix_lu_factorization in capi.cpp and runs in hlu_device.hip).  These entries let a test look at the plan --
// task counts, levels, arena sizes -- and hand its task lists to the CPU checker under oracle/ (which executes them
// with plain loops); nothing in the product calls them.
#include <cstring>
#include <unordered_map>

#include "capi_internal.hpp"
#include "hlu.hpp"

using namespace hm;

htool_hlu_plan::~htool_hlu_plan() { delete plan; }

namespace hm {
namespace hlu {
// (offset, size) -> node of the cluster tree; the deepest node when a chain of single children shares a range
std::vector<LeafIn> leaves_from_rects(const ClusterTree &T, int64_t n, const int32_t *rects5) {
    std::unordered_map<uint64_t, int> node_of;
    node_of.reserve((size_t)T.node_count() * 2);
    for (int v = 0; v < T.node_count(); v++) node_of[((uint64_t)(uint32_t)T.offset[v] << 32) | (uint32_t)T.size[v]] = v;
    std::vector<LeafIn> out((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        const int32_t *r = rects5 + 5 * i;
        auto t = node_of.find(((uint64_t)(uint32_t)r[0] << 32) | (uint32_t)r[1]), s = node_of.find(((uint64_t)(uint32_t)r[2] << 32) | (uint32_t)r[3]);
        HM_CHECK(t != node_of.end() && s != node_of.end(), "hierarchical LU: a leaf is not a pair of cluster nodes");
        out[(size_t)i] = {t->second, s->second, r[4]};
    }
    return out;
}
} // namespace hlu
} // namespace hm

extern "C" {

int htool_hlu_plan_create(const htool_cluster *root, int64_t n_leaves, const int32_t *rects5, double epsilon, int cap_min, int cap_max, double cap_factor,
                          int64_t window_scratch_elems, int64_t window_tasks, int symmetric, int super_rows, int solve_slots, htool_hlu_plan **out) {
    API_BEGIN
    HM_CHECK(root && rects5 && out, "htool_hlu_plan_create: null argument");
    const ClusterHandle *h = reinterpret_cast<const ClusterHandle *>(root);
    hlu::Params P;
    if (epsilon > 0) P.eps = epsilon;
    if (cap_min > 0) P.cap_min = cap_min;
    if (cap_max > 0) P.cap_max = cap_max;
    if (cap_factor > 0) P.cap_factor = cap_factor;
    if (window_scratch_elems > 0) P.window_scratch_elems = window_scratch_elems;
    if (window_tasks > 0) P.window_tasks = window_tasks;
    P.symmetric = symmetric != 0;
    if (super_rows >= 0) P.super_rows = super_rows;
    if (solve_slots >= 0) P.solve_slots = solve_slots != 0;
    std::vector<hlu::LeafIn> in = hlu::leaves_from_rects(*h->tree, n_leaves, rects5);
    htool_hlu_plan *p = new htool_hlu_plan;
    try { p->plan = hlu::make_plan(*h->tree, in, P, h->node); } catch (...) { delete p; throw; }
    *out = p;
    API_END
}

/* out[0..15]: n, leaves, diagonal leaves, factor elems, diag elems, scratch elems, rank slots, windows, tasks of the factorisation,
 * launches (buckets) of it, levels of it, tasks / launches / levels of solve 'N', plan microseconds, tasks of solve 'T';
 * out[16..22]: tasks per kind */
int htool_hlu_plan_info(const htool_hlu_plan *p, int64_t *out, int n_out) {
    API_BEGIN
    HM_CHECK(p && p->plan && out, "htool_hlu_plan_info: null argument");
    const hlu::Plan &P = *p->plan;
    int64_t v[23];
    int64_t tasks = 0, launches = 0, levels = 0;
    for (const hlu::Program &w : P.factor) { tasks += (int64_t)w.tasks.size(); launches += (int64_t)w.buckets.size(); levels += w.n_levels; }
    v[0] = P.n; v[1] = P.n_real_leaves; v[2] = (int64_t)P.diags.size(); v[3] = P.factor_elems; v[4] = P.diag_elems; v[5] = P.scratch_elems;
    v[6] = P.n_slots; v[7] = (int64_t)P.factor.size(); v[8] = tasks; v[9] = launches; v[10] = levels;
    v[11] = (int64_t)P.solve_n.tasks.size(); v[12] = (int64_t)P.solve_n.buckets.size(); v[13] = P.solve_n.n_levels;
    v[14] = (int64_t)(P.plan_seconds * 1e6); v[15] = (int64_t)P.solve_t.tasks.size();
    for (int q = 0; q < 7; q++) v[16 + q] = P.counts[q];
    for (int i = 0; i < n_out && i < 23; i++) out[i] = v[i];
    API_END
}

/* which >= 0: window of the factorisation, -1: solve 'N', -2: solve 'T', -3: the explicit inverse factors of the small diagonal blocks (run once after the windows).  Pointers into the plan (valid until it is freed). */
int htool_hlu_plan_program(const htool_hlu_plan *p, int which, const void **tasks, int64_t *n_tasks, const void **buckets, int64_t *n_buckets,
                           const int64_t **seg, int64_t *n_seg, int64_t *scratch_elems, const int64_t **aux, int64_t *n_aux) {
    API_BEGIN
    HM_CHECK(p && p->plan, "htool_hlu_plan_program: null argument");
    const hlu::Plan &P = *p->plan;
    HM_CHECK(which >= -3 && which < (int)P.factor.size(), "htool_hlu_plan_program: no such program");
    const hlu::Program &G = which == -1 ? P.solve_n : which == -2 ? P.solve_t : which == -3 ? P.invert : P.factor[(size_t)which];
    if (tasks) *tasks = G.tasks.data();
    if (n_tasks) *n_tasks = (int64_t)G.tasks.size();
    if (buckets) *buckets = G.buckets.data();
    if (n_buckets) *n_buckets = (int64_t)G.buckets.size();
    if (seg) *seg = G.seg.data();
    if (n_seg) *n_seg = (int64_t)G.seg.size();
    if (scratch_elems) *scratch_elems = G.scratch_elems;
    if (aux) *aux = G.aux.data();
    if (n_aux) *n_aux = (int64_t)G.aux.size();
    API_END
}

/* leaves: one 48-byte record per rank slot (the operator's leaves first: htool_hlu_plan_info's out[1] of them) */
int htool_hlu_plan_tables(const htool_hlu_plan *p, const void **leaves, const void **diags) {
    API_BEGIN
    HM_CHECK(p && p->plan, "htool_hlu_plan_tables: null argument");
    if (leaves) *leaves = p->plan->leaves.data();
    if (diags) *diags = p->plan->diags.data();
    API_END
}

void htool_hlu_plan_free(htool_hlu_plan *p) { delete p; }

} // extern "C"
