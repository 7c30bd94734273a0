// blocktree.cpp -- block cluster tree, produced directly as two flat work queues.
//
// Replaces the recursive block-tree construction inside htool::HMatrixTreeBuilder::build
// (entered from src/htool/hmatrix/hmatrix_tree_builder.hpp:36; parameters :23-43).  lib/htool is
// not vendored; the recursion follows SURVEY.md Appendix A.3:
//   admissible(t,s) := 2 min(r_t, r_s) < eta * max(0, |c_t - c_s| - r_t - r_s)
//   admissible and deep enough -> low-rank queue; both leaves -> dense queue; otherwise split the
//   larger side (both when equal).
// No pointer tree is kept: a leaf is the record (t_off, m, s_off, n, rank) of
// src/htool/matplotlib/hmatrix.hpp:18-22.
//
// Symmetric storage ('S'/'H' with UPLO): by default this engine stores BOTH triangles (the product is
// then a plain sweep over row tiles with no transposed pass).  With BuildParams::store_one_triangle the
// UPLO triangle only is kept, as in the reference, and the product applies every stored off-diagonal
// leaf a second time, transposed (device.hip, "one-triangle storage").  See DESIGN.md "symmetry".
#include <cmath>

#include "hmatrix.hpp"

namespace hm {

namespace {

typedef std::pair<int, int> NodePair; // (target node, source node)

// The traversal emits node pairs only (8 bytes); the full leaf records are materialised afterwards, by all threads.
struct Ctx {
    const ClusterTree &T, &S;
    const BuildParams &P;
    std::vector<NodePair> &adm, &dns;
};

BlockRec make_block(const ClusterTree &T, const ClusterTree &S, int t, int s) {
    BlockRec b;
    b.t_node = t;
    b.s_node = s;
    b.t_off = T.offset[t];
    b.m = T.size[t];
    b.s_off = S.offset[s];
    b.n = S.size[s];
    b.rank = -1;
    b.cap = 0;
    b.batch = -1;
    b.tmp_u = b.tmp_v = 0;
    b.ucol = b.vcol = 0;
    b.tpos = 0;
    b.v_obase = 0;
    b.v_ostride = 0;
    b.status = 0;
    b.z_obase = 0;
    b.z_ostride = 0;
    b.zfin = -1;
    return b;
}

bool admissible(const Ctx &c, int t, int s) {
    double dx = c.T.cx[t] - c.S.cx[s], dy = c.T.cy[t] - c.S.cy[s], dz = c.T.cz[t] - c.S.cz[s];
    double dist = std::sqrt(dx * dx + dy * dy + dz * dz) - c.T.radius[t] - c.S.radius[s];
    return 2 * std::min(c.T.radius[t], c.S.radius[s]) < c.P.eta * std::max(0.0, dist);
}

void visit(const Ctx &c, int t, int s);

void split(const Ctx &c, int t, int s) {
    const bool lt = c.T.is_leaf(t), ls = c.S.is_leaf(s);
    if (lt && ls) {
        c.dns.push_back(NodePair(t, s));
        return;
    }
    if (ls || (!lt && c.T.size[t] > c.S.size[s])) {
        for (int a = 0; a < c.T.n_child[t]; a++) visit(c, c.T.first_child[t] + a, s);
    } else if (lt || c.S.size[s] > c.T.size[t]) {
        for (int b = 0; b < c.S.n_child[s]; b++) visit(c, t, c.S.first_child[s] + b);
    } else {
        for (int a = 0; a < c.T.n_child[t]; a++)
            for (int b = 0; b < c.S.n_child[s]; b++) visit(c, c.T.first_child[t] + a, c.S.first_child[s] + b);
    }
}

void visit(const Ctx &c, int t, int s) {
    if (c.T.size[t] == 0 || c.S.size[s] == 0) return;
    if (c.P.store_one_triangle) { // skip what lies strictly in the triangle that is not stored
        if (c.P.uplo == 'L' && c.S.offset[s] >= c.T.offset[t] + c.T.size[t]) return;
        if (c.P.uplo == 'U' && c.T.offset[t] >= c.S.offset[s] + c.S.size[s]) return;
    }
    if (admissible(c, t, s) && c.T.depth[t] >= c.P.min_target_depth && c.S.depth[s] >= c.P.min_source_depth) {
        c.adm.push_back(NodePair(t, s));
        return;
    }
    split(c, t, s);
}

// node pairs -> leaf records appended to out (all threads)
void materialise(const ClusterTree &T, const ClusterTree &S, const std::vector<NodePair> &ids, std::vector<BlockRec> &out) {
    const size_t base = out.size();
    // (room for the leaves a build appends later -- retried, re-split ones -- without moving 75 MB of records at 1 M points)
    out.reserve(base + ids.size() + ids.size() / 8 + 4096);
    out.resize(base + ids.size());
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)ids.size(); i++) out[base + i] = make_block(T, S, ids[i].first, ids[i].second);
}

} // namespace

// The recursion is expanded sequentially down to a few hundred sub-problems, which are then solved by all threads; the pieces
// are concatenated in the order of the sequential traversal, so the queues are the same for every thread count.
void build_block_tree(const ClusterTree &T, const ClusterTree &S, const BuildParams &P, int t_root, int s_root,
                      std::vector<BlockRec> &adm, std::vector<BlockRec> &dns) {
    struct Item { int kind, t, s; }; // kind 0: admissible pair, 1: dense pair, 2: sub-problem (t, s) still to visit
    std::vector<Item> items;
    items.push_back({2, t_root, s_root < 0 ? 0 : s_root});
    const size_t want = 512;
    for (int pass = 0; pass < 12; pass++) { // one level of the traversal per pass, order preserved
        size_t open = 0;
        for (const Item &it : items) open += it.kind == 2;
        if (open == 0 || open >= want) break;
        std::vector<Item> next;
        next.reserve(items.size() * 2);
        for (const Item &it : items) {
            if (it.kind != 2) { next.push_back(it); continue; }
            // one step of visit(): what it would emit or recurse into, in its order
            std::vector<NodePair> a, d;
            Ctx c{T, S, P, a, d};
            const int t = it.t, s = it.s;
            if (T.size[t] == 0 || S.size[s] == 0) continue;
            if (P.store_one_triangle) {
                if (P.uplo == 'L' && S.offset[s] >= T.offset[t] + T.size[t]) continue;
                if (P.uplo == 'U' && T.offset[t] >= S.offset[s] + S.size[s]) continue;
            }
            if (admissible(c, t, s) && T.depth[t] >= P.min_target_depth && S.depth[s] >= P.min_source_depth) { next.push_back({0, t, s}); continue; }
            const bool lt = T.is_leaf(t), ls = S.is_leaf(s);
            if (lt && ls) { next.push_back({1, t, s}); continue; }
            if (ls || (!lt && T.size[t] > S.size[s])) {
                for (int x = 0; x < T.n_child[t]; x++) next.push_back({2, T.first_child[t] + x, s});
            } else if (lt || S.size[s] > T.size[t]) {
                for (int y = 0; y < S.n_child[s]; y++) next.push_back({2, t, S.first_child[s] + y});
            } else {
                for (int x = 0; x < T.n_child[t]; x++)
                    for (int y = 0; y < S.n_child[s]; y++) next.push_back({2, T.first_child[t] + x, S.first_child[s] + y});
            }
        }
        items.swap(next);
    }
    std::vector<std::vector<NodePair>> pa(items.size()), pd(items.size());
#pragma omp parallel for schedule(dynamic, 1)
    for (long i = 0; i < (long)items.size(); i++) {
        if (items[i].kind != 2) continue;
        Ctx c{T, S, P, pa[i], pd[i]};
        visit(c, items[i].t, items[i].s);
    }
    std::vector<NodePair> ia, id;
    for (size_t i = 0; i < items.size(); i++) {
        if (items[i].kind == 0) ia.push_back(NodePair(items[i].t, items[i].s));
        else if (items[i].kind == 1) id.push_back(NodePair(items[i].t, items[i].s));
        else {
            ia.insert(ia.end(), pa[i].begin(), pa[i].end());
            id.insert(id.end(), pd[i].begin(), pd[i].end());
        }
    }
    materialise(T, S, ia, adm);
    materialise(T, S, id, dns);
}

void split_failed_block(const ClusterTree &T, const ClusterTree &S, const BuildParams &P, const BlockRec &b,
                        std::vector<BlockRec> &adm, std::vector<BlockRec> &dns) {
    std::vector<NodePair> ia, id;
    Ctx c{T, S, P, ia, id};
    split(c, b.t_node, b.s_node);
    for (const NodePair &q : ia) adm.push_back(make_block(T, S, q.first, q.second));
    for (const NodePair &q : id) dns.push_back(make_block(T, S, q.first, q.second));
}

TileSet make_tiles(const ClusterTree &T, int root_node, int tile_max) {
    TileSet ts;
    const int nn = T.node_count();
    ts.node_tile_begin.assign(nn, 0);
    ts.node_tile_end.assign(nn, 0);
    // depth-first walk in offset order; leaves are cut into ceil(size/tile_max) nearly equal pieces
    std::vector<std::pair<int, int>> st; // (node, phase): phase 0 = entering, 1 = leaving
    st.push_back(std::make_pair(root_node, 0));
    while (!st.empty()) {
        int id = st.back().first, phase = st.back().second;
        if (phase == 0) {
            ts.node_tile_begin[id] = ts.count();
            st.back().second = 1;
            if (T.is_leaf(id)) {
                int sz = T.size[id], np = (sz + tile_max - 1) / tile_max;
                int o = T.offset[id];
                for (int p = 0; p < np; p++) {
                    int len = sz / np + (p < sz % np ? 1 : 0);
                    ts.off.push_back(o);
                    ts.size.push_back(len);
                    ts.leaf_of_tile.push_back(id);
                    o += len;
                }
            } else {
                for (int c = T.n_child[id] - 1; c >= 0; c--) st.push_back(std::make_pair(T.first_child[id] + c, 0));
            }
        } else {
            ts.node_tile_end[id] = ts.count();
            st.pop_back();
        }
    }
    return ts;
}

void assign_tile_groups(const ClusterTree &T, int root_node, const TileSet &tiles, int group_positions, std::vector<int> &group) {
    group.assign((size_t)tiles.count(), 0);
    int next = 0;
    std::vector<int> st{root_node};
    while (!st.empty()) { // depth-first in offset order, as make_tiles numbers the tiles
        const int id = st.back();
        st.pop_back();
        if (T.is_leaf(id) || T.size[id] <= group_positions) {
            for (int c = tiles.node_tile_begin[id]; c < tiles.node_tile_end[id]; c++) group[c] = next;
            if (tiles.node_tile_end[id] > tiles.node_tile_begin[id]) next++;
        } else {
            for (int c = T.n_child[id] - 1; c >= 0; c--) st.push_back(T.first_child[id] + c);
        }
    }
}

HMatrix::~HMatrix() {
    if (dev) device_free(dev);
}

} // namespace hm
