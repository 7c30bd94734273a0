// hlu_symbolic.cpp -- the plan of a hierarchical LU: block tree from the leaves, the block-recursive algorithm run on the
// structure alone, leaf-level tasks with dependency levels (see hlu.hpp).  Host code, no numbers are touched here.
//
// Reference surface: htool::lu_factorization / lu_solve (bound at src/htool/hmatrix/hmatrix.hpp:58-78), cholesky_* (:61-94).
#include "hlu.hpp"

#include <algorithm>
#include <cmath>
#include <memory>
#include <deque>
#include <unordered_map>

#include "common.hpp"

namespace hm {
namespace hlu {

namespace {

// Dependency state of one buffer whose rows are grouped by cluster-leaf cell (or of a dense leaf: one cell).
struct Track {
    int c0 = 0, nc = 1;
    int base_w = 0, base_r = 0, acc = 0, maxw = 0, maxr = 0;
    std::vector<int> w, r; // per cell, created by the first per-cell access
    void init(int c0_, int nc_) { c0 = c0_; nc = nc_; }
    int dep_read(int a, int b) const {
        int L = std::max(base_w, acc);
        if (!w.empty()) for (int c = a; c < b; c++) L = std::max(L, w[c - c0]);
        return L;
    }
    void note_read(int a, int b, int lev) {
        if (r.empty()) r.assign(nc, 0);
        for (int c = a; c < b; c++) r[c - c0] = std::max(r[c - c0], lev);
        maxr = std::max(maxr, lev);
    }
    int dep_write(int a, int b) const {
        int L = std::max(std::max(base_w, base_r), acc);
        if (!w.empty()) for (int c = a; c < b; c++) L = std::max(L, w[c - c0]);
        if (!r.empty()) for (int c = a; c < b; c++) L = std::max(L, r[c - c0]);
        return L;
    }
    void note_write(int a, int b, int lev) {
        if (w.empty()) w.assign(nc, 0);
        for (int c = a; c < b; c++) w[c - c0] = lev;
        maxw = std::max(maxw, lev);
    }
    int dep_read_all() const { return std::max(std::max(base_w, acc), maxw); }
    void note_read_all(int lev) { base_r = std::max(base_r, lev); maxr = std::max(maxr, lev); }
    int dep_write_all() const { return std::max(std::max(std::max(base_w, base_r), acc), std::max(maxw, maxr)); }
    void note_write_all(int lev) { base_w = lev; maxw = std::max(maxw, lev); }
    int dep_acc() const { return std::max(std::max(base_w, base_r), std::max(maxw, maxr)); }
    void note_acc(int lev) { acc = std::max(acc, lev); }
};

struct BNode {
    int t, s;
    int kids;  // first entry in the kid table (nct x ncs entries, -1: empty block), -1 for a leaf
    int split; // bit 0: the target node is split, bit 1: the source node
    int leaf;  // leaf id, or -1
};

// a thin block of columns whose rows are cluster positions [pos0, ...)
struct Thin {
    int64_t base = 0;
    int ld = 0, pos0 = 0, space = SP_FACTOR;
    Track *tr = nullptr; // per-cell dependency state, or
    int opleaf = -1;     // a finished leaf read as a whole
    bool transposed = false; // element (row, col) at base + row * ld + col (a dense leaf seen as its transpose)
};

struct Emitter {
    const ClusterTree &T;
    Plan &P;
    std::vector<BNode> bn;
    std::vector<int> kid;
    std::vector<int> diag_bnode, cell0, cell1;
    std::vector<Track> trU, trV, trD;
    std::vector<char> dirty;
    std::deque<Track> temp_tracks; // scratch buffers of the group being emitted
    std::vector<Task> cur;
    int window_base = 1, max_level = 0;
    int64_t scratch_used = 0;
    int64_t next_slot = 0;
    bool planning_factor = true;

    Emitter(const ClusterTree &T_, Plan &P_) : T(T_), P(P_) {}

    int end(int x) const { return T.offset[x] + T.size[x]; }
    bool overlap(int x, int y) const { return T.offset[x] < end(y) && T.offset[y] < end(x); }
    int smaller(int x, int y) const { return T.size[x] <= T.size[y] ? x : y; }
    int child_containing(int parent, int x) const {
        for (int a = 0; a < T.n_child[parent]; a++) {
            const int c = T.first_child[parent] + a;
            if (T.offset[x] >= T.offset[c] && end(x) <= end(c) && T.size[c] > 0) return a;
        }
        throw Error("hierarchical LU: inconsistent cluster tree");
    }

    // ---- block tree from the leaves: the split rule of blocktree.cpp, a leaf wherever the operator has one ----
    int build_bnode(int t, int s, const std::unordered_map<uint64_t, int> &leaf_of) {
        const int id = (int)bn.size();
        bn.push_back({t, s, -1, 0, -1});
        if (t == s) diag_bnode[t] = id;
        auto it = leaf_of.find(((uint64_t)(uint32_t)t << 32) | (uint32_t)s);
        if (it != leaf_of.end()) { bn[id].leaf = it->second; return id; }
        const bool lt = T.is_leaf(t), ls = T.is_leaf(s);
        HM_CHECK(!(lt && ls), "hierarchical LU: the leaves do not tile the operator (a pair of cluster leaves without a leaf)");
        int split;
        if (ls || (!lt && T.size[t] > T.size[s])) split = 1;
        else if (lt || T.size[s] > T.size[t]) split = 2;
        else split = 3;
        const int nct = (split & 1) ? T.n_child[t] : 1, ncs = (split & 2) ? T.n_child[s] : 1;
        const int k0 = (int)kid.size();
        kid.resize(kid.size() + (size_t)nct * ncs, -1);
        bn[id].kids = k0;
        bn[id].split = split;
        for (int a = 0; a < nct; a++)
            for (int b = 0; b < ncs; b++) {
                const int ct = (split & 1) ? T.first_child[t] + a : t, cs = (split & 2) ? T.first_child[s] + b : s;
                if (T.size[ct] == 0 || T.size[cs] == 0) continue;
                if (P.params.symmetric && end(ct) <= T.offset[cs]) continue; // (strictly above the diagonal: not part of the factorisation)
                const int c = build_bnode(ct, cs, leaf_of);
                kid[k0 + a * ncs + b] = c;
            }
        return id;
    }
    int ncs_of(const BNode &B) const { return (B.split & 2) ? T.n_child[B.s] : 1; }
    int nct_of(const BNode &B) const { return (B.split & 1) ? T.n_child[B.t] : 1; }

    // deepest block node that covers (t, s), starting from a node that does
    int descend(int b, int t, int s) const {
        for (;;) {
            const BNode &B = bn[b];
            if (B.leaf >= 0) return b;
            int ti = 0, si = 0;
            if (B.split & 1) { if (t == B.t) return b; ti = child_containing(B.t, t); }
            if (B.split & 2) { if (s == B.s) return b; si = child_containing(B.s, s); }
            const int c = kid[B.kids + ti * ncs_of(B) + si];
            HM_CHECK(c >= 0, "hierarchical LU: empty block");
            b = c;
        }
    }
    int descend_soft(int b, int t, int s) const { // the same for a TARGET of the symmetric factorisation: blocks above the diagonal do not exist
        for (;;) {
            const BNode &B = bn[b];
            if (B.leaf >= 0) return b;
            int ti = 0, si = 0;
            if (B.split & 1) { if (t == B.t) return b; ti = child_containing(B.t, t); }
            if (B.split & 2) { if (s == B.s) return b; si = child_containing(B.s, s); }
            const int c = kid[B.kids + ti * ncs_of(B) + si];
            if (c < 0) return b;
            b = c;
        }
    }
    template <typename F>
    void for_leaves(int b, int t, int s, F &&f) const { // leaves below b that meet (t, s), with the intersection
        const BNode &B = bn[b];
        if (B.leaf >= 0) { f(B.leaf, smaller(t, B.t), smaller(s, B.s)); return; }
        const int n = nct_of(B) * ncs_of(B);
        for (int q = 0; q < n; q++) {
            const int c = kid[B.kids + q];
            if (c >= 0 && overlap(bn[c].t, t) && overlap(bn[c].s, s)) for_leaves(c, t, s, f);
        }
    }

    // ---- tasks ----
    int emit(Task &tk, int dep) {
        tk.level = std::max(dep + 1, window_base);
        max_level = std::max(max_level, tk.level);
        cur.push_back(tk);
        P.counts[tk.type]++;
        return tk.level;
    }
    static Task blank(int type) {
        Task t;
        t.type = type; t.flags = 0; t.level = 0; t.leaf = -1; t.kref = -1; t.kconst = 0; t.m = t.n = t.r0 = t.c0 = 0;
        t.a_ld = t.b_ld = t.x_ld = t.y_ld = 0; t.a = t.b = t.x = t.y = t.w = 0;
        return t;
    }
    int64_t ref(const Thin &X, int pos) const { return make_ref(X.space, X.base + (X.transposed ? (int64_t)(pos - X.pos0) * X.ld : (int64_t)(pos - X.pos0))); }
    int dep_leaf_read(int l) const { return std::max(trU[l].dep_read_all(), trV[l].dep_read_all()); }
    void note_leaf_read(int l, int lev) { trU[l].note_read_all(lev); trV[l].note_read_all(lev); }
    int dep_thin_read(const Thin &X, int node) const { return X.opleaf >= 0 ? dep_leaf_read(X.opleaf) : X.tr->dep_read(cell0[node], cell1[node]); }
    void note_thin_read(const Thin &X, int node, int lev) { if (X.opleaf >= 0) note_leaf_read(X.opleaf, lev); else X.tr->note_read(cell0[node], cell1[node], lev); }

    void ensure_final(int l) {
        if (!dirty[l]) return;
        Task t = blank(T_FINAL);
        t.leaf = l;
        const int lev = emit(t, std::max(trU[l].dep_write_all(), trV[l].dep_write_all()));
        trU[l].note_write_all(lev);
        trV[l].note_write_all(lev);
        dirty[l] = 0;
    }
    Thin leaf_u(int l) { const Leaf &L = P.leaves[l]; Thin X; X.base = L.u; X.ld = L.m; X.pos0 = L.t_off; X.space = SP_FACTOR; X.opleaf = l; return X; }
    Thin leaf_v(int l) { const Leaf &L = P.leaves[l]; Thin X; X.base = L.v; X.ld = L.n; X.pos0 = L.s_off; X.space = SP_FACTOR; X.opleaf = l; return X; }
    void begin_group() { maybe_close_window(); temp_tracks.clear(); } // the scratch blocks of a group are referenced by its own tasks only
    Thin new_scratch(int node, int cols) {
        Thin X;
        X.base = scratch_used;
        X.ld = T.size[node];
        X.pos0 = T.offset[node];
        X.space = SP_SCRATCH;
        scratch_used += ((int64_t)T.size[node] * cols + 1) & ~(int64_t)1;
        temp_tracks.emplace_back();
        temp_tracks.back().init(cell0[node], cell1[node] - cell0[node]);
        X.tr = &temp_tracks.back();
        return X;
    }
    void fill_zero(const Thin &Y, int node, int kref, int kconst) {
        Task t = blank(T_FILL);
        t.m = T.size[node];
        t.kref = kref; t.kconst = kconst;
        t.y = ref(Y, T.offset[node]);
        t.y_ld = Y.ld;
        // (the number of columns is the rank of a leaf: it must be the one the consumers of the block will see)
        int dep = Y.tr->dep_write(cell0[node], cell1[node]);
        if (kref >= 0 && kref < (int)P.n_real_leaves) dep = std::max(dep, dep_leaf_read(kref));
        const int lev = emit(t, dep);
        if (kref >= 0 && kref < (int)P.n_real_leaves) note_leaf_read(kref, lev);
        Y.tr->note_write(cell0[node], cell1[node], lev);
    }
    // Y[rows of the output side] (+)= -+ op(leaf restricted to (it, is)) X[rows of the input side]
    void apply_leaf(int l, int it, int is, bool trans, const Thin &X, const Thin &Y, int kref, int kconst, bool accum, bool sub) {
        const Leaf &L = P.leaves[l];
        if (L.kind == 1) ensure_final(l);
        const int in = trans ? it : is, out = trans ? is : it;
        Task t = blank(L.kind == 1 ? T_APPLY_LR : T_APPLY_DENSE);
        t.leaf = l;
        t.flags = (trans ? F_TRANS : 0) | (accum ? F_ACCUM : 0) | (sub ? F_SUB : 0);
        t.kref = kref; t.kconst = kconst;
        t.m = T.size[out];
        t.n = T.size[in];
        if (L.kind == 1) {
            const int64_t urow = L.u + (T.offset[it] - L.t_off), vrow = L.v + (T.offset[is] - L.s_off);
            t.a = make_ref(SP_FACTOR, trans ? vrow : urow); t.a_ld = trans ? L.n : L.m;
            t.b = make_ref(SP_FACTOR, trans ? urow : vrow); t.b_ld = trans ? L.m : L.n;
        } else {
            HM_CHECK(it == bn_t(l) && is == bn_s(l), "hierarchical LU: a dense leaf cannot be restricted");
            t.a = make_ref(SP_FACTOR, L.u); t.a_ld = L.m;
        }
        t.x = ref(X, T.offset[in]); t.x_ld = X.ld;
        t.y = ref(Y, T.offset[out]); t.y_ld = Y.ld;
        int dep = std::max(dep_leaf_read(l), dep_thin_read(X, in));
        dep = std::max(dep, Y.tr->dep_write(cell0[out], cell1[out]));
        const int lev = emit(t, dep);
        note_leaf_read(l, lev);
        note_thin_read(X, in, lev);
        Y.tr->note_write(cell0[out], cell1[out], lev);
    }
    int bn_t(int l) const { return leaf_t[l]; }
    int bn_s(int l) const { return leaf_s[l]; }
    std::vector<int> leaf_t, leaf_s;

    // X[rows of t] <- op(inverse factor of the diagonal leaf of t) X[rows of t]
    void apply_diag(int t, int which, bool trans, const Thin &X, int kref, int kconst) {
        const int l = bn[diag_bnode[t]].leaf;
        const Diag &D = P.diags[P.leaves[l].diag];
        Task tk = blank(T_APPLY_DENSE);
        tk.leaf = l;
        tk.flags = F_INPLACE | (trans ? F_TRANS : 0);
        tk.kref = kref; tk.kconst = kconst;
        tk.m = tk.n = D.m;
        tk.a = make_ref(SP_DIAG, which ? D.uinv : D.linv); tk.a_ld = D.m;
        tk.x = tk.y = ref(X, T.offset[t]); tk.x_ld = tk.y_ld = X.ld;
        const int dep = std::max(trD[P.leaves[l].diag].dep_read_all(), X.tr->dep_write(cell0[t], cell1[t]));
        const int lev = emit(tk, dep);
        trD[P.leaves[l].diag].note_read_all(lev);
        X.tr->note_write(cell0[t], cell1[t], lev);
    }
    // target leaf lc, sub-block (it, is):  -= X[rows of it] Z[rows of is]^T
    void add_lr(int lc, int it, int is, const Thin &X, const Thin &Z, int kref, int kconst) {
        const Leaf &L = P.leaves[lc];
        Task t = blank(T_ADDLR);
        t.leaf = lc;
        t.flags = F_SUB | (X.transposed ? F_XT : 0) | (Z.transposed ? F_YT : 0);
        t.kref = kref; t.kconst = kconst;
        t.m = T.size[it]; t.n = T.size[is];
        t.r0 = T.offset[it] - L.t_off; t.c0 = T.offset[is] - L.s_off;
        t.x = ref(X, T.offset[it]); t.x_ld = X.ld;
        t.y = ref(Z, T.offset[is]); t.y_ld = Z.ld;
        int dep = std::max(dep_thin_read(X, it), dep_thin_read(Z, is));
        dep = std::max(dep, L.kind == 1 ? std::max(trU[lc].dep_acc(), trV[lc].dep_acc()) : trU[lc].dep_acc());
        const int lev = emit(t, dep);
        note_thin_read(X, it, lev);
        note_thin_read(Z, is, lev);
        trU[lc].note_acc(lev);
        if (L.kind == 1) { trV[lc].note_acc(lev); dirty[lc] = 1; }
    }

    void maybe_close_window() {
        if (scratch_used > P.params.window_scratch_elems || (int64_t)cur.size() > P.params.window_tasks) close_window();
    }
    void close_window() {
        if (cur.empty()) return;
        P.factor.emplace_back();
        P.factor.back().scratch_elems = scratch_used; // (the stage blocks of split runs are placed behind the window's own scratch)
        finish_program(cur, P.factor.back());
        P.scratch_elems = std::max(P.scratch_elems, P.factor.back().scratch_elems);
        scratch_used = 0;
        window_base = max_level + 1;
        temp_tracks.clear();
    }
    void finish_program(std::vector<Task> &tasks, Program &out) {
        // stable counting sort by (level, type); runs of one target inside the ADDLR / FINAL buckets
        int lo = INT32_MAX, hi = 0;
        for (const Task &t : tasks) { lo = std::min(lo, t.level); hi = std::max(hi, t.level); }
        if (tasks.empty()) { lo = 1; hi = 0; }
        const int64_t nkeys = (int64_t)(hi - lo + 1) * T_NTYPES;
        std::vector<int64_t> start((size_t)nkeys + 1, 0);
        for (const Task &t : tasks) start[(size_t)(t.level - lo) * T_NTYPES + t.type + 1]++;
        for (int64_t k = 0; k < nkeys; k++) start[k + 1] += start[k];
        out.tasks.resize(tasks.size());
        {
            std::vector<int64_t> pos(start.begin(), start.end() - 1);
            for (const Task &t : tasks) out.tasks[pos[(size_t)(t.level - lo) * T_NTYPES + t.type]++] = t;
        }
        tasks.clear();
        tasks.shrink_to_fit();
        for (int64_t k = 0; k < nkeys; k++) {
            if (start[k + 1] == start[k]) continue;
            Bucket b;
            b.type = (int)(k % T_NTYPES); b.level = lo + (int)(k / T_NTYPES);
            b.begin = start[k]; b.end = start[k + 1];
            b.seg_begin = b.seg_end = 0;
            if (b.type == T_ADDLR || b.type == T_FINAL) {
                std::stable_sort(out.tasks.begin() + b.begin, out.tasks.begin() + b.end, [](const Task &x, const Task &y) { return x.leaf < y.leaf; });
                b.seg_begin = (int64_t)out.seg.size();
                for (int64_t i = b.begin; i < b.end; i++)
                    if (i == b.begin || out.tasks[i].leaf != out.tasks[i - 1].leaf) out.seg.push_back(i);
                b.seg_end = (int64_t)out.seg.size();
                out.seg.push_back(b.end); // end marker of the last run
            }
            out.buckets.push_back(b);
        }
        out.n_levels = out.tasks.empty() ? 0 : hi - lo + 1;
        if (planning_factor) split_long_runs(out);
    }
    // A run of many updates of ONE low-rank leaf inside one launch is a serial chain of truncations in one workgroup while the rest
    // of the GPU waits for it.  Such a run is dealt out to several workgroups: part p accumulates its updates into a stage block of
    // its own (an empty low-rank block of the leaf's shape in the scratch space, truncated at the end of the part), and a merge launch
    // right behind adds the stage blocks to the leaf.  The order of the additions is fixed by the plan, so the result stays
    // reproducible (it differs from the unsplit one by truncation errors of the size of the tolerance).
    void split_long_runs(Program &G) {
        const Params &prm = P.params;
        std::vector<Task> tasks;
        std::vector<Bucket> buckets;
        std::vector<int64_t> seg;
        tasks.reserve(G.tasks.size() + G.tasks.size() / 8);
        int64_t stage_elems = 0;
        for (const Bucket &b0 : G.buckets) {
            Bucket b = b0;
            b.begin = (int64_t)tasks.size();
            if (b0.type != T_ADDLR) {
                tasks.insert(tasks.end(), G.tasks.begin() + b0.begin, G.tasks.begin() + b0.end);
                b.end = (int64_t)tasks.size();
                if (b0.type == T_FINAL) {
                    b.seg_begin = (int64_t)seg.size();
                    for (int64_t q = b0.seg_begin; q < b0.seg_end; q++) seg.push_back(G.seg[(size_t)q] - b0.begin + b.begin);
                    b.seg_end = (int64_t)seg.size();
                    seg.push_back(b.end);
                }
                buckets.push_back(b);
                continue;
            }
            std::vector<Task> merges;
            b.seg_begin = (int64_t)seg.size();
            for (int64_t q = b0.seg_begin; q < b0.seg_end; q++) {
                const int64_t r0 = G.seg[(size_t)q], r1 = G.seg[(size_t)q + 1], cnt = r1 - r0;
                const int target = G.tasks[(size_t)r0].leaf;
                const Leaf L = P.leaves[(size_t)target];
                const int parts = (int)std::min<int64_t>(prm.split_max_parts, cnt / std::max(prm.split_part, 1));
                const int64_t need = ((((int64_t)L.m * 64 + 1) & ~(int64_t)1) + (((int64_t)L.n * 64 + 1) & ~(int64_t)1)) * parts;
                if (L.kind != 1 || cnt <= prm.split_min || parts < 2 || G.scratch_elems + stage_elems + need > 2 * prm.window_scratch_elems) {
                    seg.push_back((int64_t)tasks.size());
                    tasks.insert(tasks.end(), G.tasks.begin() + r0, G.tasks.begin() + r1);
                    continue;
                }
                for (int pt = 0; pt < parts; pt++) {
                    const int64_t a = r0 + cnt * pt / parts, e = r0 + cnt * (pt + 1) / parts;
                    Leaf S = L; // the stage block
                    S.cap = 64; S.diag = -1; S.rank0 = 0;
                    S.u = make_ref(SP_SCRATCH, G.scratch_elems + stage_elems);
                    stage_elems += ((int64_t)L.m * 64 + 1) & ~(int64_t)1;
                    S.v = make_ref(SP_SCRATCH, G.scratch_elems + stage_elems);
                    stage_elems += ((int64_t)L.n * 64 + 1) & ~(int64_t)1;
                    const int sid = (int)next_slot++;
                    if ((size_t)sid >= P.leaves.size()) P.leaves.resize((size_t)sid + 1, Leaf{0, 0, 0, 0, 0, 0, 0, 0, -1, 0});
                    P.leaves[(size_t)sid] = S;
                    seg.push_back((int64_t)tasks.size());
                    for (int64_t i = a; i < e; i++) { Task t = G.tasks[(size_t)i]; t.leaf = sid; tasks.push_back(t); }
                    Task fin = blank(T_FINAL);
                    fin.leaf = sid; fin.level = b0.level;
                    tasks.push_back(fin);
                    Task mg = blank(T_ADDLR);
                    mg.leaf = target; mg.level = b0.level; mg.kref = sid;
                    mg.m = L.m; mg.n = L.n;
                    mg.x = S.u; mg.x_ld = L.m; mg.y = S.v; mg.y_ld = L.n;
                    merges.push_back(mg);
                    P.counts[T_FINAL]++; P.counts[T_ADDLR]++;
                }
            }
            b.end = (int64_t)tasks.size();
            b.seg_end = (int64_t)seg.size();
            seg.push_back(b.end);
            buckets.push_back(b);
            if (!merges.empty()) { // (sorted by target already: the runs were)
                Bucket mb = b0;
                mb.begin = (int64_t)tasks.size();
                mb.seg_begin = (int64_t)seg.size();
                for (size_t i = 0; i < merges.size(); i++) {
                    if (i == 0 || merges[i].leaf != merges[i - 1].leaf) seg.push_back((int64_t)tasks.size());
                    tasks.push_back(merges[i]);
                }
                mb.end = (int64_t)tasks.size();
                mb.seg_end = (int64_t)seg.size();
                seg.push_back(mb.end);
                buckets.push_back(mb);
            }
        }
        G.tasks.swap(tasks);
        G.buckets.swap(buckets);
        G.seg.swap(seg);
        G.scratch_elems += stage_elems;
    }

    // ---- the products C(t, s) -= A(t, r) B(r, s) ----
    void mm(int t, int r, int s, int a, int b, int c) {
        if (T.size[t] == 0 || T.size[r] == 0 || T.size[s] == 0) return;
        a = descend(a, t, r);
        b = descend(b, r, s);
        c = descend(c, t, s);
        const BNode &A = bn[a], &B = bn[b];
        const bool a_lr = A.leaf >= 0 && P.leaves[A.leaf].kind == 1, b_lr = B.leaf >= 0 && P.leaves[B.leaf].kind == 1;
        if (a_lr) { // (U_a V_a^T) B = U_a (B^T V_a)^T
            begin_group();
            const int la = A.leaf;
            ensure_final(la);
            Thin Z = new_scratch(s, P.leaves[la].cap);
            fill_zero(Z, s, la, 0);
            const Thin Va = leaf_v(la), Ua = leaf_u(la);
            for_leaves(b, r, s, [&](int l, int it, int is) { apply_leaf(l, it, is, true, Va, Z, la, 0, true, false); });
            for_leaves(c, t, s, [&](int l, int it, int is) { add_lr(l, it, is, Ua, Z, la, 0); });
            return;
        }
        if (b_lr) { // A (U_b V_b^T) = (A U_b) V_b^T
            begin_group();
            const int lb = B.leaf;
            ensure_final(lb);
            Thin X = new_scratch(t, P.leaves[lb].cap);
            fill_zero(X, t, lb, 0);
            const Thin Ub = leaf_u(lb), Vb = leaf_v(lb);
            for_leaves(a, t, r, [&](int l, int it, int is) { apply_leaf(l, it, is, false, Ub, X, lb, 0, true, false); });
            for_leaves(c, t, s, [&](int l, int it, int is) { add_lr(l, it, is, X, Vb, lb, 0); });
            return;
        }
        if (A.leaf >= 0 && B.leaf >= 0) { // two dense leaves: t, r, s are cluster leaves
            begin_group();
            const Leaf &LA = P.leaves[A.leaf], &LB = P.leaves[B.leaf];
            const BNode &C = bn[c];
            HM_CHECK(C.leaf >= 0, "hierarchical LU: the target of a product of dense leaves is not a leaf");
            Thin Da = leaf_u(A.leaf);
            if (P.leaves[C.leaf].kind == 0) {
                Thin Dbt = leaf_u(B.leaf); // D_b^T: rows = columns of D_b
                Dbt.transposed = true; Dbt.ld = LB.m; Dbt.pos0 = LB.s_off;
                add_lr(C.leaf, t, s, Da, Dbt, -1, LA.n);
                return;
            }
            const int kp = std::min(LA.m, LB.n);
            Thin X = new_scratch(t, kp), Z = new_scratch(s, kp);
            const int64_t w = scratch_used;
            scratch_used += ((int64_t)LA.m * LB.n + 1) & ~(int64_t)1;
            Task tk = blank(T_DDPROD);
            tk.leaf = C.leaf;
            tk.kref = (int)next_slot++;
            tk.kconst = kp;
            tk.m = LA.m; tk.n = LB.n; tk.r0 = LA.n;
            tk.a = make_ref(SP_FACTOR, LA.u); tk.a_ld = LA.m;
            tk.b = make_ref(SP_FACTOR, LB.u); tk.b_ld = LB.m;
            tk.x = ref(X, T.offset[t]); tk.x_ld = X.ld;
            tk.y = ref(Z, T.offset[s]); tk.y_ld = Z.ld;
            tk.w = make_ref(SP_SCRATCH, w);
            const int lev = emit(tk, std::max(dep_leaf_read(A.leaf), dep_leaf_read(B.leaf)));
            note_leaf_read(A.leaf, lev);
            note_leaf_read(B.leaf, lev);
            X.tr->note_write(cell0[t], cell1[t], lev);
            Z.tr->note_write(cell0[s], cell1[s], lev);
            add_lr(C.leaf, t, s, X, Z, tk.kref, 0);
            return;
        }
        // a covering node that cannot descend any more is split exactly where (t, r) / (r, s) touches its own clusters
        const bool st = A.leaf < 0 && (A.split & 1) && t == A.t;
        const bool sr = (A.leaf < 0 && (A.split & 2) && r == A.s) || (B.leaf < 0 && (B.split & 1) && r == B.t);
        const bool ss = B.leaf < 0 && (B.split & 2) && s == B.s;
        HM_CHECK(st || sr || ss, "hierarchical LU: the product recursion cannot descend");
        HM_CHECK((!st || T.n_child[t] > 0) && (!sr || T.n_child[r] > 0) && (!ss || T.n_child[s] > 0), "hierarchical LU: split of a cluster leaf");
        const int nt = st ? T.n_child[t] : 1, nr = sr ? T.n_child[r] : 1, ns = ss ? T.n_child[s] : 1;
        for (int i = 0; i < nt; i++)
            for (int k = 0; k < nr; k++)
                for (int j = 0; j < ns; j++)
                    mm(st ? T.first_child[t] + i : t, sr ? T.first_child[r] + k : r, ss ? T.first_child[s] + j : s, a, b, c);
    }
    int block_of(int t, int s) const { // block node of two children of one cluster node
        const BNode &D = bn[diag_bnode[T.parent[t]]];
        const int c = kid[D.kids + (t - T.first_child[T.parent[t]]) * T.n_child[T.parent[t]] + (s - T.first_child[T.parent[t]])];
        HM_CHECK(c >= 0, "hierarchical LU: empty off-diagonal block");
        return c;
    }

    // ---- diagonal blocks with explicit inverse factors (hlu.hpp: Super) ----
    std::vector<int> super_of;        // per cluster node: its record, or -1
    std::vector<Track> trS;
    bool use_super = false;           // (solve programs only: the sweep inside such a block is one dense product)
    void choose_supers(int t, int64_t &de) {
        if (T.size[t] == 0) return;
        const int d = diag_bnode[t];
        if (d < 0 || bn[d].leaf >= 0) return; // (a cluster leaf has its inverse factors already)
        if (T.size[t] <= P.params.super_rows) {
            Super S;
            S.node = t; S.m = T.size[t];
            S.linv = de; de += (int64_t)S.m * S.m;
            S.uinv = S.linv;
            if (!P.params.symmetric) { S.uinv = de; de += (int64_t)S.m * S.m; }
            super_of[t] = (int)P.supers.size();
            P.supers.push_back(S);
            return;
        }
        for (int a = 0; a < T.n_child[t]; a++) choose_supers(T.first_child[t] + a, de);
    }
    void apply_super(int t, int which, bool trans, const Thin &X, int kref, int kconst) {
        const Super &S = P.supers[(size_t)super_of[t]];
        if (X.space == SP_RHS && P.params.solve_slots && aux_out != nullptr) {
            // a dense product of ~1000 x 1000 by ONE workgroup is 0.4 ms of streaming: the rows are dealt out 64 at a time into a private slot
            // (one launch, all pieces side by side), then copied back by a REDUCE task that assigns instead of subtracting
            const int64_t slot = solve_scratch_used;
            solve_scratch_used += (((int64_t)S.m * SOLVE_SLOT_COLUMNS) + 1) & ~(int64_t)1;
            const int64_t mref = which ? S.uinv : S.linv;
            int lmax = 0;
            const int dep0 = std::max(trS[(size_t)super_of[t]].dep_read_all(), X.tr->dep_read(cell0[t], cell1[t]));
            for (int r0 = 0; r0 < S.m; r0 += 64) {
                const int mc = std::min(64, S.m - r0);
                Task tk = blank(T_APPLY_DENSE);
                tk.flags = trans ? F_TRANS : 0;
                tk.kref = kref; tk.kconst = kconst;
                tk.m = mc; tk.n = S.m;
                tk.a = make_ref(SP_DIAG, mref + (trans ? (int64_t)r0 * S.m : (int64_t)r0)); tk.a_ld = S.m;
                tk.x = ref(X, T.offset[t]); tk.x_ld = X.ld;
                tk.y = make_ref(SP_SCRATCH, slot + r0); tk.y_ld = S.m;
                lmax = std::max(lmax, emit(tk, dep0));
            }
            trS[(size_t)super_of[t]].note_read_all(lmax);
            X.tr->note_read(cell0[t], cell1[t], lmax);
            Task rd = blank(T_REDUCE); // (no F_SUB: Y = the contribution)
            rd.kref = kref; rd.kconst = 1;
            rd.m = S.m;
            rd.a = (int64_t)aux_out->size() / 2;
            aux_out->push_back(make_ref(SP_SCRATCH, slot)); aux_out->push_back(S.m);
            rd.y = ref(X, T.offset[t]); rd.y_ld = X.ld;
            const int lev = emit(rd, std::max(lmax, X.tr->dep_write(cell0[t], cell1[t])));
            X.tr->note_write(cell0[t], cell1[t], lev);
            return;
        }
        Task tk = blank(T_APPLY_DENSE);
        tk.flags = F_INPLACE | (trans ? F_TRANS : 0);
        tk.kref = kref; tk.kconst = kconst;
        tk.m = tk.n = S.m;
        tk.a = make_ref(SP_DIAG, which ? S.uinv : S.linv); tk.a_ld = S.m;
        tk.x = tk.y = ref(X, T.offset[t]); tk.x_ld = tk.y_ld = X.ld;
        const int dep = std::max(trS[(size_t)super_of[t]].dep_read_all(), X.tr->dep_write(cell0[t], cell1[t]));
        const int lev = emit(tk, dep);
        trS[(size_t)super_of[t]].note_read_all(lev);
        X.tr->note_write(cell0[t], cell1[t], lev);
    }
    void plan_inverses() { // identity, then the sweep of the block itself on all its columns at once
        for (size_t q = 0; q < P.supers.size(); q++) {
            const Super &S = P.supers[q];
            const int t = S.node;
            for (int which = 0; which < (P.params.symmetric ? 1 : 2); which++) {
                temp_tracks.emplace_back();
                temp_tracks.back().init(cell0[t], cell1[t] - cell0[t]);
                Thin X;
                X.base = which ? S.uinv : S.linv; X.ld = S.m; X.pos0 = T.offset[t]; X.space = SP_DIAG; X.tr = &temp_tracks.back();
                Task f = blank(T_FILL);
                f.flags = F_IDENT;
                f.m = S.m; f.kref = -1; f.kconst = S.m;
                f.y = ref(X, T.offset[t]); f.y_ld = S.m;
                const int lev = emit(f, X.tr->dep_write(cell0[t], cell1[t]));
                X.tr->note_write(cell0[t], cell1[t], lev);
                if (which == 0) thin_solve_l(t, X, -1, S.m);
                else thin_solve_u(t, X, -1, S.m);
            }
        }
    }
    // ---- (solve programs) one step of a sweep: X[rows of the output side of block b] -= op(block b restricted to (t, s)) X[rows of its input side] ----
    std::vector<int> cellpos, cellsize;     // first position / rows of every cluster-leaf cell
    int64_t solve_scratch_used = 0;
    std::vector<int64_t> *aux_out = nullptr;
    struct Contrib { int64_t ref; int ld; int level; };
    void block_step(int b, int t, int s, bool trans, const Thin &X, int kref, int kconst) {
        if (planning_factor || !use_super || X.space != SP_RHS || !P.params.solve_slots || aux_out == nullptr) {
            for_leaves(b, t, s, [&](int l, int it, int is) { apply_leaf(l, it, is, trans, X, X, kref, kconst, true, true); });
            return;
        }
        const int outn = trans ? s : t, c0 = cell0[outn], c1 = cell1[outn];
        std::vector<std::vector<Contrib>> per_cell((size_t)(c1 - c0));
        for_leaves(b, t, s, [&](int l, int it, int is) {
            const Leaf &L = P.leaves[l];
            const int in = trans ? it : is, out = trans ? is : it;
            const int64_t slot = solve_scratch_used;
            solve_scratch_used += (((int64_t)T.size[out] * SOLVE_SLOT_COLUMNS) + 1) & ~(int64_t)1;
            Task tk = blank(L.kind == 1 ? T_APPLY_LR : T_APPLY_DENSE);
            tk.leaf = l;
            tk.flags = trans ? F_TRANS : 0; // (Y = op(leaf) X: no accumulation, the slot is this task's own)
            tk.kref = kref; tk.kconst = kconst;
            tk.m = T.size[out];
            tk.n = T.size[in];
            if (L.kind == 1) {
                const int64_t urow = L.u + (T.offset[it] - L.t_off), vrow = L.v + (T.offset[is] - L.s_off);
                tk.a = make_ref(SP_FACTOR, trans ? vrow : urow); tk.a_ld = trans ? L.n : L.m;
                tk.b = make_ref(SP_FACTOR, trans ? urow : vrow); tk.b_ld = trans ? L.m : L.n;
            } else { tk.a = make_ref(SP_FACTOR, L.u); tk.a_ld = L.m; }
            tk.x = ref(X, T.offset[in]); tk.x_ld = X.ld;
            tk.y = make_ref(SP_SCRATCH, slot); tk.y_ld = T.size[out];
            const int dep = std::max(dep_leaf_read(l), X.tr->dep_read(cell0[in], cell1[in]));
            const int lev = emit(tk, dep);
            note_leaf_read(l, lev);
            X.tr->note_read(cell0[in], cell1[in], lev);
            for (int c = cell0[out]; c < cell1[out]; c++)
                per_cell[(size_t)(c - c0)].push_back({make_ref(SP_SCRATCH, slot + (cellpos[c] - T.offset[out])), T.size[out], lev});
        });
        for (int c = c0; c < c1; c++) {
            const std::vector<Contrib> &v = per_cell[(size_t)(c - c0)];
            if (v.empty()) continue;
            Task tk = blank(T_REDUCE);
            tk.flags = F_SUB;
            tk.kref = kref; tk.kconst = (int32_t)v.size();
            tk.m = cellsize[c];
            tk.a = (int64_t)aux_out->size() / 2;
            tk.y = make_ref(X.space, X.base + (cellpos[c] - X.pos0)); tk.y_ld = X.ld;
            int dep = X.tr->dep_write(c, c + 1);
            for (const Contrib &q : v) { aux_out->push_back(q.ref); aux_out->push_back(q.ld); dep = std::max(dep, q.level); }
            const int lev = emit(tk, dep);
            X.tr->note_write(c, c + 1, lev);
        }
    }
    // X[rows of t] <- L(t,t)^-1 X[rows of t]   (forward substitution through the leaves of L)
    void thin_solve_l(int t, const Thin &X, int kref, int kconst) {
        if (T.size[t] == 0) return;
        if (use_super && super_of[t] >= 0) { apply_super(t, 0, false, X, kref, kconst); return; }
        if (bn[diag_bnode[t]].leaf >= 0) { apply_diag(t, 0, false, X, kref, kconst); return; }
        const int nc = T.n_child[t], f = T.first_child[t];
        for (int i = 0; i < nc; i++) {
            if (T.size[f + i] == 0) continue;
            thin_solve_l(f + i, X, kref, kconst);
            for (int j = i + 1; j < nc; j++) {
                if (T.size[f + j] == 0) continue;
                block_step(block_of(f + j, f + i), f + j, f + i, false, X, kref, kconst);
            }
        }
    }
    // X[rows of s] <- U(s,s)^-T X[rows of s]   (forward substitution with the transposed upper factor)
    void thin_solve_ut(int s, const Thin &X, int kref, int kconst) {
        if (T.size[s] == 0) return;
        if (use_super && super_of[s] >= 0) { apply_super(s, P.params.symmetric ? 0 : 1, true, X, kref, kconst); return; }
        if (bn[diag_bnode[s]].leaf >= 0) { apply_diag(s, 1, true, X, kref, kconst); return; }
        const int nc = T.n_child[s], f = T.first_child[s];
        for (int i = 0; i < nc; i++) {
            if (T.size[f + i] == 0) continue;
            thin_solve_ut(f + i, X, kref, kconst);
            for (int j = i + 1; j < nc; j++) {
                if (T.size[f + j] == 0) continue;
                block_step(block_of(f + i, f + j), f + i, f + j, true, X, kref, kconst);
            }
        }
    }
    // X[rows of t] <- U(t,t)^-1 X[rows of t]   (backward substitution)
    void thin_solve_u(int t, const Thin &X, int kref, int kconst) {
        if (T.size[t] == 0) return;
        if (use_super && super_of[t] >= 0) { apply_super(t, P.params.symmetric ? 0 : 1, false, X, kref, kconst); return; }
        if (bn[diag_bnode[t]].leaf >= 0) { apply_diag(t, 1, false, X, kref, kconst); return; }
        const int nc = T.n_child[t], f = T.first_child[t];
        for (int i = nc - 1; i >= 0; i--) {
            if (T.size[f + i] == 0) continue;
            thin_solve_u(f + i, X, kref, kconst);
            for (int j = i - 1; j >= 0; j--) {
                if (T.size[f + j] == 0) continue;
                block_step(block_of(f + j, f + i), f + j, f + i, false, X, kref, kconst);
            }
        }
    }
    // X[rows of t] <- L(t,t)^-T X[rows of t]   (backward substitution with the transposed lower factor)
    void thin_solve_lt(int t, const Thin &X, int kref, int kconst) {
        if (T.size[t] == 0) return;
        if (use_super && super_of[t] >= 0) { apply_super(t, 0, true, X, kref, kconst); return; }
        if (bn[diag_bnode[t]].leaf >= 0) { apply_diag(t, 0, true, X, kref, kconst); return; }
        const int nc = T.n_child[t], f = T.first_child[t];
        for (int i = nc - 1; i >= 0; i--) {
            if (T.size[f + i] == 0) continue;
            thin_solve_lt(f + i, X, kref, kconst);
            for (int j = i - 1; j >= 0; j--) {
                if (T.size[f + j] == 0) continue;
                block_step(block_of(f + i, f + j), f + i, f + j, true, X, kref, kconst);
            }
        }
    }

    // B(t, s) <- L(t,t)^-1 B(t, s)
    void solve_l(int t, int s, int b) {
        if (T.size[t] == 0 || T.size[s] == 0) return;
        b = descend(b, t, s);
        const BNode &B = bn[b];
        HM_CHECK(B.t == t && B.s == s, "hierarchical LU: a triangular solve met a block that is not a node of the block tree");
        if (B.leaf >= 0) {
            const Leaf &L = P.leaves[B.leaf];
            if (L.kind == 1) {
                ensure_final(B.leaf);
                Thin X = leaf_u(B.leaf);
                X.opleaf = -1; X.tr = &trU[B.leaf];
                thin_solve_l(t, X, B.leaf, 0);
            } else {
                Thin X = leaf_u(B.leaf);
                X.opleaf = -1; X.tr = &trU[B.leaf];
                apply_diag(t, 0, false, X, -1, L.n);
            }
            return;
        }
        const int ns = (B.split & 2) ? T.n_child[s] : 1;
        for (int j = 0; j < ns; j++) {
            const int sj = (B.split & 2) ? T.first_child[s] + j : s;
            if (!(B.split & 1)) { solve_l(t, sj, b); continue; }
            const int nc = T.n_child[t], f = T.first_child[t];
            for (int i = 0; i < nc; i++) {
                solve_l(f + i, sj, b);
                for (int i2 = i + 1; i2 < nc; i2++) mm(f + i2, f + i, sj, diag_bnode[t], b, b);
            }
        }
    }
    // B(t, s) <- B(t, s) U(s,s)^-1
    void solve_u(int t, int s, int b) {
        if (T.size[t] == 0 || T.size[s] == 0) return;
        b = descend(b, t, s);
        const BNode &B = bn[b];
        HM_CHECK(B.t == t && B.s == s, "hierarchical LU: a triangular solve met a block that is not a node of the block tree");
        if (B.leaf >= 0) {
            const Leaf &L = P.leaves[B.leaf];
            if (L.kind == 1) {
                ensure_final(B.leaf);
                Thin X = leaf_v(B.leaf);
                X.opleaf = -1; X.tr = &trV[B.leaf];
                thin_solve_ut(s, X, B.leaf, 0);
            } else { // D <- D U^-1: the rows of D^T, in place
                const int ld = bn[diag_bnode[s]].leaf;
                const Diag &D = P.diags[P.leaves[ld].diag];
                Task tk = blank(T_APPLY_DENSE);
                tk.leaf = ld;
                tk.flags = F_INPLACE | F_TRANS | F_XT | F_YT;
                tk.kref = -1; tk.kconst = L.m;
                tk.m = tk.n = D.m;
                tk.a = make_ref(SP_DIAG, D.uinv); tk.a_ld = D.m;
                tk.x = tk.y = make_ref(SP_FACTOR, L.u); tk.x_ld = tk.y_ld = L.m;
                const int dep = std::max(trD[P.leaves[ld].diag].dep_read_all(), trU[B.leaf].dep_write_all());
                const int lev = emit(tk, dep);
                trD[P.leaves[ld].diag].note_read_all(lev);
                trU[B.leaf].note_write_all(lev);
            }
            return;
        }
        const int nt = (B.split & 1) ? T.n_child[t] : 1;
        for (int i = 0; i < nt; i++) {
            const int ti = (B.split & 1) ? T.first_child[t] + i : t;
            if (!(B.split & 2)) { solve_u(ti, s, b); continue; }
            const int nc = T.n_child[s], f = T.first_child[s];
            for (int j = 0; j < nc; j++) {
                solve_u(ti, f + j, b);
                for (int j2 = j + 1; j2 < nc; j2++) mm(ti, f + j, f + j2, b, diag_bnode[s], b);
            }
        }
    }
    // ---- symmetric positive definite operators: A = L L^T on the lower triangle ----
    // C(t, s) -= A(t, r) A'(s, r)^T with a covering (t, r), b covering (s, r) (both below the diagonal), c covering (t, s); of C only what
    // lies on or below the diagonal exists
    void mm_sym(int t, int r, int s, int a, int b, int c) {
        if (T.size[t] == 0 || T.size[r] == 0 || T.size[s] == 0) return;
        if (end(t) <= T.offset[s]) return; // strictly above the diagonal
        a = descend(a, t, r);
        b = descend(b, s, r);
        c = descend_soft(c, t, s);
        const BNode &A = bn[a], &B = bn[b];
        const bool a_lr = A.leaf >= 0 && P.leaves[A.leaf].kind == 1, b_lr = B.leaf >= 0 && P.leaves[B.leaf].kind == 1;
        if (a_lr) { // (U_a V_a^T) B^T = U_a (B V_a)^T
            begin_group();
            const int la = A.leaf;
            ensure_final(la);
            Thin Z = new_scratch(s, P.leaves[la].cap);
            fill_zero(Z, s, la, 0);
            const Thin Va = leaf_v(la), Ua = leaf_u(la);
            for_leaves(b, s, r, [&](int l, int it, int is) { apply_leaf(l, it, is, false, Va, Z, la, 0, true, false); });
            for_leaves(c, t, s, [&](int l, int it, int is) { add_lr(l, it, is, Ua, Z, la, 0); });
            return;
        }
        if (b_lr) { // A (U_b V_b^T)^T = (A V_b) U_b^T
            begin_group();
            const int lb = B.leaf;
            ensure_final(lb);
            Thin X = new_scratch(t, P.leaves[lb].cap);
            fill_zero(X, t, lb, 0);
            const Thin Ub = leaf_u(lb), Vb = leaf_v(lb);
            for_leaves(a, t, r, [&](int l, int it, int is) { apply_leaf(l, it, is, false, Vb, X, lb, 0, true, false); });
            for_leaves(c, t, s, [&](int l, int it, int is) { add_lr(l, it, is, X, Ub, lb, 0); });
            return;
        }
        if (A.leaf >= 0 && B.leaf >= 0) { // two dense leaves (t x r) and (s x r)
            begin_group();
            const Leaf &LA = P.leaves[A.leaf], &LB = P.leaves[B.leaf];
            const BNode &C = bn[c];
            HM_CHECK(C.leaf >= 0, "hierarchical Cholesky: the target of a product of dense leaves is not a leaf");
            Thin Da = leaf_u(A.leaf), Db = leaf_u(B.leaf); // D_b as it is: rows = the positions of s
            if (P.leaves[C.leaf].kind == 0) { add_lr(C.leaf, t, s, Da, Db, -1, LA.n); return; }
            const int kp = std::min(LA.m, LB.m);
            Thin X = new_scratch(t, kp), Z = new_scratch(s, kp);
            const int64_t w = scratch_used;
            scratch_used += ((int64_t)LA.m * LB.m + 1) & ~(int64_t)1;
            Task tk = blank(T_DDPROD);
            tk.leaf = C.leaf;
            tk.flags = F_TRANS;
            tk.kref = (int)next_slot++;
            tk.kconst = kp;
            tk.m = LA.m; tk.n = LB.m; tk.r0 = LA.n;
            tk.a = make_ref(SP_FACTOR, LA.u); tk.a_ld = LA.m;
            tk.b = make_ref(SP_FACTOR, LB.u); tk.b_ld = LB.m;
            tk.x = ref(X, T.offset[t]); tk.x_ld = X.ld;
            tk.y = ref(Z, T.offset[s]); tk.y_ld = Z.ld;
            tk.w = make_ref(SP_SCRATCH, w);
            const int lev = emit(tk, std::max(dep_leaf_read(A.leaf), dep_leaf_read(B.leaf)));
            note_leaf_read(A.leaf, lev);
            note_leaf_read(B.leaf, lev);
            X.tr->note_write(cell0[t], cell1[t], lev);
            Z.tr->note_write(cell0[s], cell1[s], lev);
            add_lr(C.leaf, t, s, X, Z, tk.kref, 0);
            return;
        }
        const bool st = A.leaf < 0 && (A.split & 1) && t == A.t;
        const bool sr = (A.leaf < 0 && (A.split & 2) && r == A.s) || (B.leaf < 0 && (B.split & 2) && r == B.s);
        const bool ss = B.leaf < 0 && (B.split & 1) && s == B.t;
        HM_CHECK(st || sr || ss, "hierarchical Cholesky: the product recursion cannot descend");
        HM_CHECK((!st || T.n_child[t] > 0) && (!sr || T.n_child[r] > 0) && (!ss || T.n_child[s] > 0), "hierarchical Cholesky: split of a cluster leaf");
        const int nt = st ? T.n_child[t] : 1, nr = sr ? T.n_child[r] : 1, ns = ss ? T.n_child[s] : 1;
        for (int i = 0; i < nt; i++)
            for (int k = 0; k < nr; k++)
                for (int j = 0; j < ns; j++)
                    mm_sym(st ? T.first_child[t] + i : t, sr ? T.first_child[r] + k : r, ss ? T.first_child[s] + j : s, a, b, c);
    }
    // B(t, s) <- B(t, s) L(s,s)^-T   (t below s)
    void solve_lt_sym(int t, int s, int b) {
        if (T.size[t] == 0 || T.size[s] == 0) return;
        b = descend(b, t, s);
        const BNode &B = bn[b];
        HM_CHECK(B.t == t && B.s == s, "hierarchical Cholesky: a triangular solve met a block that is not a node of the block tree");
        if (B.leaf >= 0) {
            const Leaf &L = P.leaves[B.leaf];
            if (L.kind == 1) { // V <- L^-1 V
                ensure_final(B.leaf);
                Thin X = leaf_v(B.leaf);
                X.opleaf = -1; X.tr = &trV[B.leaf];
                thin_solve_l(s, X, B.leaf, 0);
            } else { // D <- D L^-T: the rows of D, in place
                const int ld = bn[diag_bnode[s]].leaf;
                const Diag &D = P.diags[P.leaves[ld].diag];
                Task tk = blank(T_APPLY_DENSE);
                tk.leaf = ld;
                tk.flags = F_INPLACE | F_XT | F_YT;
                tk.kref = -1; tk.kconst = L.m;
                tk.m = tk.n = D.m;
                tk.a = make_ref(SP_DIAG, D.linv); tk.a_ld = D.m;
                tk.x = tk.y = make_ref(SP_FACTOR, L.u); tk.x_ld = tk.y_ld = L.m;
                const int dep = std::max(trD[P.leaves[ld].diag].dep_read_all(), trU[B.leaf].dep_write_all());
                const int lev = emit(tk, dep);
                trD[P.leaves[ld].diag].note_read_all(lev);
                trU[B.leaf].note_write_all(lev);
            }
            return;
        }
        const int nt = (B.split & 1) ? T.n_child[t] : 1;
        for (int i = 0; i < nt; i++) {
            const int ti = (B.split & 1) ? T.first_child[t] + i : t;
            if (!(B.split & 2)) { solve_lt_sym(ti, s, b); continue; }
            const int nc = T.n_child[s], f = T.first_child[s];
            for (int j = 0; j < nc; j++) {
                solve_lt_sym(ti, f + j, b);
                for (int j2 = j + 1; j2 < nc; j2++) mm_sym(ti, f + j, f + j2, b, diag_bnode[s], b);
            }
        }
    }
    void chol(int t) {
        if (T.size[t] == 0) return;
        const int d = diag_bnode[t];
        HM_CHECK(d >= 0, "hierarchical Cholesky: a diagonal block is missing");
        if (bn[d].leaf >= 0) {
            const int l = bn[d].leaf;
            const Leaf &L = P.leaves[l];
            HM_CHECK(L.kind == 0 && L.diag >= 0, "hierarchical Cholesky: a diagonal leaf is not dense");
            Task tk = blank(T_GETRF);
            tk.leaf = l;
            tk.flags = F_SYM;
            tk.m = tk.n = L.m;
            const int lev = emit(tk, std::max(trU[l].dep_write_all(), trD[L.diag].dep_write_all()));
            trU[l].note_write_all(lev);
            trD[L.diag].note_write_all(lev);
            return;
        }
        const int nc = T.n_child[t], f = T.first_child[t];
        for (int i = 0; i < nc; i++) {
            if (T.size[f + i] == 0) continue;
            chol(f + i);
            for (int j = i + 1; j < nc; j++) solve_lt_sym(f + j, f + i, d);
            for (int j = i + 1; j < nc; j++)
                for (int k = i + 1; k <= j; k++) mm_sym(f + j, f + i, f + k, d, d, d);
        }
    }
    void lu(int t) {
        if (T.size[t] == 0) return;
        const int d = diag_bnode[t];
        HM_CHECK(d >= 0, "hierarchical LU: a diagonal block is missing");
        if (bn[d].leaf >= 0) {
            const int l = bn[d].leaf;
            const Leaf &L = P.leaves[l];
            HM_CHECK(L.kind == 0 && L.diag >= 0, "hierarchical LU: a diagonal leaf is not dense");
            Task tk = blank(T_GETRF);
            tk.leaf = l;
            tk.m = tk.n = L.m;
            const int lev = emit(tk, std::max(trU[l].dep_write_all(), trD[L.diag].dep_write_all()));
            trU[l].note_write_all(lev);
            trD[L.diag].note_write_all(lev);
            return;
        }
        const int nc = T.n_child[t], f = T.first_child[t];
        for (int i = 0; i < nc; i++) {
            if (T.size[f + i] == 0) continue;
            lu(f + i);
            for (int j = i + 1; j < nc; j++) {
                solve_l(f + i, f + j, d);
                solve_u(f + j, f + i, d);
            }
            for (int j = i + 1; j < nc; j++)
                for (int k = i + 1; k < nc; k++) mm(f + j, f + i, f + k, d, d, d);
        }
    }
};

} // namespace

Plan *make_plan(const ClusterTree &T, const std::vector<LeafIn> &in, const Params &prm, int root) {
    const double t0 = wall_seconds();
    HM_CHECK(prm.cap_max <= 64 && prm.cap_min >= 8 && prm.cap_min <= prm.cap_max, "hierarchical LU: leaf capacities must lie in [8, 64]");
    std::unique_ptr<Plan> plan(new Plan);
    Plan &P = *plan;
    P.tree = &T;
    P.params = prm;
    HM_CHECK(root >= 0 && root < T.node_count(), "hierarchical LU: unknown root node");
    P.root = root;
    P.n = T.size[root];
    P.pos0 = T.offset[root];
    Emitter E(T, P);
    const int nn = T.node_count();
    // cells: the cluster leaves in position order
    E.cell0.assign(nn, 0);
    E.cell1.assign(nn, 0);
    {
        std::vector<int> leaves;
        for (int v = 0; v < nn; v++) if (T.is_leaf(v) && T.size[v] > 0) leaves.push_back(v);
        std::sort(leaves.begin(), leaves.end(), [&](int x, int y) { return T.offset[x] < T.offset[y]; });
        for (size_t c = 0; c < leaves.size(); c++) { E.cell0[leaves[c]] = (int)c; E.cell1[leaves[c]] = (int)c + 1; E.cellpos.push_back(T.offset[leaves[c]]); E.cellsize.push_back(T.size[leaves[c]]); }
        for (int v = nn - 1; v >= 0; v--) { // children have larger ids than their parent
            if (T.is_leaf(v)) continue;
            int lo = INT32_MAX, hi = 0;
            for (int a = 0; a < T.n_child[v]; a++) {
                const int c = T.first_child[v] + a;
                if (T.size[c] == 0) continue;
                lo = std::min(lo, E.cell0[c]); hi = std::max(hi, E.cell1[c]);
            }
            E.cell0[v] = lo == INT32_MAX ? 0 : lo; E.cell1[v] = hi;
        }
    }
    // leaves and arenas
    std::unordered_map<uint64_t, int> leaf_of;
    leaf_of.reserve(in.size() * 2);
    P.leaves.resize(in.size());
    E.leaf_t.resize(in.size());
    E.leaf_s.resize(in.size());
    int64_t fe = 0, de = 0;
    for (size_t i = 0; i < in.size(); i++) {
        const LeafIn &li = in[i];
        HM_CHECK(li.t_node >= 0 && li.t_node < nn && li.s_node >= 0 && li.s_node < nn, "hierarchical LU: leaf with an unknown cluster node");
        Leaf &L = P.leaves[i];
        L.t_off = T.offset[li.t_node]; L.m = T.size[li.t_node];
        L.s_off = T.offset[li.s_node]; L.n = T.size[li.s_node];
        L.kind = li.rank >= 0 ? 1 : 0;
        L.rank0 = li.rank;
        L.diag = -1;
        L.u = fe;
        if (L.kind == 1) {
            // (small leaves next to the diagonal gain the most rank from the Schur complements and cost next to nothing: they get room for half their size)
            L.cap = std::min(prm.cap_max, std::max(std::max(prm.cap_min, std::min(32, std::min(L.m, L.n) / 2)), (int)std::ceil(prm.cap_factor * li.rank) + prm.cap_extra));
            HM_CHECK(li.rank <= L.cap - 4, "hierarchical LU: the rank of a leaf exceeds what the low-rank arithmetic is sized for");
            fe += ((int64_t)L.m * L.cap + 1) & ~(int64_t)1;
            L.v = fe;
            fe += ((int64_t)L.n * L.cap + 1) & ~(int64_t)1;
        } else {
            L.cap = 0;
            L.v = 0;
            fe += ((int64_t)L.m * L.n + 1) & ~(int64_t)1;
            if (li.t_node == li.s_node) {
                L.diag = (int)P.diags.size();
                Diag D;
                D.leaf = (int)i; D.m = L.m;
                D.linv = de; de += (int64_t)L.m * L.m;
                D.uinv = de; de += (int64_t)L.m * L.m;
                P.diags.push_back(D);
            }
        }
        E.leaf_t[i] = li.t_node; E.leaf_s[i] = li.s_node;
        const bool fresh = leaf_of.emplace(((uint64_t)(uint32_t)li.t_node << 32) | (uint32_t)li.s_node, (int)i).second;
        HM_CHECK(fresh, "hierarchical LU: a block appears twice among the leaves");
    }
    P.factor_elems = fe;
    P.n_real_leaves = (int64_t)in.size();
    E.next_slot = (int64_t)in.size();
    E.diag_bnode.assign(nn, -1);
    E.build_bnode(root, root, leaf_of);
    {
        int64_t used = 0;
        for (const BNode &b : E.bn) used += b.leaf >= 0;
        HM_CHECK(used == (int64_t)in.size(), "hierarchical LU: leaves outside the block tree of the cluster tree (an operator built with other splitting rules)");
    }
    E.super_of.assign(nn, -1);
    if (prm.super_rows > 0) E.choose_supers(root, de);
    E.trS.resize(P.supers.size());
    P.diag_elems = de;
    E.trU.resize(in.size());
    E.trV.resize(in.size());
    E.trD.resize(P.diags.size());
    E.dirty.assign(in.size(), 0);
    for (size_t i = 0; i < in.size(); i++) {
        E.trU[i].init(E.cell0[in[i].t_node], E.cell1[in[i].t_node] - E.cell0[in[i].t_node]);
        E.trV[i].init(E.cell0[in[i].s_node], E.cell1[in[i].s_node] - E.cell0[in[i].s_node]);
    }
    // window 0: every low-rank leaf is truncated once (ranks of the compression -> ranks of the arithmetic; norms)
    for (size_t i = 0; i < in.size(); i++) if (P.leaves[i].kind == 1) { E.dirty[i] = 1; E.ensure_final((int)i); }
    E.close_window();
    if (prm.symmetric) E.chol(root);
    else E.lu(root);
    for (size_t i = 0; i < in.size(); i++) E.ensure_final((int)i);
    E.close_window();
    P.n_slots = E.next_slot;
    P.leaves.resize((size_t)E.next_slot, Leaf{0, 0, 0, 0, 0, 0, 0, 0, -1, 0}); // (one record per rank slot: the stage blocks have theirs, the others are unused)
    // the inverse factors of the small diagonal blocks, then the two solves: fresh dependency state, the factors are read only
    E.planning_factor = false;
    if (!P.supers.empty()) {
        for (Track &t : E.trU) t = Track();
        for (Track &t : E.trV) t = Track();
        for (Track &t : E.trD) t = Track();
        E.window_base = 1; E.max_level = 0; E.scratch_used = 0;
        E.temp_tracks.clear();
        int64_t keep[T_NTYPES];
        for (int q = 0; q < T_NTYPES; q++) keep[q] = P.counts[q];
        E.plan_inverses();
        for (int q = 0; q < T_NTYPES; q++) P.counts[q] = keep[q];
        E.finish_program(E.cur, P.invert);
        E.temp_tracks.clear();
    }
    E.use_super = true;
    for (int pass = 0; pass < 2; pass++) {
        for (Track &t : E.trU) t = Track();
        for (Track &t : E.trV) t = Track();
        for (Track &t : E.trD) t = Track();
        E.window_base = 1; E.max_level = 0; E.scratch_used = 0;
        Track rhs;
        rhs.init(E.cell0[root], E.cell1[root] - E.cell0[root]);
        Thin X;
        X.base = 0; X.ld = -1; X.pos0 = P.pos0; X.space = SP_RHS; X.tr = &rhs;
        int64_t keep[T_NTYPES];
        for (int q = 0; q < T_NTYPES; q++) keep[q] = P.counts[q];
        Program &SG = pass == 0 ? P.solve_n : P.solve_t;
        E.aux_out = &SG.aux;
        E.solve_scratch_used = 0;
        if (prm.symmetric) { E.thin_solve_l(root, X, -2, 0); E.thin_solve_lt(root, X, -2, 0); } // A = L L^T: the same sweeps for A and A^T
        else if (pass == 0) { E.thin_solve_l(root, X, -2, 0); E.thin_solve_u(root, X, -2, 0); }
        else { E.thin_solve_ut(root, X, -2, 0); E.thin_solve_lt(root, X, -2, 0); }
        for (int q = 0; q < T_NTYPES; q++) P.counts[q] = keep[q];
        E.finish_program(E.cur, SG);
        SG.scratch_elems = E.solve_scratch_used; // (the private slots of the sweeps: SOLVE_SLOT_COLUMNS right-hand sides at a time)
    }
    P.plan_seconds = wall_seconds() - t0;
    return plan.release();
}

} // namespace hlu
} // namespace hm
