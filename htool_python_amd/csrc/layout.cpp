// layout.cpp -- where every leaf's panels go in the tile-major HBM layout (hmatrix.hpp header).
//
// Columns of a row tile are ordered top-down through the target cluster tree: first the leaves
// attached to the tile's shallowest ancestor, ..., last those attached to the cluster leaf itself
// (low-rank before dense).  With that order the first column of a leaf (ucol) is the same in every
// tile the leaf covers, so one (ucol, vcol) pair per leaf describes the whole scatter.
#include <algorithm>

#include "hmatrix.hpp"

namespace hm {

static inline int round_up(int v, int q) { return (v + q - 1) / q * q; }

// Everything of a batch layout that depends on the leaves only through per-node sums -- Klr / Kdn: phase-B columns a target node
// contributes (low rank / dense), Ks: phase-A rows a source node contributes -- : first columns and rows of the nodes, the per-tile
// tables, region R of W, the slots of the transposed use.  O(nodes + tiles): the host-driven build calls it between its two
// per-leaf passes, the device-resident build (device_build2.inc) forms the sums and finishes the leaves on the GPU.
void compute_node_layout(HMatrix &H, const std::vector<int> &Klr, const std::vector<int> &Kdn, const std::vector<int> &Ks, int vec_rows, BatchLayout &L, NodeLayout &NL) {
    const ClusterTree &T = *H.tc, &S = *H.sc;
    const int TM = H.tile_max;
    const int nt = T.node_count(), ns = S.node_count();
    std::vector<int> &tbase = NL.tbase;
    tbase.assign(nt, 0);
    for (int id = 0; id < nt; id++) { // parents precede children in the node table
        int p = T.parent[id];
        if (id != H.t_root && p >= 0) tbase[id] = tbase[p] + Klr[p] + Kdn[p];
    }
    tbase[H.t_root] = 0;
    std::vector<int> &sbase = NL.sbase;
    sbase.assign(ns, 0);
    for (int id = 1; id < ns; id++) sbase[id] = sbase[S.parent[id]] + Ks[S.parent[id]];
    const int nrt = H.rtiles.count();
    L.b_ncols.assign(nrt, 0);
    L.b_pbase.assign(nrt, 0);
    L.b_cbase.assign(nrt, 0);
    int64_t pb = 0, cb = 0;
    for (int r = 0; r < nrt; r++) {
        int leaf = H.rtiles.leaf_of_tile[r];
        int nc = tbase[leaf] + Klr[leaf] + Kdn[leaf];
        L.b_ncols[r] = nc;
        L.b_pbase[r] = pb;
        L.b_cbase[r] = cb;
        pb += (int64_t)nc * round_up(H.rtiles.size[r], vec_rows);
        cb += nc;
    }
    L.panelB_elems = pb;
    L.cidxB_elems = cb;

    // ---- source side: the tile table
    const int nct = H.ctiles.count();
    L.a_nrows.assign(nct, 0);
    L.a_pbase.assign(nct, 0);
    L.a_obase.assign(nct, 0);
    L.a_flush.assign(nct, 0);
    const std::vector<int> &gid = H.ctile_group;
    int64_t pa = 0, oa = 0;
    for (int c = 0; c < nct; c++) {
        int leaf = H.ctiles.leaf_of_tile[c];
        int nr = sbase[leaf] + Ks[leaf];
        L.a_nrows[c] = nr;
        L.a_pbase[c] = pa;
        L.a_obase[c] = oa;
        if (nr > 0) {
            int nq = (nr + TM - 1) / TM;
            int padded = TM * (nq - 1) + round_up(nr - TM * (nq - 1), vec_rows);
            pa += (int64_t)padded * H.ctiles.size[c];
        }
        oa += nr;
    }
    L.panelA_elems = pa;
    L.oidxA_elems = oa;
    // rows a tile shares with its successor in the same group keep accumulating; the others are written out after it.  The
    // shared rows are those of the common ancestors: everything up to and including the lowest one
    for (int c = 0; c + 1 < nct; c++) {
        if (gid[c] != gid[c + 1]) continue; // last tile of its group: all rows are written
        int a = H.ctiles.leaf_of_tile[c], b = H.ctiles.leaf_of_tile[c + 1];
        while (a != b) {
            if (S.depth[a] >= S.depth[b]) a = S.parent[a];
            else b = S.parent[b];
        }
        L.a_flush[c] = sbase[a] + Ks[a];
    }

    // ---- region R of W: t vectors, and partial panels for source nodes spanning several tiles
    const int64_t r_start = round_up(H.col_size + 1, 2); // W index of R[0]
    std::vector<int64_t> &tb = NL.tb, &pbse = NL.pbse;
    std::vector<int> &ldp = NL.ldp;
    tb.assign(ns, -1); pbse.assign(ns, -1); ldp.assign(ns, 0);
    NL.r_start = r_start;
    int64_t cur = H.r_elems;
    for (int id = 0; id < ns; id++) {
        if (Ks[id] == 0) continue;
        tb[id] = cur;
        cur += Ks[id];
        // partial sums: one per GROUP of source tiles the node spans (a node inside one group is summed by that group's workgroup)
        const int t0 = H.ctiles.node_tile_begin[id], t1 = H.ctiles.node_tile_end[id];
        const int P = t1 > t0 ? gid[t1 - 1] - gid[t0] + 1 : 0;
        if (P > 1) {
            cur = (cur + 1) / 2 * 2;
            ldp[id] = round_up(Ks[id], vec_rows);
            pbse[id] = cur;
            cur += (int64_t)P * ldp[id];
            L.reduces.push_back({r_start + pbse[id], ldp[id], Ks[id], P, r_start + tb[id]});
        }
    }
    H.r_elems = cur;
    // ---- one-triangle storage / transposed products: slots for the transposed use of every phase-B column.  Per target node the
    // columns (low-rank first, then dense, in ucol order) get a contiguous block of final slots; nodes spanning several row
    // tiles also get a [tile][column] panel of partials that is summed after phase B (same scheme as the source side).
    if (H.one_triangle || H.transposable) {
        std::vector<int64_t> &zf = NL.zf, &zp = NL.zp;
        std::vector<int> &zld = NL.zld;
        zf.assign(nt, -1); zp.assign(nt, -1); zld.assign(nt, 0);
        int64_t cur2 = H.r_elems;
        for (int id = 0; id < nt; id++) {
            const int K = Klr[id] + Kdn[id];
            if (K == 0) continue;
            zf[id] = cur2;
            cur2 += K;
            const int P = H.rtiles.node_tile_end[id] - H.rtiles.node_tile_begin[id];
            if (P > 1) {
                cur2 = (cur2 + 1) / 2 * 2;
                zld[id] = round_up(K, vec_rows);
                zp[id] = cur2;
                cur2 += (int64_t)P * zld[id];
                L.z_reduces.push_back({r_start + zp[id], zld[id], K, P, r_start + zf[id]});
            }
        }
        H.r_elems = cur2;
    }

}

void compute_batch_layout(HMatrix &H, const std::vector<int64_t> &batch_blocks, int vec_rows, BatchLayout &L) {
    const ClusterTree &T = *H.tc, &S = *H.sc;
    const int nt = T.node_count(), ns = S.node_count();
    L.blocks = batch_blocks;

    // ---- target side: columns per node, low-rank first then dense; source side: rows (leaf, k) per node, low-rank leaves only.
    // ONE sequential pass over the leaves hands out the positions inside their nodes (the order of the batch is the order of the
    // columns); everything that only adds per-node bases afterwards runs on all threads (a 1 M-point batch is 75 MB of records).
    std::vector<int> Klr(nt, 0), Kdn(nt, 0);
    std::vector<int> Ks(ns, 0);
    std::vector<int> cs_of(batch_blocks.size(), 0);
    for (size_t q = 0; q < batch_blocks.size(); q++) {
        BlockRec &b = H.blocks()[batch_blocks[q]];
        if (b.rank >= 0) {
            b.ucol = Klr[b.t_node]; Klr[b.t_node] += b.rank;
            cs_of[q] = Ks[b.s_node]; b.vcol = Ks[b.s_node]; Ks[b.s_node] += b.rank;
        } else {
            b.ucol = Kdn[b.t_node]; Kdn[b.t_node] += b.n; // (behind the node's low-rank columns: their count is added below)
        }
    }
    NodeLayout NL;
    compute_node_layout(H, Klr, Kdn, Ks, vec_rows, L, NL);
    const std::vector<int> &tbase = NL.tbase, &sbase = NL.sbase;
    const int64_t r_start = NL.r_start;
    parallel_for((long long)batch_blocks.size(), [&](long long q) {
        BlockRec &b = H.blocks()[batch_blocks[(size_t)q]];
        b.ucol += tbase[b.t_node] + (b.rank < 0 ? Klr[b.t_node] : 0);
        if (b.rank < 0) return;
        b.vcol += sbase[b.s_node];
        const int id = b.s_node;
        b.tpos = r_start + NL.tb[id] + cs_of[(size_t)q];
        if (NL.pbse[id] >= 0) { b.v_obase = r_start + NL.pbse[id] + cs_of[(size_t)q]; b.v_ostride = NL.ldp[id]; }
        else { b.v_obase = b.tpos; b.v_ostride = 0; }
    });
    if (H.one_triangle || H.transposable) {
        for (int64_t bi : batch_blocks) {
            BlockRec &b = H.blocks()[bi];
            const int local = b.ucol - tbase[b.t_node]; // column of the leaf inside its node's block
            b.zfin = r_start + NL.zf[b.t_node] + local;
            if (NL.zp[b.t_node] >= 0) { b.z_obase = r_start + NL.zp[b.t_node] + local; b.z_ostride = NL.zld[b.t_node]; }
            else { b.z_obase = b.zfin; b.z_ostride = 0; }
            if (H.one_triangle && b.t_off == b.s_off) continue; // one-triangle storage: a diagonal leaf is applied once
            if (b.rank < 0) // dense: A^T x goes to the y rows of the leaf's source cluster, tile by tile
                for (int c = H.ctiles.node_tile_begin[b.s_node]; c < H.ctiles.node_tile_end[b.s_node]; c++) {
                    L.zd_tile.push_back(c);
                    L.zd_woff.push_back(b.zfin + (H.ctiles.off[c] - b.s_off));
                }
        }
    }

    // ---- pack work items: (leaf, row tile) and (low-rank leaf, source tile) pairs in leaf order; counted, then written by all threads
    // (or expanded on the device from the counts: L.host_items = false)
    const size_t nb = batch_blocks.size();
    std::vector<int64_t> u_first(nb + 1, 0), v_first(nb + 1, 0);
    for (size_t q = 0; q < nb; q++) {
        const BlockRec &b = H.blocks()[batch_blocks[q]];
        u_first[q + 1] = u_first[q] + (H.rtiles.node_tile_end[b.t_node] - H.rtiles.node_tile_begin[b.t_node]);
        v_first[q + 1] = v_first[q] + (b.rank >= 0 ? H.ctiles.node_tile_end[b.s_node] - H.ctiles.node_tile_begin[b.s_node] : 0);
    }
    HM_CHECK(u_first[nb] < ((int64_t)1 << 31) && v_first[nb] < ((int64_t)1 << 31), "a batch has more pack work items than a 32-bit index counts");
    L.u_first.assign(u_first.begin(), u_first.end());
    L.v_first.assign(v_first.begin(), v_first.end());
    if (!L.host_items) return;
    L.u_item_block.resize((size_t)u_first[nb]); L.u_item_tile.resize((size_t)u_first[nb]);
    L.v_item_block.resize((size_t)v_first[nb]); L.v_item_tile.resize((size_t)v_first[nb]);
    parallel_for((long long)nb, [&](long long q) {
        const BlockRec &b = H.blocks()[batch_blocks[(size_t)q]];
        int64_t pos = u_first[(size_t)q];
        for (int r = H.rtiles.node_tile_begin[b.t_node]; r < H.rtiles.node_tile_end[b.t_node]; r++, pos++) {
            L.u_item_block[(size_t)pos] = (int)q;
            L.u_item_tile[(size_t)pos] = r;
        }
        if (b.rank >= 0) {
            pos = v_first[(size_t)q];
            for (int c = H.ctiles.node_tile_begin[b.s_node]; c < H.ctiles.node_tile_end[b.s_node]; c++, pos++) {
                L.v_item_block[(size_t)pos] = (int)q;
                L.v_item_tile[(size_t)pos] = c;
            }
        }
    });
}

} // namespace hm
