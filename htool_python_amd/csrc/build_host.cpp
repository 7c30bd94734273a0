// build_host.cpp -- leaf filling for CALLBACK generators (user code in the host language).
//
// A callback generator (Python VirtualGenerator.build_submatrix, reference trampoline
// src/htool/hmatrix/interfaces/virtual_generator.hpp:16-25) can only be evaluated on the host, on
// the calling thread (the reference builds with HTOOL_WITH_PYTHON_INTERFACE for the same reason,
// CMakeLists.txt:97).  This file therefore runs, on the host: the partially pivoted ACA that drives
// the callback row by row / column by column, or the user's compressor
// (virtual_low_rank_generator.hpp:25-45), and the dense fill (virtual_generator.hpp or
// virtual_dense_blocks_generator.hpp:21-35).  The resulting panels go to a host arena laid out like
// the device's temporary arena and are packed and multiplied on the GPU exactly like device-built
// ones.  Native generators never come here (device_build_native).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "hmatrix.hpp"
#include "aca_stop.hpp"

namespace hm {

namespace {

inline double abs2(double x) { return x * x; }
inline double abs2(const cplx &x) { return std::norm(x); }
inline double cj(double x) { return x; }
inline cplx cj(const cplx &x) { return std::conj(x); }
inline double re(double x) { return x; }
inline double re(const cplx &x) { return x.real(); }

// returns rank (>=0) or -1 when the block is not worth storing in low-rank form.
// swp: run on the transposed block ("sym" role rule, SURVEY.md A.4: leaves below the diagonal), so that for a
// symmetric generator the leaves (t,s) and (s,t) get exactly transposed factors.  U (M0 x r, [k][i]) and
// V (r x N0, [k][j]) always refer to the block as given.
template <typename T>
int host_aca(const Generator &g, int M0, int N0, const int *rows0, const int *cols0, double eps, int reqrank, bool swp, std::vector<T> &Uout, std::vector<T> &Vout) {
    const int M = swp ? N0 : M0, N = swp ? M0 : N0; // sizes of the matrix the algorithm sees
    std::vector<T> U, V;
    std::vector<char> urow(M, 0), ucol(N, 0);
    std::vector<T> r(N), c(M);
    // row I / column J of the matrix the algorithm sees, through the user's generator
    auto get_row = [&](int I, T *out) { if (swp) g.fn(g.ctx, N, 1, rows0, cols0 + I, out); else g.fn(g.ctx, 1, N, rows0 + I, cols0, out); };
    auto get_col = [&](int J, T *out) { if (swp) g.fn(g.ctx, 1, M, rows0 + J, cols0, out); else g.fn(g.ctx, M, 1, rows0, cols0 + J, out); };
    int k = 0, I = 0;
    double frob2 = 0;
    const int kmax = std::min(M, N);
    int result = -2;
    AcaStop stop;
    const int confirm = aca_confirm_steps(reqrank);
    while (k < kmax) {
        if (reqrank >= 0 && k >= reqrank) break;
        get_row(I, r.data());
        for (int l = 0; l < k; l++) {
            T u = U[(size_t)l * M + I];
            const T *v = &V[(size_t)l * N];
            for (int j = 0; j < N; j++) r[j] -= u * v[j];
        }
        urow[I] = 1;
        int J = -1;
        double best = -1;
        for (int j = 0; j < N; j++)
            if (!ucol[j]) {
                double a = abs2(r[j]);
                if (a > best) best = a, J = j;
            }
        if (J < 0) break;
        if (std::sqrt(best) <= 1e-15) {
            int nI = -1;
            for (int i = 0; i < M; i++)
                if (!urow[i]) { nI = i; break; }
            if (nI < 0) break;
            I = nI;
            continue;
        }
        T piv = r[J];
        get_col(J, c.data());
        for (int l = 0; l < k; l++) {
            T v = V[(size_t)l * N + J];
            const T *u = &U[(size_t)l * M];
            for (int i = 0; i < M; i++) c[i] -= v * u[i];
        }
        T inv = T(1) / piv;
        for (int i = 0; i < M; i++) c[i] *= inv;
        ucol[J] = 1;
        double cn2 = 0, rn2 = 0;
        for (int i = 0; i < M; i++) cn2 += abs2(c[i]);
        for (int j = 0; j < N; j++) rn2 += abs2(r[j]);
        double cross = 0;
        for (int l = 0; l < k; l++) {
            T a = 0, b = 0;
            const T *u = &U[(size_t)l * M], *v = &V[(size_t)l * N];
            for (int i = 0; i < M; i++) a += cj(u[i]) * c[i];
            for (int j = 0; j < N; j++) b += cj(v[j]) * r[j];
            cross += re(a * b);
        }
        frob2 += 2 * cross + cn2 * rn2;
        U.insert(U.end(), c.begin(), c.end());
        V.insert(V.end(), r.begin(), r.end());
        k++;
        int nI = -1;
        double bc = -1;
        for (int i = 0; i < M; i++)
            if (!urow[i]) {
                double a = abs2(c[i]);
                if (a > bc) bc = a, nI = i;
            }
        const int verdict = stop.after_step(k, reqrank < 0 && std::sqrt(cn2 * rn2) <= eps * std::sqrt(std::max(frob2, 0.0)), (int64_t)k * (M + N) > (int64_t)M * N, nI < 0, confirm);
        if (verdict == 2) { result = -1; break; }
        if (verdict == 1) break;
        I = nI;
    }
    if (result == -1) return -1;
    stop.settle(k); // (the iteration ran out of rows / columns while a passed test was waiting for its confirmation)
    U.resize((size_t)k * M); // confirming terms, if any, are dropped
    V.resize((size_t)k * N);
    if (swp) { Uout.swap(V); Vout.swap(U); } // A = B^T = (U_B V_B)^T: U = V_B^T ([k][i] layout is V_B's), V = U_B^T
    else { Uout.swap(U); Vout.swap(V); }
    return k;
}

} // namespace

template <typename T>
void host_fill_blocks(const Generator &g, HMatrix &H, std::vector<T> &arena) {
    const ClusterTree &Tt = *H.tc, &Ss = *H.sc;
    const BuildParams &P = H.params;
    std::vector<BlockRec> adm, dns, done;
    build_block_tree(Tt, Ss, P, H.t_root, H.s_root, adm, dns);
    arena.clear();
    std::vector<T> U, V;
    // borrowed factors (compress_borrows: the hook's U / V stay valid until the build returns): remembered here and copied
    // ONCE, into their place in the arena, when the queue is done
    struct Borrowed { size_t block; const T *u, *v; };
    std::vector<Borrowed> borrowed;
    // low-rank queue; failures are re-split and appended to the queues
    for (size_t q = 0; q < adm.size(); q++) {
        BlockRec b = adm[q];
        const int *rows = &Tt.perm[b.t_off], *cols = &Ss.perm[b.s_off];
        int rank = -1;
        bool lent = false;
        if (P.compress) {
            const void *pu = nullptr, *pv = nullptr;
            int r = 0;
            int ok = P.compress(P.compress_ctx, b.m, b.n, rows, cols, P.epsilon, &pu, &pv, &r);
            if (ok) {
                rank = r;
                if (P.compress_borrows) {
                    lent = true;
                    borrowed.push_back({done.size(), (const T *)pu, (const T *)pv});
                } else {
                    U.assign((const T *)pu, (const T *)pu + (size_t)r * b.m);
                    V.resize((size_t)r * b.n);
                    const T *vv = (const T *)pv; // r x n column-major -> step-major [k][j]
                    for (int j = 0; j < b.n; j++)
                        for (int k = 0; k < r; k++) V[(size_t)k * b.n + j] = vv[(size_t)j * r + k];
                }
            }
        } else {
            rank = host_aca<T>(g, b.m, b.n, rows, cols, P.epsilon, aca_reqrank_argument(P.reqrank, P.aca_confirm_steps), b.t_off > b.s_off, U, V);
        }
        if (rank < 0) {
            split_failed_block(Tt, Ss, P, b, adm, dns);
            continue;
        }
        b.rank = rank;
        b.cap = rank;
        b.tmp_u = (int64_t)arena.size();
        if (lent) arena.resize(arena.size() + (size_t)rank * b.m); // (filled below)
        else arena.insert(arena.end(), U.begin(), U.begin() + (size_t)rank * b.m);
        b.tmp_v = (int64_t)arena.size();
        if (lent) arena.resize(arena.size() + (size_t)rank * b.n);
        else arena.insert(arena.end(), V.begin(), V.begin() + (size_t)rank * b.n);
        done.push_back(b);
    }
    for (const Borrowed &w : borrowed) {
        const BlockRec &b = done[w.block];
        std::copy(w.u, w.u + (size_t)b.rank * b.m, arena.begin() + b.tmp_u);
        T *vd = &arena[0] + b.tmp_v; // r x n column-major -> step-major [k][j]
        for (int j = 0; j < b.n; j++)
            for (int k = 0; k < b.rank; k++) vd[(size_t)k * b.n + j] = w.v[(size_t)j * b.rank + k];
    }
    // dense queue
    size_t base = arena.size(), total = 0;
    for (BlockRec &b : dns) {
        b.rank = -1;
        b.tmp_u = (int64_t)(base + total);
        total += (size_t)b.m * b.n;
    }
    arena.resize(base + total);
    if (P.dense_blocks && !dns.empty()) {
        std::vector<int> M, N, ro, co;
        std::vector<void *> ptrs;
        for (BlockRec &b : dns) {
            M.push_back(b.m);
            N.push_back(b.n);
            ro.push_back(b.t_off);
            co.push_back(b.s_off);
            ptrs.push_back(&arena[b.tmp_u]);
        }
        P.dense_blocks(P.dense_blocks_ctx, (int)dns.size(), M.data(), N.data(), ro.data(), co.data(), ptrs.data());
    } else {
        for (BlockRec &b : dns) g.fn(g.ctx, b.m, b.n, &Tt.perm[b.t_off], &Ss.perm[b.s_off], &arena[b.tmp_u]);
    }
    H.blocks() = done;
    H.blocks().insert(H.blocks().end(), dns.begin(), dns.end());
}

template void host_fill_blocks<double>(const Generator &, HMatrix &, std::vector<double> &);
template void host_fill_blocks<cplx>(const Generator &, HMatrix &, std::vector<cplx> &);

} // namespace hm
