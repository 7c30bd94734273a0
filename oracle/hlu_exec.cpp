// hlu_exec.cpp -- CPU checker of the hierarchical LU: executes the task lists of a plan (htool_hlu_plan_*, made by
// htool_python_amd/csrc/hlu_symbolic.cpp) with plain loops on host arrays.
//
// TEST INFRASTRUCTURE ONLY (tests/, never the product): it lets the block-recursive algorithm, its dependency levels
// and its low-rank arithmetic be checked without a GPU, and it is what the device kernels (hlu_device.hip) are compared
// with task kind by task kind.  The reference's own H-LU (htool::lu_factorization, bound at
// src/htool/hmatrix/hmatrix.hpp:58-78) lives in lib/htool, which is not in /root/reference: there are no golden
// vectors for it -- "parity unpinned"; the pin used by the tests is x = A^-1 b of the DENSE copy of the same operator
// (numpy / LAPACK), i.e. the bar of tests/test_hmatrix.py:98-128.
//
// Record layouts restated from csrc/hlu.hpp (Task 96 bytes, Leaf 48, Diag 24, Bucket 40).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

enum { T_FILL = 0, T_APPLY_DENSE = 1, T_APPLY_LR = 2, T_ADDLR = 3, T_FINAL = 4, T_DDPROD = 5, T_GETRF = 6, T_REDUCE = 7 };
enum { F_TRANS = 1, F_INPLACE = 2, F_ACCUM = 4, F_SUB = 8, F_XT = 16, F_YT = 32, F_SYM = 64, F_IDENT = 128 };
struct Task {
    int32_t type, flags, level, leaf, kref, kconst, m, n, r0, c0, a_ld, b_ld, x_ld, y_ld;
    int64_t a, b, x, y, w;
};
struct Leaf { int32_t t_off, m, s_off, n, kind, cap; int64_t u, v; int32_t diag, rank0; };
struct Diag { int32_t leaf, m; int64_t linv, uinv; };
struct Bucket { int32_t type, level; int64_t begin, end, seg_begin, seg_end; };
static_assert(sizeof(Task) == 96 && sizeof(Leaf) == 48 && sizeof(Diag) == 24 && sizeof(Bucket) == 40, "record layouts");

struct State {
    const Leaf *leaves;
    const Diag *diags;
    double *space[4];
    int64_t ld_rhs;
    int nrhs;
    int32_t *rank;
    double *norm0, *norm2;
    double eps;
    int64_t *counters; // [0] forced truncations, [1] recompressions, [2] appended columns, [3] ddprod columns
    const int64_t *aux = nullptr; // REDUCE tasks of a solve program: (reference, leading dimension) of every contribution
};

inline double *at(const State &S, int64_t ref) { return S.space[(int)(ref >> 60)] + (ref & (((int64_t)1 << 60) - 1)); }
inline int cols_of(const State &S, const Task &t) { return t.kref >= 0 ? S.rank[t.kref] : t.kref == -1 ? t.kconst : S.nrhs; }
inline int64_t ld_of(const State &S, int64_t ref, int ld) { return (ref >> 60) == 3 ? S.ld_rhs : ld; }

int keep_max(const Leaf &L) { return L.cap - std::max(4, L.cap / 8); }

// Cholesky factor with diagonal pivoting of a Gram matrix G (K x K, destroyed): rows R[j][:] (all K columns, ORIGINAL column order)
// with G = R^T R up to what is left; piv[j] = column chosen at step j.  Stops when the largest remaining diagonal entry is below
// 1e-14 of the first one (numerical rank), or -- stop2 >= 0 -- as soon as the remaining trace is at most stop2 times the trace
// (the Frobenius norm of what the chosen columns do not span), or after max_steps.  Returns the number of steps; *left = remaining trace.
int pivoted_cholesky(double *G, double *R, int *piv, int K, int ld, double stop2, int max_steps, double *trace0, double *left) {
    double dmax0 = 0, tr0 = 0;
    for (int i = 0; i < K; i++) { dmax0 = std::max(dmax0, G[i * ld + i]); tr0 += std::max(G[i * ld + i], 0.0); }
    if (trace0) *trace0 = tr0;
    int r = 0;
    double tr = tr0;
    for (; r < K && r < max_steps; r++) {
        int p = 0;
        double best = -1;
        tr = 0;
        for (int i = 0; i < K; i++) { tr += std::max(G[i * ld + i], 0.0); if (G[i * ld + i] > best) { best = G[i * ld + i]; p = i; } }
        if (!(best > 1e-14 * dmax0) || !(best > 0)) break;
        if (stop2 >= 0 && tr <= stop2 * tr0) break;
        piv[r] = p;
        const double inv = 1.0 / std::sqrt(best);
        for (int i = 0; i < K; i++) R[r * ld + i] = G[p * ld + i] * inv;
        for (int a = 0; a < K; a++) for (int b = 0; b < K; b++) G[a * ld + b] -= R[r * ld + a] * R[r * ld + b];
        G[p * ld + p] = 0.0; // (exactly: the column is used up)
        tr = 0;
        for (int i = 0; i < K; i++) tr += std::max(G[i * ld + i], 0.0);
    }
    if (left) *left = tr;
    return r;
}

// U V^T (K columns) -> the truncated form, without any SVD: rank-revealing Cholesky factorisations of small Gram matrices.
// With the columns balanced (u_a d_a, v_a / d_a: U^, V^) and G_v = R_v^T R_v factorised with diagonal pivoting (rank rv),
// V^ = Q_v R_v with Q_v = V^[:, piv_v] R_v[:, piv_v]^-1 orthonormal, so U^ V^^T = B Q_v^T with B = U^ R_v^T (m x rv) and
// |U V^T|_F = |B|_F.  The Gram matrix of B is M = R_v G_u R_v^T (rv x rv, formed from the small matrices); its Cholesky
// factorisation with diagonal pivoting M = R_B^T R_B picks columns of B one at a time, and the trace of what is left IS the squared
// Frobenius distance of B to the span of the picked columns: the factorisation stops when that is below eps^2 |B|_F^2 (r' steps).
// Then B ~ Q_B R_B with Q_B = B[:, piv_B] R_B[:, piv_B]^-1, and the leaf becomes Q_B (Q_v R_B^T)^T:
//   U' = U T_u,  T_u = D R_v^T[:, piv_B] R_B[:, piv_B]^-1  (K x r');      V' = V T_v,  T_v[piv_v, :] = D^-1 R_v[:, piv_v]^-1 R_B^T.
// A truncation by column selection instead of singular vectors: the same Frobenius bound, a rank that can be a little above the
// optimal one, a few dozen microseconds on the device instead of the millisecond of a Jacobi SVD.
void recompress(State &S, int l) {
    const Leaf &L = S.leaves[l];
    const int K = S.rank[l], m = L.m, n = L.n;
    double *U = at(S, L.u), *V = at(S, L.v);
    S.counters[1]++;
    if (K == 0) { S.norm2[l] = 0; if (S.norm0[l] < 0) S.norm0[l] = 0; return; }
    const int ld = K;
    std::vector<double> Gu((size_t)K * K), Gv((size_t)K * K), Rv((size_t)K * K), T1((size_t)K * K), M((size_t)K * K), RB((size_t)K * K), d(K);
    std::vector<int> pv(K), pb(K);
    for (int a = 0; a < K; a++)
        for (int b = a; b < K; b++) {
            double s = 0;
            for (int i = 0; i < m; i++) s += U[(int64_t)a * m + i] * U[(int64_t)b * m + i];
            Gu[a * ld + b] = Gu[b * ld + a] = s;
            s = 0;
            for (int i = 0; i < n; i++) s += V[(int64_t)a * n + i] * V[(int64_t)b * n + i];
            Gv[a * ld + b] = Gv[b * ld + a] = s;
        }
    for (int a = 0; a < K; a++) {
        const double gu = Gu[a * ld + a], gv = Gv[a * ld + a];
        d[a] = (gu > 0 && gv > 0) ? std::sqrt(std::sqrt(gv / gu)) : 0.0;
    }
    for (int a = 0; a < K; a++)
        for (int b = 0; b < K; b++) {
            const double dd = d[a] * d[b];
            Gu[a * ld + b] *= dd;
            Gv[a * ld + b] = dd > 0 ? Gv[a * ld + b] / dd : 0.0;
        }
    const int rv = pivoted_cholesky(Gv.data(), Rv.data(), pv.data(), K, ld, -1.0, K, nullptr, nullptr);
    if (rv == 0) { S.rank[l] = 0; S.norm2[l] = 0; if (S.norm0[l] < 0) S.norm0[l] = 0; return; }
    for (int a = 0; a < K; a++) // T1 = G_u R_v^T (K x rv)
        for (int b = 0; b < rv; b++) {
            double s = 0;
            for (int i = 0; i < K; i++) s += Gu[a * ld + i] * Rv[b * ld + i];
            T1[a * ld + b] = s;
        }
    for (int a = 0; a < rv; a++) // M = R_v T1 (rv x rv), symmetrised
        for (int b = a; b < rv; b++) {
            double s = 0;
            for (int i = 0; i < K; i++) s += Rv[a * ld + i] * T1[i * ld + b];
            M[a * ld + b] = M[b * ld + a] = s;
        }
    double tot = 0, left = 0;
    int newr = pivoted_cholesky(M.data(), RB.data(), pb.data(), rv, ld, S.eps * S.eps, keep_max(L), &tot, &left);
    if (newr == keep_max(L) && left > S.eps * S.eps * tot) S.counters[0]++;
    S.norm2[l] = tot - left;
    if (S.norm0[l] < 0) S.norm0[l] = tot - left;
    // T_u: row a = d_a * (R_v[piv_B, a])^T R_B11^-1, R_B11[j][c] = RB[j][pb[c]] upper triangular in pivot order
    std::vector<double> Tu((size_t)K * std::max(newr, 1), 0.0), Tv((size_t)K * std::max(newr, 1), 0.0), y(K);
    for (int a = 0; a < K; a++) {
        for (int c = 0; c < newr; c++) {
            double s = Rv[pb[c] * ld + a];
            for (int j = 0; j < c; j++) s -= y[j] * RB[j * ld + pb[c]];
            y[c] = s / RB[c * ld + pb[c]];
        }
        for (int c = 0; c < newr; c++) Tu[(size_t)a * newr + c] = d[a] * y[c];
    }
    // T_v[pv[b], c] = (1 / d) * (R_v11^-1 R_B^T)[b, c], R_v11[a][b] = Rv[a][pv[b]] upper triangular in pivot order
    for (int c = 0; c < newr; c++) {
        for (int a = 0; a < rv; a++) y[a] = RB[c * ld + a];
        for (int a = rv - 1; a >= 0; a--) {
            double s = y[a];
            for (int b = a + 1; b < rv; b++) s -= Rv[a * ld + pv[b]] * y[b];
            y[a] = s / Rv[a * ld + pv[a]];
        }
        for (int b = 0; b < rv; b++) Tv[(size_t)pv[b] * newr + c] = y[b] / d[pv[b]];
    }
    std::vector<double> row(K);
    for (int i = 0; i < m; i++) {
        for (int k = 0; k < K; k++) row[k] = U[(int64_t)k * m + i];
        for (int c = 0; c < newr; c++) { double s = 0; for (int k = 0; k < K; k++) s += row[k] * Tu[(size_t)k * newr + c]; U[(int64_t)c * m + i] = s; }
    }
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < K; k++) row[k] = V[(int64_t)k * n + i];
        for (int c = 0; c < newr; c++) { double s = 0; for (int b = 0; b < rv; b++) s += row[pv[b]] * Tv[(size_t)pv[b] * newr + c]; V[(int64_t)c * n + i] = s; }
    }
    S.rank[l] = newr;
}

void run_task(State &S, const Task &t) {
    switch (t.type) {
    case T_REDUCE: { // Y[rows of a cluster leaf] -= the private contributions, summed in the order of the list
        const int q = cols_of(S, t);
        double *y = at(S, t.y);
        const int64_t yl = ld_of(S, t.y, t.y_ld);
        for (int c = 0; c < q; c++)
            for (int i = 0; i < t.m; i++) {
                double s = 0;
                for (int k = 0; k < t.kconst; k++) s += at(S, S.aux[2 * (t.a + k)])[i + c * S.aux[2 * (t.a + k) + 1]];
                if (t.flags & F_SUB) y[i + c * yl] -= s; else y[i + c * yl] = s; // (without F_SUB: a copy back from a slot)
            }
        break;
    }
    case T_FILL: {
        const int q = cols_of(S, t);
        double *y = at(S, t.y);
        const int64_t ld = ld_of(S, t.y, t.y_ld);
        for (int c = 0; c < q; c++) for (int i = 0; i < t.m; i++) y[i + c * ld] = ((t.flags & F_IDENT) && i == c) ? 1.0 : 0.0;
        break;
    }
    case T_APPLY_DENSE: {
        const int q = cols_of(S, t);
        const double *M = at(S, t.a);
        const double *x = at(S, t.x);
        double *y = at(S, t.y);
        const int64_t xl = ld_of(S, t.x, t.x_ld), yl = ld_of(S, t.y, t.y_ld);
        const double alpha = (t.flags & F_SUB) ? -1.0 : 1.0;
        std::vector<double> xc(t.n), yc(t.m);
        for (int c = 0; c < q; c++) {
            for (int i = 0; i < t.n; i++) xc[i] = (t.flags & F_XT) ? x[i * xl + c] : x[i + c * xl];
            for (int i = 0; i < t.m; i++) {
                double s = 0;
                if (t.flags & F_TRANS) for (int j = 0; j < t.n; j++) s += M[j + (int64_t)i * t.a_ld] * xc[j];
                else for (int j = 0; j < t.n; j++) s += M[i + (int64_t)j * t.a_ld] * xc[j];
                yc[i] = alpha * s;
            }
            for (int i = 0; i < t.m; i++) {
                double &dst = (t.flags & F_YT) ? y[i * yl + c] : y[i + c * yl];
                dst = (t.flags & F_ACCUM) ? dst + yc[i] : yc[i];
            }
        }
        break;
    }
    case T_APPLY_LR: {
        const int q = cols_of(S, t), k = S.rank[t.leaf];
        const double *A = at(S, t.a), *B = at(S, t.b), *x = at(S, t.x);
        double *y = at(S, t.y);
        const int64_t xl = ld_of(S, t.x, t.x_ld), yl = ld_of(S, t.y, t.y_ld);
        const double alpha = (t.flags & F_SUB) ? -1.0 : 1.0;
        std::vector<double> w(k);
        for (int c = 0; c < q; c++) {
            for (int l = 0; l < k; l++) {
                double s = 0;
                for (int j = 0; j < t.n; j++) s += B[j + (int64_t)l * t.b_ld] * x[j + c * xl];
                w[l] = s;
            }
            for (int i = 0; i < t.m; i++) {
                double s = 0;
                for (int l = 0; l < k; l++) s += A[i + (int64_t)l * t.a_ld] * w[l];
                y[i + c * yl] = (t.flags & F_ACCUM) ? y[i + c * yl] + alpha * s : alpha * s;
            }
        }
        break;
    }
    case T_ADDLR: {
        const Leaf &L = S.leaves[t.leaf];
        const int k = cols_of(S, t);
        const double *X = at(S, t.x), *Z = at(S, t.y);
        const double alpha = (t.flags & F_SUB) ? -1.0 : 1.0;
        auto xe = [&](int i, int c) { return (t.flags & F_XT) ? X[(int64_t)i * t.x_ld + c] : X[i + (int64_t)c * t.x_ld]; };
        auto ze = [&](int j, int c) { return (t.flags & F_YT) ? Z[(int64_t)j * t.y_ld + c] : Z[j + (int64_t)c * t.y_ld]; };
        if (L.kind == 0) {
            double *D = at(S, L.u);
            for (int j = 0; j < t.n; j++)
                for (int i = 0; i < t.m; i++) {
                    double s = 0;
                    for (int c = 0; c < k; c++) s += xe(i, c) * ze(j, c);
                    D[(t.r0 + i) + (int64_t)(t.c0 + j) * L.m] += alpha * s;
                }
            break;
        }
        double *U = at(S, L.u), *V = at(S, L.v);
        for (int c = 0; c < k; c++) {
            if (S.rank[t.leaf] == L.cap) recompress(S, t.leaf);
            const int f = S.rank[t.leaf];
            for (int i = 0; i < L.m; i++) U[(int64_t)f * L.m + i] = 0.0;
            for (int j = 0; j < L.n; j++) V[(int64_t)f * L.n + j] = 0.0;
            for (int i = 0; i < t.m; i++) U[(int64_t)f * L.m + t.r0 + i] = alpha * xe(i, c);
            for (int j = 0; j < t.n; j++) V[(int64_t)f * L.n + t.c0 + j] = ze(j, c);
            S.rank[t.leaf] = f + 1;
            S.counters[2]++;
        }
        break;
    }
    case T_FINAL: recompress(S, t.leaf); break;
    case T_DDPROD: {
        const int m = t.m, n = t.n, q = t.r0;
        const double *A = at(S, t.a), *B = at(S, t.b);
        double *W = at(S, t.w), *X = at(S, t.x), *Z = at(S, t.y);
        for (int j = 0; j < n; j++)
            for (int i = 0; i < m; i++) {
                double s = 0;
                if (t.flags & F_TRANS) for (int l = 0; l < q; l++) s += A[i + (int64_t)l * t.a_ld] * B[j + (int64_t)l * t.b_ld]; // (b is n x q: a b^T)
                else for (int l = 0; l < q; l++) s += A[i + (int64_t)l * t.a_ld] * B[l + (int64_t)j * t.b_ld];
                W[i + (int64_t)j * m] = s;
            }
        // cross approximation with full pivoting on the explicit residual, until its Frobenius norm is below the tolerance
        const double tol2 = 0.01 * S.eps * S.eps * std::max(S.norm0[t.leaf], 0.0);
        int k = 0;
        while (k < t.kconst) {
            double fro = 0, best = 0;
            int bi = 0, bj = 0;
            for (int j = 0; j < n; j++)
                for (int i = 0; i < m; i++) {
                    const double v = W[i + (int64_t)j * m];
                    fro += v * v;
                    if (std::fabs(v) > best) { best = std::fabs(v); bi = i; bj = j; }
                }
            if (fro <= tol2 || best == 0.0) break;
            const double piv = W[bi + (int64_t)bj * m];
            for (int i = 0; i < m; i++) X[i + (int64_t)k * t.x_ld] = W[i + (int64_t)bj * m] / piv;
            for (int j = 0; j < n; j++) Z[j + (int64_t)k * t.y_ld] = W[bi + (int64_t)j * m];
            for (int j = 0; j < n; j++) {
                const double zj = Z[j + (int64_t)k * t.y_ld];
                for (int i = 0; i < m; i++) W[i + (int64_t)j * m] -= X[i + (int64_t)k * t.x_ld] * zj;
            }
            k++;
        }
        S.rank[t.kref] = k;
        S.counters[3] += k;
        break;
    }
    case T_GETRF: {
        const Leaf &L = S.leaves[t.leaf];
        const Diag &Dg = S.diags[L.diag];
        const int m = L.m;
        double *A = S.space[0] + L.u, *Li = S.space[1] + Dg.linv, *Ui = S.space[1] + Dg.uinv;
        std::vector<int> piv(m);
        for (int j = 0; j < m; j++) {
            int p = j;
            if (!(t.flags & F_SYM)) { for (int i = j + 1; i < m; i++) if (std::fabs(A[i + (int64_t)j * m]) > std::fabs(A[p + (int64_t)j * m])) p = i; }
            else if (!(A[j + (int64_t)j * m] > 0)) S.counters[4]++; // (a symmetric positive definite leaf needs no pivoting: L U with U = D L^T)
            piv[j] = p;
            if (p != j) for (int c = 0; c < m; c++) std::swap(A[j + (int64_t)c * m], A[p + (int64_t)c * m]);
            const double d = A[j + (int64_t)j * m];
            for (int i = j + 1; i < m; i++) A[i + (int64_t)j * m] /= d;
            for (int c = j + 1; c < m; c++) {
                const double u = A[j + (int64_t)c * m];
                for (int i = j + 1; i < m; i++) A[i + (int64_t)c * m] -= A[i + (int64_t)j * m] * u;
            }
        }
        // (P^T L)^-1 = L^-1 P: column c of the result solves L y = P e_c;  U^-1 column by column
        for (int c = 0; c < m; c++) {
            std::vector<double> y(m, 0.0);
            y[c] = 1.0;
            for (int j = 0; j < m; j++) if (piv[j] != j) std::swap(y[j], y[piv[j]]);
            for (int j = 0; j < m; j++) for (int i = j + 1; i < m; i++) y[i] -= A[i + (int64_t)j * m] * y[j];
            for (int i = 0; i < m; i++) Li[i + (int64_t)c * m] = y[i];
            std::vector<double> z(m, 0.0);
            z[c] = 1.0;
            for (int j = m - 1; j >= 0; j--) {
                z[j] /= A[j + (int64_t)j * m];
                for (int i = 0; i < j; i++) z[i] -= A[i + (int64_t)j * m] * z[j];
            }
            for (int i = 0; i < m; i++) Ui[i + (int64_t)c * m] = z[i];
        }
        if (t.flags & F_SYM) { // the inverse of the CHOLESKY factor L_c = L D^(1/2): rows of L^-1 scaled by 1 / sqrt(d), and its transpose
            for (int c = 0; c < m; c++) for (int i = 0; i < m; i++) Li[i + (int64_t)c * m] /= std::sqrt(A[i + (int64_t)i * m]);
            for (int c = 0; c < m; c++) for (int i = 0; i < m; i++) Ui[i + (int64_t)c * m] = Li[c + (int64_t)i * m];
        }
        break;
    }
    }
}

} // namespace

extern "C" {

// Runs one program (a window of the factorisation, or a solve) bucket by bucket.  shuffle != 0: the independent work
// items of a bucket (tasks, or runs of one target) are executed in a pseudo-random order -- the result must not change.
int hluo_run(const void *tasks_, int64_t n_tasks, const void *buckets_, int64_t n_buckets, const int64_t *seg, const void *leaves, const void *diags,
             double *factor, double *diag, double *scratch, double *rhs, int64_t ld_rhs, int nrhs, int32_t *rank, double *norm0, double *norm2, double eps,
             int64_t *counters, uint64_t shuffle, const int64_t *aux) {
    const Task *tasks = (const Task *)tasks_;
    const Bucket *buckets = (const Bucket *)buckets_;
    State S;
    S.leaves = (const Leaf *)leaves; S.diags = (const Diag *)diags;
    S.space[0] = factor; S.space[1] = diag; S.space[2] = scratch; S.space[3] = rhs;
    S.aux = aux;
    S.ld_rhs = ld_rhs; S.nrhs = nrhs; S.rank = rank; S.norm0 = norm0; S.norm2 = norm2; S.eps = eps; S.counters = counters;
    uint64_t rng = shuffle * 0x9E3779B97F4A7C15ull + 12345;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    for (int64_t b = 0; b < n_buckets; b++) {
        const Bucket &B = buckets[b];
        std::vector<std::pair<int64_t, int64_t>> items; // [first, last) task ranges that must stay in order
        if (B.type == T_ADDLR || B.type == T_FINAL) for (int64_t s = B.seg_begin; s < B.seg_end; s++) items.push_back({seg[s], seg[s + 1]});
        else for (int64_t i = B.begin; i < B.end; i++) items.push_back({i, i + 1});
        if (shuffle) for (size_t i = items.size(); i > 1; i--) std::swap(items[i - 1], items[next() % i]);
        for (auto &it : items) for (int64_t i = it.first; i < it.second; i++) run_task(S, tasks[i]);
    }
    (void)n_tasks;
    return 0;
}

} // extern "C"
