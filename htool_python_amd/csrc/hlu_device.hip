// hlu_device.hip -- hierarchical LU on the device: the executor of the plan of hlu_symbolic.cpp (see hlu.hpp).
//
// Stands where the reference calls htool::lu_factorization / lu_solve / cholesky_factorization / cholesky_solve
// (bound at src/htool/hmatrix/hmatrix.hpp:58-94).  The leaves of the operator are unpacked from the product's tile
// panels into one FACTOR ARENA (dense leaf: m x n column-major; low-rank leaf: U m x cap and V n x cap, `cap` columns
// of room), the plan's windows are uploaded one after the other and every (level, kind) bucket of a window is ONE
// launch: a workgroup per task -- or per run of updates of one target leaf, executed in plan order, so that nothing is
// accumulated by atomics and the factors are bitwise reproducible.  Kernels (fp64, HBM / latency bound: the leaves are
// small, the arithmetic intensity is that of thin products):
//   hlu_fill_kernel         zero a scratch block
//   hlu_apply_dense_kernel  Y (+)= -+ op(M) X, M a dense leaf or the inverse factor of a diagonal leaf (in place for the latter)
//   hlu_apply_lr_kernel     Y (+)= -+ A (B^T X), (A, B) the rows of the two factors of a low-rank leaf restricted to a sub-block
//   hlu_update_kernel       the updates of ONE target leaf: dense D -= X Z^T; low rank: columns appended, the leaf re-truncated
//                           when its room is used up (and by FINAL tasks): Gram matrices of both factors on the fp64 matrix
//                           cores, two Cholesky factorisations with diagonal pivoting of <= 64 x 64 matrices in LDS (a
//                           truncation by column selection with an exact Frobenius bound, no SVD), two K x r transforms
//                           applied row by row with the new columns in registers
//   hlu_ddprod_kernel       product of two dense leaves that lands in a low-rank leaf: formed in a work block, compressed by
//                           cross approximation with full pivoting on the explicit residual, handed on as X' Z'^T
//   hlu_getrf_[lds_]kernel  LU with partial pivoting of a diagonal leaf + the explicit inverses (P^T L)^-1 and U^-1, with
//                           which every triangular solve against a diagonal leaf is a product (leaves of at most 128 rows: in LDS)
//   hlu_reduce_kernel       (solves) the private contributions of the leaves of a block step summed per cluster leaf, in plan order
// After the factorisation the plan's `invert` program forms explicit inverse factors of the diagonal blocks of up to 1024 rows
// (the block's own sweep on the identity), and the leaves move to a tight arena.  lu_solve replays the plan's solve program
// (forward and backward sweep) on the caller's block of right-hand sides, eight columns at a time.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>

#include "capi_internal.hpp"
#include "device_internal.hpp"
#include "hlu.hpp"

using namespace hm;
using namespace hm::hlu;

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int QC = 8;           // right-hand-side columns per pass of the apply kernels
constexpr int HLU_MAX_DIM = 1024; // rows / columns of a dense leaf the apply kernels stage in LDS

struct Ctx {
    double *space[4];
    long long ld_rhs;
    int nrhs;
    const Leaf *leaves;
    const Diag *diags;
    int *rank;
    double *norm0, *norm2;
    double eps;
    long long *counters; // [0] forced truncations, [1] truncations, [2] appended columns, [3] columns out of dense products, [4] zero pivots
};

__device__ __forceinline__ double *at(const Ctx &c, long long ref) { return c.space[(int)(ref >> SPACE_SHIFT)] + (ref & (((long long)1 << SPACE_SHIFT) - 1)); }
__device__ __forceinline__ int cols_of(const Ctx &c, const Task &t) { return t.kref >= 0 ? c.rank[t.kref] : t.kref == -1 ? t.kconst : c.nrhs; }
__device__ __forceinline__ long long ld_of(const Ctx &c, long long ref, int ld) { return (int)(ref >> SPACE_SHIFT) == SP_RHS ? c.ld_rhs : (long long)ld; }

__global__ __launch_bounds__(256) void hlu_fill_kernel(Ctx c, const Task *tasks) {
    const Task t = tasks[blockIdx.x];
    const int q = cols_of(c, t);
    double *y = at(c, t.y);
    const long long ld = ld_of(c, t.y, t.y_ld);
    const bool ident = t.flags & F_IDENT;
    for (long long e = threadIdx.x; e < (long long)q * t.m; e += 256) {
        const int col = (int)(e / t.m), i = (int)(e - (long long)col * t.m);
        y[i + col * ld] = (ident && i == col) ? 1.0 : 0.0;
    }
}

__device__ __forceinline__ void apply_dense_body(const Ctx &c, const Task &t, double *sm /* n x QC */) {
    const int tid = threadIdx.x, q = cols_of(c, t);
    const double *M = at(c, t.a), *x = at(c, t.x);
    double *y = at(c, t.y);
    const long long xl = ld_of(c, t.x, t.x_ld), yl = ld_of(c, t.y, t.y_ld);
    const bool tr = t.flags & F_TRANS, xt = t.flags & F_XT, yt = t.flags & F_YT, acc = t.flags & F_ACCUM;
    const double alpha = (t.flags & F_SUB) ? -1.0 : 1.0;
    for (int c0 = 0; c0 < q; c0 += QC) {
        const int qc = min(QC, q - c0);
        __syncthreads();
        for (int e = tid; e < t.n * QC; e += 256) {
            int i, col;
            if (xt) { col = e % QC; i = e / QC; } else { i = e % t.n; col = e / t.n; }
            sm[i * QC + col] = col < qc ? (xt ? x[(long long)i * xl + c0 + col] : x[i + (long long)(c0 + col) * xl]) : 0.0;
        }
        __syncthreads();
        for (int i = tid; i < t.m; i += 256) {
            double a[QC];
#pragma unroll
            for (int col = 0; col < QC; col++) a[col] = 0.0;
            // (eight entries of the row of M are loaded before they are used: one load in flight per trip made the loop wait a memory
            // round trip per entry -- 45 us for a 187 x 187 leaf, most of the time of a solve)
            const long long ms = tr ? 1 : (long long)t.a_ld;
            const double *Mi = tr ? M + (long long)i * t.a_ld : M + i;
            int j = 0;
            for (; j + 8 <= t.n; j += 8) {
                double mv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) mv[u] = Mi[(long long)(j + u) * ms];
#pragma unroll
                for (int u = 0; u < 8; u++)
#pragma unroll
                    for (int col = 0; col < QC; col++) a[col] = fma(mv[u], sm[(j + u) * QC + col], a[col]);
            }
            for (; j < t.n; j++) {
                const double mv = Mi[(long long)j * ms];
#pragma unroll
                for (int col = 0; col < QC; col++) a[col] = fma(mv, sm[j * QC + col], a[col]);
            }
#pragma unroll
            for (int col = 0; col < QC; col++)
                if (col < qc) {
                    double *dst = yt ? y + (long long)i * yl + c0 + col : y + i + (long long)(c0 + col) * yl;
                    *dst = acc ? *dst + alpha * a[col] : alpha * a[col];
                }
        }
    }
}

__global__ __launch_bounds__(256) void hlu_apply_dense_kernel(Ctx c, const Task *tasks) {
    extern __shared__ double sm[];
    const Task t = tasks[blockIdx.x];
    apply_dense_body(c, t, sm);
}

__device__ __forceinline__ void apply_lr_body(const Ctx &c, const Task &t, double *W /* 64 x QC */, double (*red)[64] /* 4 x 64 */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = cols_of(c, t), k = c.rank[t.leaf];
    const double *A = at(c, t.a), *B = at(c, t.b), *x = at(c, t.x);
    double *y = at(c, t.y);
    const long long xl = ld_of(c, t.x, t.x_ld), yl = ld_of(c, t.y, t.y_ld);
    const bool acc = t.flags & F_ACCUM;
    const double alpha = (t.flags & F_SUB) ? -1.0 : 1.0;
    if (k == 0) {
        if (!acc) for (long long e = tid; e < (long long)q * t.m; e += 256) { const int col = (int)(e / t.m), i = (int)(e - (long long)col * t.m); y[i + col * yl] = 0.0; }
        return;
    }
    for (int c0 = 0; c0 < q; c0 += QC) {
        const int qc = min(QC, q - c0);
        for (int l0 = 0; l0 < k; l0 += 8) { // W[l0 .. l0 + 8) = B^T X
            double a[8][QC];
#pragma unroll
            for (int l = 0; l < 8; l++)
#pragma unroll
                for (int col = 0; col < QC; col++) a[l][col] = 0.0;
            for (int j = tid; j < t.n; j += 256) {
                double xv[QC];
#pragma unroll
                for (int col = 0; col < QC; col++) xv[col] = col < qc ? x[j + (long long)(c0 + col) * xl] : 0.0;
#pragma unroll
                for (int l = 0; l < 8; l++) {
                    const double b = l0 + l < k ? B[j + (long long)(l0 + l) * t.b_ld] : 0.0;
#pragma unroll
                    for (int col = 0; col < QC; col++) a[l][col] = fma(b, xv[col], a[l][col]);
                }
            }
#pragma unroll
            for (int l = 0; l < 8; l++)
#pragma unroll
                for (int col = 0; col < QC; col++) {
                    double v = a[l][col];
                    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
                    if (lane == 0) red[wave][l * 8 + col] = v;
                }
            __syncthreads();
            if (tid < 64 && l0 + tid / 8 < 64) W[(l0 + tid / 8) * QC + (tid & 7)] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            __syncthreads();
        }
        for (int i = tid; i < t.m; i += 256) {
            double s[QC];
#pragma unroll
            for (int col = 0; col < QC; col++) s[col] = 0.0;
            int l = 0;
            for (; l + 8 <= k; l += 8) { // (eight loads in flight, as above)
                double av[8];
#pragma unroll
                for (int u = 0; u < 8; u++) av[u] = A[i + (long long)(l + u) * t.a_ld];
#pragma unroll
                for (int u = 0; u < 8; u++)
#pragma unroll
                    for (int col = 0; col < QC; col++) s[col] = fma(av[u], W[(l + u) * QC + col], s[col]);
            }
            for (; l < k; l++) {
                const double av = A[i + (long long)l * t.a_ld];
#pragma unroll
                for (int col = 0; col < QC; col++) s[col] = fma(av, W[l * QC + col], s[col]);
            }
#pragma unroll
            for (int col = 0; col < QC; col++)
                if (col < qc) {
                    double *dst = y + i + (long long)(c0 + col) * yl;
                    *dst = acc ? *dst + alpha * s[col] : alpha * s[col];
                }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void hlu_apply_lr_kernel(Ctx c, const Task *tasks) {
    __shared__ double W[64 * QC];
    __shared__ double red[4][64];
    const Task t = tasks[blockIdx.x];
    apply_lr_body(c, t, W, red);
}
// the dense and the low-rank applications of ONE level in one launch (a solve is thousands of small levels: every launch counts)
__global__ __launch_bounds__(256) void hlu_apply_kernel(Ctx c, const Task *tasks) {
    extern __shared__ double sm[];
    __shared__ double W[64 * QC];
    __shared__ double red[4][64];
    const Task t = tasks[blockIdx.x];
    if (t.type == T_APPLY_DENSE) apply_dense_body(c, t, sm);
    else apply_lr_body(c, t, W, red);
}

// (solve programs) Y[rows of one cluster leaf] -= the private contributions of the leaves of a block step, summed in the order of the list
__global__ __launch_bounds__(256) void hlu_reduce_kernel(Ctx c, const Task *tasks, const long long *aux) {
    const Task t = tasks[blockIdx.x];
    const int q = cols_of(c, t);
    double *y = at(c, t.y);
    const long long yl = ld_of(c, t.y, t.y_ld);
    const long long *list = aux + 2 * t.a;
    for (int e = threadIdx.x; e < t.m * q; e += 256) {
        const int i = e % t.m, col = e / t.m;
        double s = 0;
        for (int k = 0; k < t.kconst; k++) s += at(c, list[2 * k])[i + (long long)col * list[2 * k + 1]];
        if (t.flags & F_SUB) y[i + col * yl] -= s; else y[i + col * yl] = s; // (without F_SUB: a copy back from a slot)
    }
}

__device__ __forceinline__ int keep_max(const Leaf &L) { return L.cap - max(4, L.cap / 8); }

// Cholesky factor with diagonal pivoting of a Gram matrix G (K x K in LDS, K <= 64, destroyed): rows R[j][0..K) in the ORIGINAL column
// order, piv[j] the column chosen at step j; right-looking (every step downdates the whole matrix: no swaps).  Stops at the numerical
// rank, or -- stop2 >= 0 -- when the remaining trace is at most stop2 times the trace, or after max_steps.  Returns the steps made;
// tr0 / left: the trace before / after (the same for every thread).  s_red: 2 doubles + 1 int of LDS.
__device__ int pchol_lds(double *G, double *R, int *piv, int K, int KS, double stop2, int max_steps, double *s_red, int *s_p, double *tr0_out, double *left_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double dmax0 = 0, tr0 = 0, tr = 0;
    int r = 0;
    for (;; r++) {
        if (wave == 0) {
            double v = lane < K ? G[lane * KS + lane] : -1.0;
            double t = lane < K ? fmax(v, 0.0) : 0.0;
            int idx = lane;
            for (int dd = 32; dd > 0; dd >>= 1) {
                const double ov = __shfl_xor(v, dd);
                const int oi = __shfl_xor(idx, dd);
                t += __shfl_xor(t, dd);
                if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
            }
            if (lane == 0) { s_red[0] = v; s_red[1] = t; *s_p = idx; }
        }
        __syncthreads();
        const double best = s_red[0];
        const int p = *s_p;
        tr = s_red[1];
        if (r == 0) { dmax0 = best; tr0 = tr; }
        if (r >= K || r >= max_steps || !(best > 1e-14 * dmax0) || !(best > 0)) break;
        if (stop2 >= 0 && tr <= stop2 * tr0) break;
        const double inv = 1.0 / sqrt(best);
        if (tid < K) R[r * KS + tid] = G[p * KS + tid] * inv;
        if (tid == 0) piv[r] = p;
        __syncthreads();
        for (int e = tid; e < K * K; e += 256) { const int a = e / K, b = e - a * K; G[a * KS + b] -= R[r * KS + a] * R[r * KS + b]; }
        __syncthreads();
        if (tid == 0) G[p * KS + p] = 0.0;
        __syncthreads();
    }
    __syncthreads();
    *tr0_out = tr0;
    *left_out = tr;
    return r;
}

// U V^T with K = rank[l] columns -> the truncated form, without an SVD (the algorithm is spelt out in oracle/hlu_exec.cpp: recompress):
// Gram matrices, two Cholesky factorisations with diagonal pivoting of K x K matrices in LDS, two K x r transforms applied row by row.
// The whole workgroup.  sm: 4 matrices of Kc x (Kc + 1) doubles, then vectors.
__device__ void recompress(const Ctx &c, int l, const Leaf &L, double *sm, int Kc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = c.rank[l], m = L.m, n = L.n, KS = Kc + 1;
    double *U = at(c, L.u), *V = at(c, L.v);
    double *Gu = sm, *Gv = Gu + Kc * KS, *Rv = Gv + Kc * KS, *Mm = Rv + Kc * KS;
    double *T1 = Gv, *RB = Gv, *Tu = Gu, *Tv = Mm; // (what lives where once its predecessor is dead)
    double *d = Mm + Kc * KS, *s_red = d + Kc;
    int *pv = (int *)(s_red + 2), *pb = pv + Kc, *s_p = pb + Kc;
    if (tid == 0) atomicAdd((unsigned long long *)&c.counters[1], 1ull);
    if (K == 0) {
        if (tid == 0) { c.norm2[l] = 0.0; if (c.norm0[l] < 0) c.norm0[l] = 0.0; }
        __syncthreads();
        return;
    }
    const long long tk0 = wall_clock64();
    // Gram matrices on the fp64 matrix cores: a wave takes a 16 x 16 block of column pairs; v_mfma_f64_16x16x4_f64 sums over four rows
    // per instruction (A[i][k] = X[row k][column ca 16 + i], B[k][j] = X[row k][column cb 16 + j]: one load per lane and operand)
    {
        const int nch = (K + 15) / 16, npair = nch * (nch + 1) / 2;
        const int li = lane & 15, lk = lane >> 4;
        for (int item = wave; item < 2 * npair; item += 4) {
            const bool vside = item >= npair;
            int q = vside ? item - npair : item, ca = 0;
            while (q >= nch - ca) { q -= nch - ca; ca++; }
            const int cb = ca + q;
            const double *X = vside ? V : U;
            const int len = vside ? n : m;
            const bool a_ok = ca * 16 + li < K, b_ok = cb * 16 + li < K;
            const double *xa = X + (long long)(ca * 16 + li) * len, *xb = X + (long long)(cb * 16 + li) * len;
            v4f64 acc = {0.0, 0.0, 0.0, 0.0};
            for (int r0 = 0; r0 < len; r0 += 16) { // four instructions per trip: rows r0 + 4 u + lk
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int row = r0 + 4 * u + lk;
                    const double av = (a_ok && row < len) ? xa[row] : 0.0, bv = (b_ok && row < len) ? xb[row] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
            }
            double *G = vside ? Gv : Gu;
#pragma unroll
            for (int t4 = 0; t4 < 4; t4++) { // D[(lane >> 4) + 4 t][lane & 15]
                const int ga = ca * 16 + lk + 4 * t4, gb = cb * 16 + li;
                if (ga < K && gb < K) { G[ga * KS + gb] = acc[t4]; if (ca != cb) G[gb * KS + ga] = acc[t4]; }
            }
        }
    }
    __syncthreads();
    if (tid < K) {
        const double gu = Gu[tid * KS + tid], gv = Gv[tid * KS + tid];
        d[tid] = (gu > 0 && gv > 0) ? sqrt(sqrt(gv / gu)) : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < K * K; e += 256) {
        const int a = e / K, b = e - a * K;
        const double dd = d[a] * d[b];
        Gu[a * KS + b] *= dd;
        Gv[a * KS + b] = dd > 0 ? Gv[a * KS + b] / dd : 0.0;
    }
    __syncthreads();
    const long long tk1 = wall_clock64();
    double tot = 0, left = 0;
    const int rv = pchol_lds(Gv, Rv, pv, K, KS, -1.0, K, s_red, s_p, &tot, &left);
    if (rv == 0) {
        if (tid == 0) { c.rank[l] = 0; c.norm2[l] = 0.0; if (c.norm0[l] < 0) c.norm0[l] = 0.0; }
        __syncthreads();
        return;
    }
    for (int e = tid; e < K * rv; e += 256) { // T1 = G_u R_v^T (K x rv), over the dead G_v
        const int a = e / rv, b = e - a * rv;
        double s = 0;
        for (int i = 0; i < K; i++) s = fma(Gu[a * KS + i], Rv[b * KS + i], s);
        T1[a * KS + b] = s;
    }
    __syncthreads();
    for (int e = tid; e < rv * rv; e += 256) { // M = R_v T1 (rv x rv), symmetric by construction of the upper half
        const int a = e / rv, b = e - a * rv;
        if (a > b) continue;
        double s = 0;
        for (int i = 0; i < K; i++) s = fma(Rv[a * KS + i], T1[i * KS + b], s);
        Mm[a * KS + b] = s;
        Mm[b * KS + a] = s;
    }
    __syncthreads();
    const long long tk2 = wall_clock64();
    const int kmax = keep_max(L);
    const int newr = pchol_lds(Mm, RB, pb, rv, KS, c.eps * c.eps, kmax, s_red, s_p, &tot, &left);
    if (tid == 0) {
        if (newr == kmax && left > c.eps * c.eps * tot) atomicAdd((unsigned long long *)&c.counters[0], 1ull);
        c.norm2[l] = tot - left;
        if (c.norm0[l] < 0) c.norm0[l] = tot - left;
    }
    const long long tk3 = wall_clock64();
    if (tid < K) { // T_u, row tid: d * (R_v[piv_B, tid])^T R_B[:, piv_B]^-1 (upper triangular in pivot order), over the dead G_u
        for (int q = 0; q < newr; q++) {
            double s = Rv[pb[q] * KS + tid];
            for (int j = 0; j < q; j++) s -= Tu[tid * KS + j] * RB[j * KS + pb[q]];
            Tu[tid * KS + q] = s / RB[q * KS + pb[q]];
        }
        const double da = d[tid];
        for (int q = 0; q < newr; q++) Tu[tid * KS + q] *= da;
    }
    __syncthreads(); // (M is dead: the factorisation has consumed it)
    if (tid < newr) { // T_v, column tid, rows in the pivot order of R_v: R_v[:, piv_v]^-1 (row tid of R_B)^T / d, over the dead M
        for (int a = rv - 1; a >= 0; a--) {
            double s = RB[tid * KS + a];
            for (int b = a + 1; b < rv; b++) s -= Rv[a * KS + pv[b]] * Tv[b * KS + tid];
            Tv[a * KS + tid] = s / Rv[a * KS + pv[a]];
        }
        for (int b = 0; b < rv; b++) Tv[b * KS + tid] /= d[pv[b]];
    }
    __syncthreads();
    const long long tk4 = wall_clock64();
    // the transforms applied row by row: 64 accumulators in registers (eight at a time are skipped beyond the new rank), the
    // coefficients read from LDS by every lane at once
    for (int i = tid; i < m; i += 256) {
        double acc[64];
#pragma unroll
        for (int q = 0; q < 64; q++) acc[q] = 0.0;
        for (int k = 0; k < K; k++) {
            const double x = U[(long long)k * m + i];
            const double *tr = Tu + k * KS;
#pragma unroll
            for (int q0 = 0; q0 < 64; q0 += 8)
                if (q0 < newr) {
#pragma unroll
                    for (int q = q0; q < q0 + 8; q++) acc[q] = fma(x, tr[q], acc[q]);
                }
        }
#pragma unroll
        for (int q = 0; q < 64; q++) if (q < newr) U[(long long)q * m + i] = acc[q];
    }
    for (int i = tid; i < n; i += 256) {
        double acc[64];
#pragma unroll
        for (int q = 0; q < 64; q++) acc[q] = 0.0;
        for (int b = 0; b < rv; b++) {
            const double x = V[(long long)pv[b] * n + i];
            const double *tr = Tv + b * KS;
#pragma unroll
            for (int q0 = 0; q0 < 64; q0 += 8)
                if (q0 < newr) {
#pragma unroll
                    for (int q = q0; q < q0 + 8; q++) acc[q] = fma(x, tr[q], acc[q]);
                }
        }
#pragma unroll
        for (int q = 0; q < 64; q++) if (q < newr) V[(long long)q * n + i] = acc[q];
    }
    __syncthreads();
    if (tid == 0) {
        c.rank[l] = newr;
        const long long tk5 = wall_clock64(); // (phase clocks of the truncation, 10 ns ticks: Gram, first Cholesky + products, second Cholesky, transforms, application)
        atomicAdd((unsigned long long *)&c.counters[5], (unsigned long long)(((tk1 - tk0) << 32) | (unsigned)(tk2 - tk1)));
        atomicAdd((unsigned long long *)&c.counters[6], (unsigned long long)(((tk3 - tk2) << 32) | (unsigned)(tk4 - tk3)));
        atomicAdd((unsigned long long *)&c.counters[7], (unsigned long long)(tk5 - tk4));
    }
    __syncthreads();
}

// C (m x n column-major, leading dimension ldc) += alpha X Z^T with X m x k, Z n x k given by row / column strides; the whole workgroup:
// 64 x 64 tiles of C, 16 columns of X and Z staged in LDS per step (lds: 2 x 64 x 17 doubles), 4 x 4 entries per thread.
__device__ void wg_gemm_nt(double *C, long long ldc, int m, int n, int k, double alpha, const double *X, long long xrs, long long xcs, const double *Z, long long zrs, long long zcs,
                           double *lds) {
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    double *Xs = lds, *Zs = lds + 64 * 17;
    for (int j0 = 0; j0 < n; j0 += 64)
        for (int i0 = 0; i0 < m; i0 += 64) {
            double acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
            for (int q0 = 0; q0 < k; q0 += 16) {
                __syncthreads();
                for (int e = tid; e < 64 * 16; e += 256) {
                    int r, q;
                    if (xrs == 1) { r = e & 63; q = e >> 6; } else { q = e & 15; r = e >> 4; }
                    Xs[r * 17 + q] = (i0 + r < m && q0 + q < k) ? X[(long long)(i0 + r) * xrs + (long long)(q0 + q) * xcs] : 0.0;
                    if (zrs == 1) { r = e & 63; q = e >> 6; } else { q = e & 15; r = e >> 4; }
                    Zs[r * 17 + q] = (j0 + r < n && q0 + q < k) ? Z[(long long)(j0 + r) * zrs + (long long)(q0 + q) * zcs] : 0.0;
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    double xv[4], zv[4];
#pragma unroll
                    for (int a = 0; a < 4; a++) { xv[a] = Xs[(tx + 16 * a) * 17 + q]; zv[a] = Zs[(ty + 16 * a) * 17 + q]; }
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++) acc[a][b] = fma(xv[a], zv[b], acc[a][b]);
                }
            }
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int i = i0 + tx + 16 * a, j = j0 + ty + 16 * b;
                    if (i < m && j < n) C[i + (long long)j * ldc] += alpha * acc[a][b];
                }
        }
    __syncthreads();
}

__global__ __launch_bounds__(256) void hlu_update_kernel(Ctx c, const Task *tasks, const long long *seg, int Kc) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x;
    const long long first = seg[blockIdx.x], last = seg[blockIdx.x + 1];
    for (long long ti = first; ti < last; ti++) {
        const Task t = tasks[ti];
        const Leaf L = c.leaves[t.leaf];
        if (t.type == T_FINAL) { recompress(c, t.leaf, L, sm, Kc); continue; }
        const int k = cols_of(c, t);
        const double *X = at(c, t.x), *Z = at(c, t.y);
        const bool xt = t.flags & F_XT, zt = t.flags & F_YT;
        const double alpha = (t.flags & F_SUB) ? -1.0 : 1.0;
        if (L.kind == 0) {
            double *D = at(c, L.u) + t.r0 + (long long)t.c0 * L.m;
            wg_gemm_nt(D, L.m, t.m, t.n, k, alpha, X, xt ? t.x_ld : 1, xt ? 1 : t.x_ld, Z, zt ? t.y_ld : 1, zt ? 1 : t.y_ld, sm);
            continue;
        }
        double *U = at(c, L.u), *V = at(c, L.v);
        int done = 0;
        while (done < k) {
            int fill = c.rank[t.leaf];
            if (fill >= L.cap) { recompress(c, t.leaf, L, sm, Kc); fill = c.rank[t.leaf]; }
            const int take = min(k - done, L.cap - fill);
            for (long long e = tid; e < (long long)take * L.m; e += 256) {
                const int col = (int)(e / L.m), i = (int)(e - (long long)col * L.m), r = i - t.r0, q = done + col;
                U[(long long)(fill + col) * L.m + i] = (r >= 0 && r < t.m) ? alpha * (xt ? X[(long long)r * t.x_ld + q] : X[r + (long long)q * t.x_ld]) : 0.0;
            }
            for (long long e = tid; e < (long long)take * L.n; e += 256) {
                const int col = (int)(e / L.n), j = (int)(e - (long long)col * L.n), r = j - t.c0, q = done + col;
                V[(long long)(fill + col) * L.n + j] = (r >= 0 && r < t.n) ? (zt ? Z[(long long)r * t.y_ld + q] : Z[r + (long long)q * t.y_ld]) : 0.0;
            }
            __syncthreads();
            if (tid == 0) { c.rank[t.leaf] = fill + take; atomicAdd((unsigned long long *)&c.counters[2], (unsigned long long)take); }
            __syncthreads();
            done += take;
        }
    }
}

__global__ __launch_bounds__(256) void hlu_ddprod_kernel(Ctx c, const Task *tasks) {
    __shared__ double rfro[256], rbest[256];
    __shared__ int ridx[256];
    __shared__ int s_stop;
    const Task t = tasks[blockIdx.x];
    const int tid = threadIdx.x, m = t.m, n = t.n, qn = t.r0;
    const double *A = at(c, t.a), *B = at(c, t.b);
    double *W = at(c, t.w), *X = at(c, t.x), *Z = at(c, t.y);
    __shared__ double tiles[2 * 64 * 17];
    for (int e = tid; e < m * n; e += 256) W[e] = 0.0;
    __syncthreads();
    if (t.flags & F_TRANS) wg_gemm_nt(W, m, m, n, qn, 1.0, A, 1, t.a_ld, B, 1, t.b_ld, tiles); // (b is n x q: a b^T)
    else wg_gemm_nt(W, m, m, n, qn, 1.0, A, 1, t.a_ld, B, t.b_ld, 1, tiles);
    const double tol2 = 0.01 * c.eps * c.eps * fmax(c.norm0[t.leaf], 0.0);
    int k = 0;
    while (k < t.kconst) {
        double fro = 0, best = 0;
        int bidx = 0;
        for (int e = tid; e < m * n; e += 256) { const double v = W[e]; fro = fma(v, v, fro); if (fabs(v) > best) { best = fabs(v); bidx = e; } }
        rfro[tid] = fro; rbest[tid] = best; ridx[tid] = bidx;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                rfro[tid] += rfro[tid + s];
                if (rbest[tid + s] > rbest[tid] || (rbest[tid + s] == rbest[tid] && ridx[tid + s] < ridx[tid])) { rbest[tid] = rbest[tid + s]; ridx[tid] = ridx[tid + s]; }
            }
            __syncthreads();
        }
        if (tid == 0) s_stop = (rfro[0] <= tol2 || rbest[0] == 0.0) ? 1 : 0;
        __syncthreads();
        if (s_stop) break;
        const int pidx = ridx[0], bi = pidx % m, bj = pidx / m;
        const double piv = W[pidx];
        __syncthreads();
        for (int i = tid; i < m; i += 256) X[i + (long long)k * t.x_ld] = W[i + (long long)bj * m] / piv;
        for (int j = tid; j < n; j += 256) Z[j + (long long)k * t.y_ld] = W[bi + (long long)j * m];
        __syncthreads();
        for (int e = tid; e < m * n; e += 256) { const int i = e % m, j = e / m; W[e] -= X[i + (long long)k * t.x_ld] * Z[j + (long long)k * t.y_ld]; }
        __syncthreads();
        k++;
    }
    if (tid == 0) { c.rank[t.kref] = k; atomicAdd((unsigned long long *)&c.counters[3], (unsigned long long)k); }
}

__global__ __launch_bounds__(256) void hlu_getrf_kernel(Ctx c, const Task *tasks) {
    extern __shared__ int piv[];
    __shared__ double rbest[256];
    __shared__ int ridx[256];
    const Task t = tasks[blockIdx.x];
    const Leaf L = c.leaves[t.leaf];
    const Diag Dg = c.diags[L.diag];
    const int tid = threadIdx.x, m = L.m;
    double *A = c.space[0] + L.u, *Li = c.space[1] + Dg.linv, *Ui = c.space[1] + Dg.uinv;
    const bool sym = t.flags & F_SYM; // symmetric positive definite leaf: no pivoting (L U with U = D L^T), a positive diagonal expected
    for (int j = 0; j < m; j++) {
        double best = -1.0;
        int bi = j;
        if (!sym) for (int i = j + tid; i < m; i += 256) { const double v = fabs(A[i + (long long)j * m]); if (v > best) { best = v; bi = i; } }
        else if (tid == 0) best = A[j + (long long)j * m] > 0 ? 1.0 : 0.0;
        rbest[tid] = best; ridx[tid] = bi;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s && (rbest[tid + s] > rbest[tid] || (rbest[tid + s] == rbest[tid] && ridx[tid + s] < ridx[tid]))) { rbest[tid] = rbest[tid + s]; ridx[tid] = ridx[tid + s]; }
            __syncthreads();
        }
        const int p = sym ? j : ridx[0];
        if (tid == 0) { piv[j] = p; if (rbest[0] == 0.0) atomicAdd((unsigned long long *)&c.counters[4], 1ull); }
        __syncthreads();
        if (p != j) for (int col = tid; col < m; col += 256) { const double a = A[j + (long long)col * m]; A[j + (long long)col * m] = A[p + (long long)col * m]; A[p + (long long)col * m] = a; }
        __syncthreads();
        const double dinv = A[j + (long long)j * m];
        __syncthreads();
        for (int i = j + 1 + tid; i < m; i += 256) A[i + (long long)j * m] /= dinv;
        __syncthreads();
        const int nn = m - j - 1;
        for (int e = tid; e < nn * nn; e += 256) {
            const int i = j + 1 + e % nn, col = j + 1 + e / nn;
            A[i + (long long)col * m] -= A[i + (long long)j * m] * A[j + (long long)col * m];
        }
        __syncthreads();
    }
    // (P^T L)^-1 = L^-1 P and U^-1 by substitution on ALL columns at once: step j is a rank-one update of the m x m right-hand sides
    for (int e = tid; e < m * m; e += 256) { const int i = e % m, col = e / m; Li[e] = 0.0; Ui[e] = i == col ? 1.0 : 0.0; }
    __syncthreads();
    for (int col = tid; col < m; col += 256) { // Li starts as P: the swaps, in order, move the one of e_col
        int pos = col;
        for (int j = 0; j < m; j++) if (piv[j] != j) { if (pos == j) pos = piv[j]; else if (pos == piv[j]) pos = j; }
        Li[pos + (long long)col * m] = 1.0;
    }
    __syncthreads();
    for (int j = 0; j < m; j++) { // forward: rows i > j of Li -= L[i, j] * row j of Li
        const int nn = m - j - 1;
        for (int e = tid; e < nn * m; e += 256) {
            const int i = j + 1 + e % nn, col = e / nn;
            Li[i + (long long)col * m] -= A[i + (long long)j * m] * Li[j + (long long)col * m];
        }
        __syncthreads();
    }
    for (int j = m - 1; j >= 0; j--) { // backward: row j of Ui /= U[j, j]; rows i < j -= U[i, j] * row j
        const double dj = A[j + (long long)j * m];
        for (int col = tid; col < m; col += 256) Ui[j + (long long)col * m] /= dj;
        __syncthreads();
        for (int e = tid; e < j * m; e += 256) {
            const int i = e % j, col = e / j;
            Ui[i + (long long)col * m] -= A[i + (long long)j * m] * Ui[j + (long long)col * m];
        }
        __syncthreads();
    }
    if (sym) { // the inverse of the Cholesky factor L D^(1/2): rows of L^-1 scaled by 1 / sqrt(d), and its transpose
        for (int e = tid; e < m * m; e += 256) { const int i = e % m; Li[e] /= sqrt(A[i + (long long)i * m]); }
        __syncthreads();
        for (int e = tid; e < m * m; e += 256) { const int i = e % m, col = e / m; Ui[e] = Li[col + (long long)i * m]; }
    }
}

// The same for diagonal leaves of at most 128 rows, with the matrix in LDS: the elimination and the substitutions (blocks of nb
// right-hand-side columns at a time) never touch global memory between their steps.
__global__ __launch_bounds__(256) void hlu_getrf_lds_kernel(Ctx c, const Task *tasks, int lds_doubles) {
    extern __shared__ double smg[];
    __shared__ double rbest[256];
    __shared__ int ridx[256];
    __shared__ int piv[128];
    const Task t = tasks[blockIdx.x];
    const Leaf L = c.leaves[t.leaf];
    const Diag Dg = c.diags[L.diag];
    const int tid = threadIdx.x, m = L.m;
    double *A = c.space[0] + L.u, *Li = c.space[1] + Dg.linv, *Ui = c.space[1] + Dg.uinv;
    double *As = smg, *Bs = smg + m * m;
    const int nb = max(1, min(m, (lds_doubles - m * m) / max(m, 1)));
    for (int e = tid; e < m * m; e += 256) As[e] = A[e];
    __syncthreads();
    const bool sym = t.flags & F_SYM;
    for (int j = 0; j < m; j++) {
        double best = -1.0;
        int bi = j;
        int p = j;
        if (!sym) {
            for (int i = j + tid; i < m; i += 256) { const double v = fabs(As[i + j * m]); if (v > best) { best = v; bi = i; } }
            rbest[tid] = best; ridx[tid] = bi;
            __syncthreads();
            for (int s = 128; s > 0; s >>= 1) {
                if (tid < s && (rbest[tid + s] > rbest[tid] || (rbest[tid + s] == rbest[tid] && ridx[tid + s] < ridx[tid]))) { rbest[tid] = rbest[tid + s]; ridx[tid] = ridx[tid + s]; }
                __syncthreads();
            }
            p = ridx[0];
            if (tid == 0) { piv[j] = p; if (rbest[0] == 0.0) atomicAdd((unsigned long long *)&c.counters[4], 1ull); }
        } else if (tid == 0) { // (no pivot search, none of its barriers: the diagonal entry must be positive)
            piv[j] = j;
            if (!(As[j + j * m] > 0)) atomicAdd((unsigned long long *)&c.counters[4], 1ull);
        }
        __syncthreads();
        if (p != j) for (int col = tid; col < m; col += 256) { const double a = As[j + col * m]; As[j + col * m] = As[p + col * m]; As[p + col * m] = a; }
        __syncthreads();
        const double dinv = As[j + j * m];
        __syncthreads();
        for (int i = j + 1 + tid; i < m; i += 256) As[i + j * m] /= dinv;
        __syncthreads();
        const int nn = m - j - 1;
        for (int e = tid; e < nn * nn; e += 256) {
            const int i = j + 1 + e % nn, col = j + 1 + e / nn;
            As[i + col * m] -= As[i + j * m] * As[j + col * m];
        }
        __syncthreads();
    }
    for (int e = tid; e < m * m; e += 256) A[e] = As[e];
    for (int c0 = 0; c0 < m; c0 += nb) {
        const int w = min(nb, m - c0);
        // (P^T L)^-1 = L^-1 P, columns c0 .. c0 + w
        for (int e = tid; e < m * w; e += 256) Bs[e] = 0.0;
        __syncthreads();
        for (int cc = tid; cc < w; cc += 256) {
            int pos = c0 + cc;
            for (int j = 0; j < m; j++) if (piv[j] != j) { if (pos == j) pos = piv[j]; else if (pos == piv[j]) pos = j; }
            Bs[pos + cc * m] = 1.0;
        }
        __syncthreads();
        for (int j = sym ? c0 : 0; j < m; j++) { // (without a permutation the columns from c0 on are zero above row c0)
            const int nn = m - j - 1;
            for (int e = tid; e < nn * w; e += 256) { const int i = j + 1 + e % nn, cc = e / nn; Bs[i + cc * m] -= As[i + j * m] * Bs[j + cc * m]; }
            __syncthreads();
        }
        if (sym) for (int e = tid; e < m * w; e += 256) Bs[e] /= sqrt(As[(e % m) * (m + 1)]); // rows of L^-1 scaled: the inverse of the Cholesky factor L D^(1/2)
        __syncthreads();
        for (int e = tid; e < m * w; e += 256) Li[(long long)c0 * m + e] = Bs[e];
        __syncthreads();
        if (sym) continue; // (its second inverse factor is the transpose of the first: below)
        // U^-1, columns c0 .. c0 + w
        for (int e = tid; e < m * w; e += 256) { const int i = e % m, cc = e / m; Bs[e] = i == c0 + cc ? 1.0 : 0.0; }
        __syncthreads();
        for (int j = min(m - 1, c0 + w - 1); j >= 0; j--) { // (rows below the block's last column stay zero)
            const double dj = As[j + j * m];
            for (int cc = tid; cc < w; cc += 256) Bs[j + cc * m] /= dj;
            __syncthreads();
            for (int e = tid; e < j * w; e += 256) { const int i = e % j, cc = e / j; Bs[i + cc * m] -= As[i + j * m] * Bs[j + cc * m]; }
            __syncthreads();
        }
        for (int e = tid; e < m * w; e += 256) Ui[(long long)c0 * m + e] = Bs[e];
        __syncthreads();
    }
    if (sym) {
        __threadfence_block();
        __syncthreads();
        for (int e = tid; e < m * m; e += 256) { const int i = e % m, col = e / m; Ui[e] = Li[col + (long long)i * m]; }
    }
}

__global__ void hlu_shift_kernel(Ctx c, int n_diag, double shift) {
    const int dgi = blockIdx.x;
    if (dgi >= n_diag) return;
    const Leaf L = c.leaves[c.diags[dgi].leaf];
    double *A = c.space[0] + L.u;
    for (int i = threadIdx.x; i < L.m; i += blockDim.x) A[i + (long long)i * L.m] += shift;
}
// the leaf (s, t) of an operator that stores one triangle: the transpose of the stored leaf (t, s)
__global__ __launch_bounds__(256) void hlu_mirror_kernel(Ctx c, const int2 *pairs, int rank_known) {
    const Leaf S = c.leaves[pairs[blockIdx.x].x], M = c.leaves[pairs[blockIdx.x].y];
    double *F = c.space[0];
    if (S.kind == 0) {
        for (int e = threadIdx.x; e < S.m * S.n; e += 256) { const int i = e % S.m, j = e / S.m; F[M.u + j + (long long)i * M.m] = F[S.u + e]; }
        return;
    }
    const int k = c.rank[pairs[blockIdx.x].x];
    for (long long e = threadIdx.x; e < (long long)k * S.m; e += 256) F[M.v + e] = F[S.u + e];
    for (long long e = threadIdx.x; e < (long long)k * S.n; e += 256) F[M.u + e] = F[S.v + e];
    if (threadIdx.x == 0) c.rank[pairs[blockIdx.x].y] = k;
    (void)rank_known;
}
// After the factorisation every leaf moves to a tight arena (its rank in columns instead of the 64 of room): one workgroup per leaf
__global__ __launch_bounds__(256) void hlu_compact_kernel(const double *from, double *to, const Leaf *old_leaves, const Leaf *new_leaves, const int *rank, int n_leaves) {
    const int l = blockIdx.x;
    if (l >= n_leaves) return;
    const Leaf A = old_leaves[l], B = new_leaves[l];
    const long long nu = A.kind == 0 ? (long long)A.m * A.n : (long long)A.m * rank[l], nv = A.kind == 0 ? 0 : (long long)A.n * rank[l];
    for (long long e = threadIdx.x; e < nu; e += 256) to[B.u + e] = from[A.u + e];
    for (long long e = threadIdx.x; e < nv; e += 256) to[B.v + e] = from[A.v + e];
}
template <typename T>
__global__ void hlu_permute_rows_kernel(const T *src, T *dst, const int *perm, int n, int mu, int gather) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)n * mu) return;
    const int col = (int)(e / n), i = (int)(e - (long long)col * n);
    if (gather) dst[e] = src[(long long)col * n + perm[i]];
    else dst[(long long)col * n + perm[i]] = src[e];
}

// r <- b - r - shift x   (r holds H x on entry)
__global__ void hlu_residual_kernel(double *r, const double *b, const double *x, double shift, long long count) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) r[e] = b[e] - r[e] - shift * x[e];
}
// x += dx; out2 = (|dx|^2, |x|^2), one workgroup, fixed order
__global__ __launch_bounds__(1024) void hlu_correct_kernel(double *x, const double *dx, long long count, double *out2) {
    __shared__ double s1[1024], s2[1024];
    double a = 0, b = 0;
    for (long long e = threadIdx.x; e < count; e += 1024) { const double d = dx[e], v = x[e] + d; x[e] = v; a = fma(d, d, a); b = fma(v, v, b); }
    s1[threadIdx.x] = a; s2[threadIdx.x] = b;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if ((int)threadIdx.x < s) { s1[threadIdx.x] += s1[threadIdx.x + s]; s2[threadIdx.x] += s2[threadIdx.x + s]; } __syncthreads(); }
    if (threadIdx.x == 0) { out2[0] = s1[0]; out2[1] = s2[0]; }
}

size_t update_lds_bytes(int Kc) { return std::max((size_t)(4 * Kc * (Kc + 1) + Kc + 4) * sizeof(double) + (size_t)(2 * Kc + 8) * sizeof(int), (size_t)2 * 64 * 17 * sizeof(double)); }

struct DevProgram {
    Task *tasks = nullptr;
    long long *seg = nullptr, *aux = nullptr;
    size_t n_tasks = 0, n_seg = 0;
};

void attributes_once() {
    static std::once_flag flag;
    std::call_once(flag, [] {
    HIP_OK(hipFuncSetAttribute((const void *)hlu_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)update_lds_bytes(64)));
    HIP_OK(hipFuncSetAttribute((const void *)hlu_apply_dense_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HLU_MAX_DIM * QC * (int)sizeof(double)));
    HIP_OK(hipFuncSetAttribute((const void *)hlu_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HLU_MAX_DIM * QC * (int)sizeof(double)));
    HIP_OK(hipFuncSetAttribute((const void *)hlu_getrf_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 18 * 1024 * (int)sizeof(double)));
    });
}

// HTOOL_HLU_PROFILE=1: every launch is waited for and its time added up per task kind (printed by device_hlu_factor)
double g_prof_seconds[T_NTYPES] = {0, 0, 0, 0, 0, 0, 0, 0};
long long g_prof_launches[T_NTYPES] = {0, 0, 0, 0, 0, 0, 0, 0}, g_prof_items[T_NTYPES] = {0, 0, 0, 0, 0, 0, 0, 0};
double g_prof_longest[T_NTYPES] = {0, 0, 0, 0, 0, 0, 0, 0};

// one program, bucket by bucket, on `st`; tasks / runs already on the device
void run_program(const Program &G, const DevProgram &dp, const Ctx &c, const std::vector<Leaf> &leaves, hipStream_t st) {
    attributes_once();
    long long limit = -1; // diagnostic: HTOOL_HLU_DEBUG_BUCKETS=<n> stops every program after its first n launches
    if (const char *e = getenv("HTOOL_HLU_DEBUG_BUCKETS")) limit = atoll(e);
    long long launched = 0;
    static const bool profile = getenv("HTOOL_HLU_PROFILE") && atoi(getenv("HTOOL_HLU_PROFILE")) > 0;
    if (profile) HIP_OK(hipStreamSynchronize(st));
    for (size_t bi = 0; bi < G.buckets.size(); bi++) {
        const Bucket &b = G.buckets[bi];
        if (limit >= 0 && launched++ >= limit) break;
        const double tp0 = profile ? wall_seconds() : 0.0;
        if (b.type == T_APPLY_DENSE && bi + 1 < G.buckets.size() && G.buckets[bi + 1].type == T_APPLY_LR && G.buckets[bi + 1].level == b.level && G.buckets[bi + 1].begin == b.end &&
            limit < 0 && !profile) { // both kinds of application of this level in one launch
            const Bucket &b2 = G.buckets[bi + 1];
            int nmax = 1;
            for (int64_t i = b.begin; i < b.end; i++) nmax = std::max(nmax, G.tasks[(size_t)i].n);
            HM_CHECK(nmax <= HLU_MAX_DIM, "hierarchical LU: a dense leaf has more rows or columns than the kernels stage on chip");
            hipLaunchKernelGGL(hlu_apply_kernel, dim3((unsigned)(b2.end - b.begin)), dim3(256), (size_t)nmax * QC * sizeof(double), st, c, dp.tasks + b.begin);
            bi++;
            continue;
        }
        const unsigned n = (unsigned)(b.end - b.begin);
        const Task *t0 = dp.tasks + b.begin;
        switch (b.type) {
        case T_FILL: hipLaunchKernelGGL(hlu_fill_kernel, dim3(n), dim3(256), 0, st, c, t0); break;
        case T_APPLY_DENSE: {
            int nmax = 1;
            for (int64_t i = b.begin; i < b.end; i++) nmax = std::max(nmax, G.tasks[(size_t)i].n);
            HM_CHECK(nmax <= HLU_MAX_DIM, "hierarchical LU: a dense leaf has more rows or columns than the kernels stage on chip");
            hipLaunchKernelGGL(hlu_apply_dense_kernel, dim3(n), dim3(256), (size_t)nmax * QC * sizeof(double), st, c, t0);
            break;
        }
        case T_APPLY_LR: hipLaunchKernelGGL(hlu_apply_lr_kernel, dim3(n), dim3(256), 0, st, c, t0); break;
        case T_ADDLR:
        case T_FINAL: {
            int Kc = 2;
            for (int64_t i = b.begin; i < b.end; i++) Kc = std::max(Kc, leaves[(size_t)G.tasks[(size_t)i].leaf].cap);
            hipLaunchKernelGGL(hlu_update_kernel, dim3((unsigned)(b.seg_end - b.seg_begin)), dim3(256), update_lds_bytes(Kc), st, c, dp.tasks, dp.seg + b.seg_begin, Kc);
            break;
        }
        case T_DDPROD: hipLaunchKernelGGL(hlu_ddprod_kernel, dim3(n), dim3(256), 0, st, c, t0); break;
        case T_GETRF: {
            int mmax = 1;
            for (int64_t i = b.begin; i < b.end; i++) mmax = std::max(mmax, G.tasks[(size_t)i].m);
            constexpr int LDS_DOUBLES = 18 * 1024; // 144 KB of the 160 KB of a CU
            if (mmax <= 128 && mmax * mmax + mmax <= LDS_DOUBLES) hipLaunchKernelGGL(hlu_getrf_lds_kernel, dim3(n), dim3(256), (size_t)LDS_DOUBLES * sizeof(double), st, c, t0, LDS_DOUBLES);
            else hipLaunchKernelGGL(hlu_getrf_kernel, dim3(n), dim3(256), (size_t)mmax * sizeof(int), st, c, t0);
            break;
        }
        case T_REDUCE: hipLaunchKernelGGL(hlu_reduce_kernel, dim3(n), dim3(256), 0, st, c, t0, dp.aux); break;
        default: throw Error("hierarchical LU: unknown task kind");
        }
        if (profile) {
            HIP_OK(hipStreamSynchronize(st));
            const double dt = wall_seconds() - tp0;
            g_prof_seconds[b.type] += dt; g_prof_launches[b.type]++; g_prof_items[b.type] += b.end - b.begin;
            g_prof_longest[b.type] = std::max(g_prof_longest[b.type], dt);
        }
    }
    HIP_OK(hipGetLastError());
}

struct DevBuf { // device allocation released on scope exit
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void alloc(size_t bytes) { HIP_OK(hipMalloc(&p, std::max<size_t>(bytes, 16))); }
    template <typename T> T *as() const { return (T *)p; }
};

void upload_program(const Program &G, DevBuf &tasks, DevBuf &seg, DevProgram &dp, hipStream_t st) {
    dp.n_tasks = G.tasks.size(); dp.n_seg = G.seg.size();
    dp.tasks = tasks.as<Task>(); dp.seg = seg.as<long long>();
    if (!G.tasks.empty()) HIP_OK(hipMemcpyAsync(dp.tasks, G.tasks.data(), G.tasks.size() * sizeof(Task), hipMemcpyHostToDevice, st));
    if (!G.seg.empty()) HIP_OK(hipMemcpyAsync(dp.seg, G.seg.data(), G.seg.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
}

} // namespace

// ---- the factorisation object ------------------------------------------------------------------------------------------
struct DeviceHLU {
    Plan *plan = nullptr;
    int device = 0, n = 0, asked = 1;
    bool whole = true;
    double shift = 0;
    mutable int last_refinements = 0;
    mutable double last_correction = 0;
    hipStream_t stream = nullptr;
    const int *perm = nullptr; // the operator's permutation (device; owned by the operator)
    double *factor = nullptr, *diag = nullptr;
    Leaf *leaves = nullptr;
    Diag *diags = nullptr;
    int *rank = nullptr;
    double *norm0 = nullptr, *norm2 = nullptr;
    long long *counters = nullptr;
    DevProgram solve_n, solve_t;
    double *solve_scratch = nullptr;
    hipEvent_t last_solve = nullptr;   // the solves of one factorisation share the slots of the sweeps: each waits for the one before, whatever their streams
    mutable std::mutex solve_mu;
    int64_t stats[16] = {0};
    double seconds[4] = {0, 0, 0, 0}; // plan, unpack, factorisation, total
    ~DeviceHLU() {
        (void)hipSetDevice(device);
        if (last_solve) (void)hipEventDestroy(last_solve);
        for (void *p : {(void *)factor, (void *)diag, (void *)leaves, (void *)diags, (void *)rank, (void *)norm0, (void *)norm2, (void *)counters, (void *)solve_n.tasks, (void *)solve_n.seg, (void *)solve_n.aux,
                        (void *)solve_t.tasks, (void *)solve_t.seg, (void *)solve_t.aux, (void *)solve_scratch})
            if (p) (void)hipFree(p);
        delete plan;
    }
    Ctx ctx(double *scratch, double *rhs, long long ld, int nrhs) const {
        Ctx c;
        c.space[SP_FACTOR] = factor; c.space[SP_DIAG] = diag; c.space[SP_SCRATCH] = scratch; c.space[SP_RHS] = rhs;
        c.ld_rhs = ld; c.nrhs = nrhs; c.leaves = leaves; c.diags = diags; c.rank = rank; c.norm0 = norm0; c.norm2 = norm2;
        c.eps = plan->params.eps; c.counters = counters;
        return c;
    }
};

void device_hlu_free(DeviceHLU *f) { delete f; }
int device_hlu_kind(const DeviceHLU *f) { return f ? f->asked : 0; }

namespace {

// device tables of a plan; the factor arena is allocated but not filled
void hlu_allocate(DeviceHLU &f) {
    const Plan &P = *f.plan;
    HIP_OK(hipMalloc((void **)&f.factor, std::max<size_t>((size_t)P.factor_elems * 8, 16)));
    HIP_OK(hipMalloc((void **)&f.diag, std::max<size_t>((size_t)P.diag_elems * 8, 16)));
    HIP_OK(hipMalloc((void **)&f.leaves, std::max<size_t>(P.leaves.size() * sizeof(Leaf), 16)));
    HIP_OK(hipMalloc((void **)&f.diags, std::max<size_t>(P.diags.size() * sizeof(Diag), 16)));
    HIP_OK(hipMalloc((void **)&f.rank, std::max<size_t>((size_t)P.n_slots * 4, 16)));
    HIP_OK(hipMalloc((void **)&f.norm0, std::max<size_t>(P.leaves.size() * 8, 16)));
    HIP_OK(hipMalloc((void **)&f.norm2, std::max<size_t>(P.leaves.size() * 8, 16)));
    HIP_OK(hipMalloc((void **)&f.counters, 8 * sizeof(long long)));
    HIP_OK(hipMemcpy(f.leaves, P.leaves.data(), P.leaves.size() * sizeof(Leaf), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(f.diags, P.diags.data(), P.diags.size() * sizeof(Diag), hipMemcpyHostToDevice));
    std::vector<int> r0((size_t)std::max<int64_t>(P.n_slots, 1), 0);
    for (size_t i = 0; i < P.leaves.size(); i++) r0[i] = std::max(P.leaves[i].rank0, 0);
    HIP_OK(hipMemcpy(f.rank, r0.data(), (size_t)P.n_slots * 4, hipMemcpyHostToDevice));
    std::vector<double> minus(P.leaves.size(), -1.0);
    HIP_OK(hipMemcpy(f.norm0, minus.data(), minus.size() * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(f.norm2, 0, std::max<size_t>(P.leaves.size() * 8, 16)));
    HIP_OK(hipMemset(f.counters, 0, 8 * sizeof(long long)));
}

// windows [first, last] of the factorisation on the stream; the tasks of a window are uploaded while the previous one runs
void hlu_run_windows(DeviceHLU &f, int first, int last, hipStream_t st, double *scratch_host = nullptr) {
    const Plan &P = *f.plan;
    size_t max_tasks = 1, max_seg = 1;
    for (int w = first; w <= last; w++) { max_tasks = std::max(max_tasks, P.factor[(size_t)w].tasks.size()); max_seg = std::max(max_seg, P.factor[(size_t)w].seg.size()); }
    DevBuf tasks[2], seg[2], scratch;
    for (int q = 0; q < 2; q++) { tasks[q].alloc(max_tasks * sizeof(Task)); seg[q].alloc(max_seg * sizeof(int64_t)); }
    scratch.alloc((size_t)P.scratch_elems * 8);
    if (scratch_host) HIP_OK(hipMemcpy(scratch.p, scratch_host, (size_t)P.scratch_elems * 8, hipMemcpyHostToDevice));
    const Ctx c = f.ctx(scratch.as<double>(), nullptr, 0, 0);
    hipStream_t up = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    hipEvent_t uploaded[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    try {
        for (int q = 0; q < 2; q++) { HIP_OK(hipEventCreateWithFlags(&uploaded[q], hipEventDisableTiming)); HIP_OK(hipEventCreateWithFlags(&done[q], hipEventDisableTiming)); }
        DevProgram dp[2];
        upload_program(P.factor[(size_t)first], tasks[0], seg[0], dp[0], up);
        HIP_OK(hipEventRecord(uploaded[0], up));
        for (int w = first; w <= last; w++) {
            const int q = (w - first) & 1;
            if (w + 1 <= last) { // the other buffer is free once the window before this one has run
                if (w > first) HIP_OK(hipStreamWaitEvent(up, done[q ^ 1], 0));
                upload_program(P.factor[(size_t)w + 1], tasks[q ^ 1], seg[q ^ 1], dp[q ^ 1], up);
                HIP_OK(hipEventRecord(uploaded[q ^ 1], up));
            }
            HIP_OK(hipStreamWaitEvent(st, uploaded[q], 0));
            run_program(P.factor[(size_t)w], dp[q], c, P.leaves, st);
            HIP_OK(hipEventRecord(done[q], st));
        }
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipStreamSynchronize(up));
        if (scratch_host) HIP_OK(hipMemcpy(scratch_host, scratch.p, (size_t)P.scratch_elems * 8, hipMemcpyDeviceToHost));
    } catch (...) {
        (void)hipDeviceSynchronize();
        for (int q = 0; q < 2; q++) { if (uploaded[q]) (void)hipEventDestroy(uploaded[q]); if (done[q]) (void)hipEventDestroy(done[q]); }
        (void)hipStreamDestroy(up);
        throw;
    }
    for (int q = 0; q < 2; q++) { (void)hipEventDestroy(uploaded[q]); (void)hipEventDestroy(done[q]); }
    (void)hipStreamDestroy(up);
}

void hlu_upload_solves(DeviceHLU &f) {
    const Plan &P = *f.plan;
    const size_t slots = (size_t)std::max(P.solve_n.scratch_elems, P.solve_t.scratch_elems); // the private slots of the sweeps
    HIP_OK(hipMalloc((void **)&f.solve_scratch, std::max<size_t>(slots * 8, 16)));
    HIP_OK(hipEventCreateWithFlags(&f.last_solve, hipEventDisableTiming));
    for (int pass = 0; pass < 2; pass++) {
        const Program &G = pass ? P.solve_t : P.solve_n;
        DevProgram &dp = pass ? f.solve_t : f.solve_n;
        HIP_OK(hipMalloc((void **)&dp.tasks, std::max<size_t>(G.tasks.size() * sizeof(Task), 16)));
        HIP_OK(hipMalloc((void **)&dp.seg, std::max<size_t>(G.seg.size() * sizeof(int64_t), 16)));
        HIP_OK(hipMalloc((void **)&dp.aux, std::max<size_t>(G.aux.size() * sizeof(int64_t), 16)));
        if (!G.aux.empty()) HIP_OK(hipMemcpy(dp.aux, G.aux.data(), G.aux.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        dp.n_tasks = G.tasks.size(); dp.n_seg = G.seg.size();
        if (!G.tasks.empty()) HIP_OK(hipMemcpy(dp.tasks, G.tasks.data(), G.tasks.size() * sizeof(Task), hipMemcpyHostToDevice));
        if (!G.seg.empty()) HIP_OK(hipMemcpy(dp.seg, G.seg.data(), G.seg.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    }
}

} // namespace

// B (n x mu, column c at B_dev + c * ldb, CLUSTER numbering) <- A^-1 B or A^-T B, enqueued on `stream`
void device_hlu_solve(const DeviceHLU *f, char trans, void *B_dev, long long ldb, int mu, void *stream) {
    HM_CHECK(f != nullptr, "factor solve: no factorisation");
    HM_CHECK(trans == 'N' || trans == 'T' || trans == 'C', "factor solve: trans must be 'N', 'T' or 'C'");
    HM_CHECK(ldb >= f->n && mu >= 0, "factor solve: bad leading dimension");
    if (f->n == 0 || mu == 0) return;
    HIP_OK(hipSetDevice(f->device));
    std::lock_guard<std::mutex> lock(f->solve_mu);
    if (f->last_solve) HIP_OK(hipStreamWaitEvent((hipStream_t)stream, f->last_solve, 0));
    // (the private slots of a sweep hold SOLVE_SLOT_COLUMNS right-hand sides: more run in chunks)
    for (int c0 = 0; c0 < mu; c0 += SOLVE_SLOT_COLUMNS) {
    const Ctx c = f->ctx(f->solve_scratch, (double *)B_dev + (long long)c0 * ldb, ldb, std::min(SOLVE_SLOT_COLUMNS, mu - c0));
    // (the caller's stream as it is: NULL is the legacy default stream, which orders this against the caller's other default-stream work and
    // against every blocking stream -- substituting the operator's own stream here would let a product on ANOTHER handle's stream overtake it)
    run_program(trans == 'N' ? f->plan->solve_n : f->plan->solve_t, trans == 'N' ? f->solve_n : f->solve_t, c, f->plan->leaves, (hipStream_t)stream);
    }
    if (f->last_solve) HIP_OK(hipEventRecord(f->last_solve, (hipStream_t)stream));
    if (getenv("HTOOL_HLU_PROFILE") && atoi(getenv("HTOOL_HLU_PROFILE")) > 1) { // (every launch was waited for: where the time of a solve goes)
        static const char *names[T_NTYPES] = {"FILL", "APPLY_DENSE", "APPLY_LR", "ADDLR", "FINAL", "DDPROD", "GETRF", "REDUCE"};
        for (int q : {1, 2, 7}) {
            fprintf(stderr, "[hlu solve profile] %-12s %9.4f s  %8lld launches  %10lld tasks  longest launch %.6f s\n", names[q], g_prof_seconds[q], g_prof_launches[q], g_prof_items[q], g_prof_longest[q]);
            g_prof_seconds[q] = 0; g_prof_launches[q] = 0; g_prof_items[q] = 0; g_prof_longest[q] = 0;
        }
    }
}

// ---- the factorisation of an operator -----------------------------------------------------------------------------------------
// kind: 1 LU, 2 Cholesky (the same factorisation: the hierarchical arithmetic here is the LU; a symmetric operator stored as one
// triangle gets its other triangle as transposed leaves).  eps_lu <= 0: a tenth of the operator's epsilon (HTOOL_HLU_EPS overrides): the
// errors of the blockwise truncations are amplified by the condition number of the operator, which grows with its size; lu_solve besides
// refines its answer against the operator's product (device_hlu_solve_host), which is what meets the bar of the reference's tests
// (tests/test_hmatrix.py:104: error below epsilon).
// Throws hm::Error when the operator is not one this factorisation covers (the caller falls back to the dense one).
static DeviceHLU *hlu_factor_impl(const HMatrix &H, int kind, double shift, double eps_lu, bool symmetric);
// A symmetric operator (declared 'S', or any operator handed to cholesky_factorization, whose contract is that only the UPLO triangle is
// looked at) is factorised as A = L L^T on its lower triangle -- half the work; an LU request that meets a matrix which is not positive
// definite starts again as an LU (HTOOL_HLU_SYMMETRIC=0: always the LU).
DeviceHLU *device_hlu_factor(const HMatrix &H, int kind, double shift, double eps_lu) {
    static const bool sym_off = getenv("HTOOL_HLU_SYMMETRIC") && std::string(getenv("HTOOL_HLU_SYMMETRIC")) == "0";
    const bool upper_only = H.one_triangle && H.params.uplo == 'U'; // (stored as the upper triangle: served by the LU on mirrored leaves)
    const bool symmetric = !sym_off && !H.is_complex && !upper_only && (kind == 2 || H.params.symmetry == 'S');
    if (!symmetric) return hlu_factor_impl(H, kind, shift, eps_lu, false);
    try {
        return hlu_factor_impl(H, kind, shift, eps_lu, true);
    } catch (const Error &e) {
        if (kind == 2 || std::string(e.what()).find("not positive definite") == std::string::npos) throw;
        log_message(LOG_DEBUG, "lu_factorization of a symmetric operator: not positive definite, factorising as an LU");
        return hlu_factor_impl(H, kind, shift, eps_lu, false);
    }
}
static DeviceHLU *hlu_factor_impl(const HMatrix &H, int kind, double shift, double eps_lu, bool symmetric) {
    DeviceHMatrix *D = H.dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    HM_CHECK(!H.is_complex, "hierarchical LU: complex operators are factorised by the dense fallback");
    HM_CHECK(H.tc == H.sc && H.t_root == H.s_root && H.row_size == H.col_size, "hierarchical LU needs a square H-matrix on one cluster (sub)tree");
    const double t_begin = wall_seconds();
    HIP_OK(hipSetDevice(D->device));
    const ClusterTree &T = *H.tc;
    const std::vector<BlockRec> &blocks = H.blocks();
    std::vector<LeafIn> in;
    std::vector<int64_t> sel; // plan leaf i = block sel[i] of the operator (then the mirrored ones)
    in.reserve(blocks.size() * (H.one_triangle ? 2 : 1));
    for (size_t i = 0; i < blocks.size(); i++) {
        const BlockRec &b = blocks[i];
        if (symmetric && b.t_off < b.s_off) continue; // (the lower triangle and the diagonal leaves only)
        in.push_back({b.t_node, b.s_node, b.rank < 0 ? -1 : b.rank});
        sel.push_back((int64_t)i);
    }
    std::vector<int2> mirrors; // (stored leaf, its transposed twin)
    if (H.one_triangle && !symmetric)
        for (size_t i = 0; i < blocks.size(); i++)
            if (blocks[i].t_node != blocks[i].s_node) {
                mirrors.push_back(make_int2((int)i, (int)in.size()));
                in.push_back({blocks[i].s_node, blocks[i].t_node, blocks[i].rank < 0 ? -1 : blocks[i].rank});
            }
    Params prm;
    prm.symmetric = symmetric;
    prm.eps = eps_lu > 0 ? eps_lu : 0.1 * H.params.epsilon;
    if (const char *e = getenv("HTOOL_HLU_EPS")) if (atof(e) > 0) prm.eps = atof(e);
    HM_CHECK(prm.eps >= 1e-7, "hierarchical LU: tolerances below 1e-7 are beyond the Gram-matrix truncation of the low-rank arithmetic (the dense factorisation takes over)");
    const double eb = std::min(std::max(H.params.epsilon, 1e-12), 0.5);
    prm.cap_factor = 2.5 * std::max(1.0, std::log(prm.eps) / std::log(eb)); // (measured: the ranks of the factors reach 2.5-3 x those of the operator at a tenth of its tolerance)
    if (const char *e = getenv("HTOOL_HLU_CAP_FACTOR")) if (atof(e) > 0) prm.cap_factor = atof(e);
    if (const char *e = getenv("HTOOL_HLU_WINDOW_MB")) if (atof(e) > 0) prm.window_scratch_elems = (int64_t)(atof(e) * 1e6 / 8);
    if (const char *e = getenv("HTOOL_HLU_SUPER_ROWS")) prm.super_rows = atoi(e); // (diagonal blocks of at most this many rows get explicit inverse factors; 0: none)
    if (const char *e = getenv("HTOOL_HLU_SOLVE_SLOTS")) prm.solve_slots = atoi(e) != 0; // (0: the leaves of a block step subtract one after the other, as in the first version)
    if (const char *e = getenv("HTOOL_HLU_SPLIT")) { // "min,part,max": how long runs of updates of one leaf are dealt out (hlu.hpp: Params::split_*)
        int a = 0, b = 0, c3 = 0;
        if (sscanf(e, "%d,%d,%d", &a, &b, &c3) == 3 && a >= 1 && b >= 1 && c3 >= 1) { prm.split_min = a; prm.split_part = b; prm.split_max_parts = c3; }
    }
    std::unique_ptr<DeviceHLU> f(new DeviceHLU);
    f->plan = make_plan(T, in, prm, H.t_root);
    const Plan &P = *f->plan;
    for (const Leaf &L : P.leaves) if (L.kind == 0) HM_CHECK(L.m <= HLU_MAX_DIM && L.n <= HLU_MAX_DIM, "hierarchical LU: a dense leaf has more rows or columns than the kernels stage on chip");
    f->device = D->device; f->n = P.n; f->asked = kind; f->stream = D->stream; f->perm = D->perm_t;
    f->whole = H.t_root == 0 && !H.local_numbering;
    f->shift = shift;
    f->seconds[0] = P.plan_seconds;
    {
        size_t free_b = 0, total_b = 0;
        HIP_OK(hipMemGetInfo(&free_b, &total_b));
        const double need = ((double)P.factor_elems + P.diag_elems + P.scratch_elems) * 8 + 2.0 * 96 * (double)prm.window_tasks + 512e6;
        if (need > (double)free_b) (void)device_release_workspace();
        HIP_OK(hipMemGetInfo(&free_b, &total_b));
        HM_CHECK(need <= (double)free_b, strprintf("hierarchical LU: %.1f GB needed (factors %.1f, scratch %.1f), %.1f GB free", need / 1e9, P.factor_elems * 8e-9, P.scratch_elems * 8e-9, free_b / 1e9));
    }
    hlu_allocate(*f);
    const double t_unpack = wall_seconds();
    {
        std::vector<int64_t> uo(sel.size()), vo(sel.size());
        for (size_t i = 0; i < sel.size(); i++) { uo[i] = P.leaves[i].u; vo[i] = P.leaves[i].v; }
        device_unpack_leaves(H, (int64_t)sel.size(), sel.data(), uo.data(), vo.data(), f->factor);
    }
    const Ctx c0 = f->ctx(nullptr, nullptr, 0, 0);
    if (!mirrors.empty()) {
        DevBuf mp;
        mp.alloc(mirrors.size() * sizeof(int2));
        HIP_OK(hipMemcpy(mp.p, mirrors.data(), mirrors.size() * sizeof(int2), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(hlu_mirror_kernel, dim3((unsigned)mirrors.size()), dim3(256), 0, D->stream, c0, mp.as<int2>(), 0);
        HIP_OK(hipStreamSynchronize(D->stream));
    }
    if (shift != 0.0 && !P.diags.empty()) hipLaunchKernelGGL(hlu_shift_kernel, dim3((unsigned)P.diags.size()), dim3(128), 0, D->stream, c0, (int)P.diags.size(), shift);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(D->stream));
    f->seconds[1] = wall_seconds() - t_unpack;
    const double t_fact = wall_seconds();
    hlu_run_windows(*f, 0, (int)P.factor.size() - 1, D->stream);
    if (!P.invert.tasks.empty()) { // the explicit inverse factors of the small diagonal blocks: what the sweeps of a solve will multiply with
        DevBuf it, is;
        it.alloc(P.invert.tasks.size() * sizeof(Task));
        is.alloc(std::max<size_t>(P.invert.seg.size(), 1) * sizeof(int64_t));
        DevProgram dp;
        upload_program(P.invert, it, is, dp, D->stream);
        run_program(P.invert, dp, f->ctx(nullptr, nullptr, 0, 0), P.leaves, D->stream);
        HIP_OK(hipStreamSynchronize(D->stream));
    }
    f->seconds[2] = wall_seconds() - t_fact;
    if (getenv("HTOOL_HLU_PROFILE") && atoi(getenv("HTOOL_HLU_PROFILE")) > 0) {
        long long cc[8];
        HIP_OK(hipMemcpy(cc, f->counters, sizeof(cc), hipMemcpyDeviceToHost));
        fprintf(stderr, "[hlu profile] truncations %lld: Gram %.3f s, first Cholesky + products %.3f s, second Cholesky %.3f s, transforms %.3f s, application %.3f s (workgroup time, summed)\n", cc[1],
                (double)((unsigned long long)cc[5] >> 32) * 1e-8, (double)(cc[5] & 0xffffffffll) * 1e-8, (double)((unsigned long long)cc[6] >> 32) * 1e-8, (double)(cc[6] & 0xffffffffll) * 1e-8, (double)cc[7] * 1e-8);
        static const char *names[T_NTYPES] = {"FILL", "APPLY_DENSE", "APPLY_LR", "ADDLR", "FINAL", "DDPROD", "GETRF", "REDUCE"};
        for (int q = 0; q < T_NTYPES; q++) {
            fprintf(stderr, "[hlu profile] %-12s %9.3f s  %8lld launches  %10lld tasks  longest launch %.4f s\n", names[q], g_prof_seconds[q], g_prof_launches[q], g_prof_items[q], g_prof_longest[q]);
            g_prof_seconds[q] = 0; g_prof_launches[q] = 0; g_prof_items[q] = 0; g_prof_longest[q] = 0;
        }
    }
    int64_t arena_peak_bytes = (P.factor_elems + P.diag_elems) * 8;
    static const bool no_compact = getenv("HTOOL_HLU_COMPACT") && std::string(getenv("HTOOL_HLU_COMPACT")) == "0";
    if (!no_compact) { // the factors keep only their rank in columns: a tight arena replaces the one with 64 columns of room per leaf
        Plan &PW = *f->plan;
        const size_t nl = (size_t)PW.n_real_leaves;
        std::vector<int> r(nl);
        HIP_OK(hipMemcpy(r.data(), f->rank, nl * 4, hipMemcpyDeviceToHost));
        std::vector<Leaf> tight(PW.leaves.begin(), PW.leaves.begin() + nl);
        int64_t fe = 0;
        for (size_t i = 0; i < nl; i++) {
            Leaf &L = tight[i];
            L.u = fe;
            if (L.kind == 1) {
                L.cap = std::max(r[i], 1);
                fe += ((int64_t)L.m * L.cap + 1) & ~(int64_t)1;
                L.v = fe;
                fe += ((int64_t)L.n * L.cap + 1) & ~(int64_t)1;
            } else fe += ((int64_t)L.m * L.n + 1) & ~(int64_t)1;
        }
        double *packed = nullptr;
        if (hipMalloc((void **)&packed, std::max<size_t>((size_t)fe * 8, 16)) == hipSuccess) {
            DevBuf d_new;
            d_new.alloc(nl * sizeof(Leaf));
            HIP_OK(hipMemcpy(d_new.p, tight.data(), nl * sizeof(Leaf), hipMemcpyHostToDevice));
            if (nl) hipLaunchKernelGGL(hlu_compact_kernel, dim3((unsigned)nl), dim3(256), 0, D->stream, f->factor, packed, f->leaves, d_new.as<Leaf>(), f->rank, (int)nl);
            HIP_OK(hipGetLastError());
            HIP_OK(hipStreamSynchronize(D->stream));
            // the solve programs refer to rows of the factors: moved leaf by leaf
            for (Program *G : {&PW.solve_n, &PW.solve_t})
                for (Task &t : G->tasks) {
                    if (t.leaf < 0 || (size_t)t.leaf >= nl) continue;
                    const Leaf &O = PW.leaves[(size_t)t.leaf], &N = tight[(size_t)t.leaf];
                    auto move = [&](int64_t ref) -> int64_t {
                        if ((int)(ref >> SPACE_SHIFT) != SP_FACTOR) return ref;
                        const int64_t off = ref & (((int64_t)1 << SPACE_SHIFT) - 1);
                        if (O.kind == 1 && off >= O.v && off < O.v + (int64_t)O.n * O.cap) return make_ref(SP_FACTOR, off - O.v + N.v);
                        return make_ref(SP_FACTOR, off - O.u + N.u);
                    };
                    if (t.type == T_APPLY_LR) { t.a = move(t.a); t.b = move(t.b); }
                    else if (t.type == T_APPLY_DENSE && !(t.flags & F_INPLACE)) t.a = move(t.a);
                }
            for (size_t i = 0; i < nl; i++) PW.leaves[i] = tight[i];
            HIP_OK(hipMemcpy(f->leaves, PW.leaves.data(), nl * sizeof(Leaf), hipMemcpyHostToDevice));
            (void)hipFree(f->factor);
            f->factor = packed;
            PW.factor_elems = fe;
        } else (void)hipGetLastError(); // (no room for the second arena: the factors stay where they are)
    }
    hlu_upload_solves(*f);
    long long counters[8];
    HIP_OK(hipMemcpy(counters, f->counters, sizeof(counters), hipMemcpyDeviceToHost));
    if (symmetric) HM_CHECK(counters[4] == 0, kind == 1 ? "lu_factorization: the symmetric operator is not positive definite" : "cholesky_factorization: matrix is not positive definite");
    HM_CHECK(counters[4] == 0, kind == 1 ? "lu_factorization: singular matrix (zero pivot in a diagonal leaf)" : "cholesky_factorization: singular matrix (zero pivot in a diagonal leaf)");
    int64_t tasks = 0, launches = 0, rank_sum = 0, lr_rows = 0;
    for (const Program &w : P.factor) { tasks += (int64_t)w.tasks.size(); launches += (int64_t)w.buckets.size(); }
    {
        std::vector<int> r((size_t)P.n_real_leaves);
        HIP_OK(hipMemcpy(r.data(), f->rank, r.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < r.size(); i++) if (P.leaves[i].kind == 1) { rank_sum += (int64_t)r[i] * (P.leaves[i].m + P.leaves[i].n); lr_rows += P.leaves[i].m + P.leaves[i].n; }
    }
    f->seconds[3] = wall_seconds() - t_begin;
    int64_t *s = f->stats;
    s[0] = P.n; s[1] = P.n_real_leaves; s[2] = tasks; s[3] = launches; s[4] = (int64_t)P.factor.size();
    s[5] = (P.factor_elems + P.diag_elems) * 8; s[6] = arena_peak_bytes + P.scratch_elems * 8; // (the factors as they stay resident; arena with 64 columns of room per leaf + scratch while factorising)
    s[7] = counters[0]; s[8] = counters[1]; s[9] = counters[2]; s[10] = counters[3];
    s[11] = (int64_t)P.solve_n.tasks.size(); s[12] = (int64_t)P.solve_n.buckets.size(); s[13] = rank_sum; s[14] = lr_rows; s[15] = (int64_t)(prm.eps * 1e12);
    if (counters[0] > 0)
        log_message(LOG_WARNING, strprintf("hierarchical LU: %lld truncations were cut at the capacity of their leaf (accuracy below the asked %.1e: raise HTOOL_HLU_CAP_FACTOR, now %.2f)",
                                           counters[0], prm.eps, prm.cap_factor));
    log_message(LOG_INFO, strprintf("hierarchical %s of the %d x %d operator: plan %.3f s (%lld tasks, %lld launches, %d windows), leaves into the factor arena %.3f s, factorisation %.3f s; "
                                    "factors %.2f GB resident (%.2f GB of arena and scratch while factorising), eps %.1e, mean rank %.1f", symmetric ? (kind == 1 ? "LU by Cholesky (symmetric positive definite)" : "Cholesky") : (kind == 1 ? "LU" : "Cholesky (as LU)"), P.n, P.n, f->seconds[0], (long long)tasks, (long long)launches,
                                    (int)P.factor.size(), f->seconds[1], f->seconds[2], s[5] / 1e9, s[6] / 1e9, prm.eps, lr_rows ? (double)rank_sum / lr_rows : 0.0));
    // the programs of the factorisation are not needed any more (the solves are on the device)
    for (Program &w : f->plan->factor) { std::vector<Task>().swap(w.tasks); std::vector<int64_t>().swap(w.seg); }
    return f.release();
}

void device_hlu_stats(const DeviceHLU *f, int64_t *out16, double *seconds4) {
    HM_CHECK(f != nullptr, "no hierarchical factorisation");
    if (out16) for (int i = 0; i < 16; i++) out16[i] = f->stats[i];
    if (seconds4) for (int i = 0; i < 4; i++) seconds4[i] = f->seconds[i];
}

// host right-hand sides in USER numbering (the reference's lu_solve / cholesky_solve): to cluster numbering on the device, solved, back.
// The factors are accurate to the tolerance of the low-rank arithmetic; the answer is then REFINED against the operator's own product
// (x += (LU)^-1 (b - H x): at most HTOOL_HLU_REFINE steps, default 3, until the correction is below 1e-10 of the solution), so what
// comes back solves the H-matrix's system the way the reference's exact-arithmetic reading of lu_solve would, not merely to the
// truncation tolerance times the condition number.  The device entry (device_hlu_solve: the preconditioner of a Krylov loop) applies
// the factors once.
void device_hlu_solve_host(const HMatrix &H, const DeviceHLU *f, char trans, void *B, int mu) {
    const int n = f->n;
    if (n == 0 || mu == 0) return;
    DeviceHMatrix *D = H.dev;
    HIP_OK(hipSetDevice(f->device));
    const long long count = (long long)n * mu;
    DevBuf d_in, d_x, d_b, d_r, d_nrm;
    d_in.alloc((size_t)count * 8);
    d_x.alloc((size_t)count * 8);
    hipStream_t st = D->stream;
    HIP_OK(hipMemcpyAsync(d_in.p, B, (size_t)count * 8, hipMemcpyHostToDevice, st));
    const unsigned nblk = (unsigned)((count + 255) / 256);
    if (f->whole) hipLaunchKernelGGL(hlu_permute_rows_kernel<double>, dim3(nblk), dim3(256), 0, st, d_in.as<double>(), d_x.as<double>(), D->perm_t, n, mu, 1);
    else HIP_OK(hipMemcpyAsync(d_x.p, d_in.p, (size_t)count * 8, hipMemcpyDeviceToDevice, st));
    int max_refine = 3;
    if (const char *e = getenv("HTOOL_HLU_REFINE")) max_refine = std::max(0, atoi(e));
    if (max_refine > 0) {
        d_b.alloc((size_t)count * 8);
        d_r.alloc((size_t)count * 8);
        d_nrm.alloc(16);
        HIP_OK(hipMemcpyAsync(d_b.p, d_x.p, (size_t)count * 8, hipMemcpyDeviceToDevice, st));
    }
    device_hlu_solve(f, trans, d_x.p, n, mu, (void *)st);
    f->last_refinements = 0;
    f->last_correction = 0;
    for (int it = 0; it < max_refine; it++) {
        device_matmat_device(H, d_x.p, n, d_r.p, n, mu, 1, (void *)st, trans == 'N' ? 'N' : 'T');
        hipLaunchKernelGGL(hlu_residual_kernel, dim3(nblk), dim3(256), 0, st, d_r.as<double>(), d_b.as<double>(), d_x.as<double>(), f->shift, count);
        device_hlu_solve(f, trans, d_r.p, n, mu, (void *)st);
        hipLaunchKernelGGL(hlu_correct_kernel, dim3(1), dim3(1024), 0, st, d_x.as<double>(), d_r.as<double>(), count, d_nrm.as<double>());
        double nrm[2];
        HIP_OK(hipMemcpyAsync(nrm, d_nrm.p, 16, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        f->last_refinements = it + 1;
        f->last_correction = nrm[1] > 0 ? std::sqrt(nrm[0] / nrm[1]) : 0.0;
        if (!(nrm[0] > 1e-20 * nrm[1])) break;
    }
    if (f->whole) hipLaunchKernelGGL(hlu_permute_rows_kernel<double>, dim3(nblk), dim3(256), 0, st, d_x.as<double>(), d_in.as<double>(), D->perm_t, n, mu, 0);
    else HIP_OK(hipMemcpyAsync(d_in.p, d_x.p, (size_t)count * 8, hipMemcpyDeviceToDevice, st));
    HIP_OK(hipMemcpyAsync(B, d_in.p, (size_t)count * 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    log_message(LOG_DEBUG, strprintf("hierarchical solve: %d refinement steps, last correction %.2e of the solution", f->last_refinements, f->last_correction));
}

// ---- diagnostic entry: a plan executed by the device kernels on HOST arrays (uploaded, run, downloaded) -----------------------
// The counterpart of oracle/hlu_exec.cpp's hluo_run: tests feed both the same arrays, window by window, and compare.
extern "C" int htool_hlu_debug_execute(const htool_hlu_plan *plan_, int first, int last, double *factor, double *diag, int32_t *rank, double *norm0, double *norm2,
                                       int64_t *counters, double *rhs, int64_t ld_rhs, int nrhs, double *scratch);
extern "C" int htool_hlu_debug_execute(const htool_hlu_plan *plan_, int first, int last, double *factor, double *diag, int32_t *rank, double *norm0, double *norm2,
                                       int64_t *counters, double *rhs, int64_t ld_rhs, int nrhs, double *scratch) {
    API_BEGIN
    HM_CHECK(plan_ && plan_->plan, "htool_hlu_debug_execute: null plan");
    HM_CHECK(device_count() > 0, "no HIP device available for the hierarchical LU");
    Plan *P = plan_->plan;
    DeviceHLU f;
    f.plan = P;
    struct Unown { DeviceHLU &f; ~Unown() { f.plan = nullptr; } } unown{f}; // (the plan belongs to the caller)
    HIP_OK(hipGetDevice(&f.device));
    f.n = P->n;
    hlu_allocate(f);
    const size_t nl = P->leaves.size();
    HIP_OK(hipMemcpy(f.factor, factor, (size_t)P->factor_elems * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(f.diag, diag, (size_t)P->diag_elems * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(f.rank, rank, (size_t)P->n_slots * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(f.norm0, norm0, nl * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(f.norm2, norm2, nl * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(f.counters, counters, 8 * sizeof(long long), hipMemcpyHostToDevice));
    hipStream_t st = nullptr;
    HIP_OK(hipStreamCreate(&st));
    try {
        if (first >= 0) {
            HM_CHECK(last < (int)P->factor.size() && first <= last, "htool_hlu_debug_execute: no such window");
            hlu_run_windows(f, first, last, st, scratch);
        } else if (first == -3) {
            if (!P->invert.tasks.empty()) {
                DevBuf it, is;
                it.alloc(P->invert.tasks.size() * sizeof(Task));
                is.alloc(std::max<size_t>(P->invert.seg.size(), 1) * sizeof(int64_t));
                DevProgram dp;
                upload_program(P->invert, it, is, dp, st);
                run_program(P->invert, dp, f.ctx(nullptr, nullptr, 0, 0), P->leaves, st);
                HIP_OK(hipStreamSynchronize(st));
            }
        } else {
            HM_CHECK(rhs != nullptr && ld_rhs >= P->n, "htool_hlu_debug_execute: a solve needs right-hand sides");
            hlu_upload_solves(f);
            DevBuf b;
            b.alloc((size_t)ld_rhs * nrhs * 8);
            HIP_OK(hipMemcpy(b.p, rhs, (size_t)ld_rhs * nrhs * 8, hipMemcpyHostToDevice));
            f.stream = st;
            device_hlu_solve(&f, first == -1 ? 'N' : 'T', b.p, ld_rhs, nrhs, st);
            HIP_OK(hipStreamSynchronize(st));
            HIP_OK(hipMemcpy(rhs, b.p, (size_t)ld_rhs * nrhs * 8, hipMemcpyDeviceToHost));
        }
        HIP_OK(hipStreamSynchronize(st));
    } catch (...) {
        (void)hipDeviceSynchronize();
        (void)hipStreamDestroy(st);
        throw;
    }
    (void)hipStreamDestroy(st);
    HIP_OK(hipMemcpy(factor, f.factor, (size_t)P->factor_elems * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(diag, f.diag, (size_t)P->diag_elems * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(rank, f.rank, (size_t)P->n_slots * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(norm0, f.norm0, nl * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(norm2, f.norm2, nl * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(counters, f.counters, 8 * sizeof(long long), hipMemcpyDeviceToHost));
    API_END
}

// library warm-up (device.hip: device_warm_up): the first launch of a kernel of this translation unit loads its code object
namespace hm {
__global__ void warm_kernel_hlu() {}
void warm_up_hlu() { hipLaunchKernelGGL(warm_kernel_hlu, dim3(1), dim3(64), 0, 0); }
} // namespace hm
