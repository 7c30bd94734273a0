// device.hip -- gfx950 kernels of the H-matrix product and of the panel pack, plus their drivers.
//
// The H-matvec (replaces htool::add_hmatrix_vector_product as called from
// src/htool/hmatrix/hmatrix.hpp:113) is HBM-bound: 2 flops per 8-byte panel element.  It runs as
// four launches over the tile-major layout of hmatrix.hpp:
//   gather_x        W[0:n)   = x[perm_s]                       (user -> cluster numbering)
//   tile_gemv_tall  R        = V-panels * x        (phase A,  one workgroup per source tile)
//   tile_gemv_tall  t        = sum of partials     (phase A2, only leaves spanning several tiles)
//   tile_gemv_wide  y[perm_t]= [U|D]-panels * W[.] (phase B,  one workgroup per row tile)
// Every workgroup streams one contiguous panel with 16-byte loads per lane (64 lanes = 1 KiB per
// wave instruction), 16 loads in flight per wave; coefficients are fetched once per 16/64 columns
// and broadcast with v_readlane; sums are formed in a fixed order (no atomics), so the product is
// bitwise reproducible.
// A symmetric / Hermitian operator stored as one triangle (the reference's 'S' / 'H' storage) is multiplied
// by a fused variant that uses every stored off-diagonal leaf as A and as A^T in one pass over its panels:
//   gather_x -> tile_gemv_tall -> A2 -> tile_gemv_wide_sym (y and the transposed dot products z = U^T x)
//            -> sums of the z partials -> tile_gemv_tall_transposed (y += V^T z) -> finish_sym
// Also here: the pack kernels (per-leaf factors -> tile panels), the device-memory helpers (workspace
// cache of the large temporary buffers), the table assembly and the host drivers.  The device ACA and the
// native build live in device_build.inc, the SVD recompression in device_recompress.inc.
#include "device_internal.hpp"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <numeric>

namespace hm {

// ------------------------------------------------------------------------------------------------
// element traits: a lane always moves one double2 (two real rows, or one complex entry)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double bcast_f64(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

typedef double dvec2_t __attribute__((ext_vector_type(2)));
// 16-byte streaming load of panel data (read exactly once per product): non-temporal hint keeps the
// coefficient workspace W resident in L2 / Infinity Cache instead
__device__ __forceinline__ double2 ldnt16(const void *p) {
    dvec2_t v = __builtin_nontemporal_load((const dvec2_t *)p);
    return make_double2(v.x, v.y);
}

struct RealOps {
    typedef double T;
    static constexpr int RPL = 2; // rows per lane
    static __device__ __forceinline__ T zero() { return 0.0; }
    static __device__ __forceinline__ T bcast(T v, int src) { return bcast_f64(v, src); }
    static __device__ __forceinline__ T shfl(T v, int src) { return __shfl(v, src); } // per-lane source
    static __device__ __forceinline__ void fma(double2 &acc, const double2 v, const T w) {
        acc.x = ::fma(v.x, w, acc.x);
        acc.y = ::fma(v.y, w, acc.y);
    }
    // transposed use: the lane's two rows against their two coefficients
    static __device__ __forceinline__ void tacc(T &acc, const double2 v, const double2 z) { acc = ::fma(v.x, z.x, ::fma(v.y, z.y, acc)); }
    static __device__ __forceinline__ T shfl_xor(T v, int mask) { return __shfl_xor(v, mask); }
    static __device__ __forceinline__ T add(T a, T b) { return a + b; }
    static __device__ __forceinline__ T sel(bool c, T a, T b) { return c ? a : b; }
};

struct CplxOps {
    typedef double2 T;
    static constexpr int RPL = 1;
    static __device__ __forceinline__ T zero() { return make_double2(0.0, 0.0); }
    static __device__ __forceinline__ T bcast(T v, int src) { return make_double2(bcast_f64(v.x, src), bcast_f64(v.y, src)); }
    static __device__ __forceinline__ T shfl(T v, int src) { return make_double2(__shfl(v.x, src), __shfl(v.y, src)); }
    static __device__ __forceinline__ void fma(double2 &acc, const double2 v, const T w) {
        acc.x = ::fma(v.x, w.x, ::fma(-v.y, w.y, acc.x));
        acc.y = ::fma(v.x, w.y, ::fma(v.y, w.x, acc.y));
    }
    static __device__ __forceinline__ void tacc(T &acc, const double2 v, const double2 z) { fma(acc, v, z); } // no conjugation ('S')
    static __device__ __forceinline__ T shfl_xor(T v, int mask) { return make_double2(__shfl_xor(v.x, mask), __shfl_xor(v.y, mask)); }
    static __device__ __forceinline__ T add(T a, T b) { return make_double2(a.x + b.x, a.y + b.y); }
    static __device__ __forceinline__ T sel(bool c, T a, T b) { return make_double2(c ? a.x : b.x, c ? a.y : b.y); }
};

template <typename Ops>
__device__ __forceinline__ void store_rows(typename Ops::T *out, const GTile &tl, int row0, int nrows_total, const double2 acc) {
    if (Ops::RPL == 2) {
        double *o = (double *)out;
        if (row0 < nrows_total) o[tl.omap ? (long long)tl.omap[row0] : tl.out_begin + row0] = acc.x;
        if (row0 + 1 < nrows_total) o[tl.omap ? (long long)tl.omap[row0 + 1] : tl.out_begin + row0 + 1] = acc.y;
    } else {
        double2 *o = (double2 *)out;
        if (row0 < nrows_total) o[tl.omap ? (long long)tl.omap[row0] : tl.out_begin + row0] = acc;
    }
}

// The kernels are templated on NR, the number of right-hand sides multiplied in one sweep of the panels
// (H @ X, src/htool/hmatrix/hmatrix.hpp:134): every panel element is loaded once and used NR times, so the
// arithmetic intensity is 0.25 NR flop/byte and the sweep stays HBM-bound up to NR = 8.  Right-hand side r
// uses W + r * w_stride and writes out + r * out_stride.

// ------------------------------------------------------------------------------------------------
// phase B: few rows (<= 64*RPL), many columns; the four waves split the columns, LDS combine.
// ------------------------------------------------------------------------------------------------
// F = columns handled side by side by one wave instruction: 1 for tiles that fill the 64 lanes, 2 / 4 for
// short tiles (small cluster leaves), where lane group g = lane / (64/F) takes column u*F + g.  Tiles are sorted
// into the three classes at assembly time; every class is its own launch (registers stay those of its code path).
template <typename Ops, int CH, int NR, int F>
__global__ __launch_bounds__(256) void tile_gemv_wide(const GTile *__restrict__ tiles, const GSeg *__restrict__ segs,
                                                      const typename Ops::T *__restrict__ W, typename Ops::T *__restrict__ out,
                                                      long long w_stride, long long out_stride) {
    typedef typename Ops::T T;
    const GTile tl = tiles[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double2 acc[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) acc[r] = make_double2(0.0, 0.0);
    if (F == 1) {
        const int row0 = lane * Ops::RPL;
        const bool active = row0 < tl.nrows;
        for (int s = 0; s < tl.nseg; s++) {
            const GSeg sg = segs[tl.seg_begin + s];
            const T *base = (const T *)sg.panel + row0;
            const long long ld = sg.ld_last;
            const int ncols = sg.ncols;
            // the coefficient INDEX of a batch is fetched one batch ahead, so that the dependent chain index ->
            // coefficient never leaves a wave without loads in flight
            int ci = 0;
            if (wave * CH + lane < ncols && lane < CH) ci = sg.cidx[wave * CH + lane];
            for (int c0 = wave * CH; c0 < ncols; c0 += 4 * CH) {
                const int nc = min(CH, ncols - c0);
                T coef[NR];
#pragma unroll
                for (int r = 0; r < NR; r++) coef[r] = Ops::zero();
                if (lane < nc) {
#pragma unroll
                    for (int r = 0; r < NR; r++) coef[r] = W[r * w_stride + ci];
                }
                {
                    const int c1 = c0 + 4 * CH;
                    if (c1 + lane < ncols && lane < CH) ci = sg.cidx[c1 + lane];
                }
                const T *p = base + (long long)c0 * ld;
                if (nc == CH) {
                    double2 v[CH];
#pragma unroll
                    for (int u = 0; u < CH; u++) v[u] = active ? ldnt16(p + u * ld) : make_double2(0.0, 0.0);
#pragma unroll
                    for (int u = 0; u < CH; u++) {
#pragma unroll
                        for (int r = 0; r < NR; r++) Ops::fma(acc[r], v[u], Ops::bcast(coef[r], u));
                    }
                } else { // tail group: same unrolled batch, loads predicated (coefficients beyond nc are zero)
                    double2 v[CH];
#pragma unroll
                    for (int u = 0; u < CH; u++) v[u] = (active && u < nc) ? ldnt16(p + u * ld) : make_double2(0.0, 0.0);
#pragma unroll
                    for (int u = 0; u < CH; u++) {
#pragma unroll
                        for (int r = 0; r < NR; r++) Ops::fma(acc[r], v[u], Ops::bcast(coef[r], u));
                    }
                }
            }
        }
    } else {
        constexpr int LPG = 64 / F;             // lanes per column group
        constexpr int GC = CH * F;              // columns per batch of CH loads (<= 64)
        const int g = lane / LPG, li = lane - g * LPG;
        const int row0 = li * Ops::RPL;
        const bool active = row0 < tl.nrows;
        for (int s = 0; s < tl.nseg; s++) {
            const GSeg sg = segs[tl.seg_begin + s];
            const long long ld = sg.ld_last;
            const T *base = (const T *)sg.panel + row0 + (long long)g * ld;
            const int ncols = sg.ncols;
            int ci = 0;
            if (wave * GC + lane < ncols && lane < GC) ci = sg.cidx[wave * GC + lane];
            for (int c0 = wave * GC; c0 < ncols; c0 += 4 * GC) {
                const int nc = min(GC, ncols - c0);
                T coef[NR];
#pragma unroll
                for (int r = 0; r < NR; r++) coef[r] = Ops::zero();
                if (lane < nc) {
#pragma unroll
                    for (int r = 0; r < NR; r++) coef[r] = W[r * w_stride + ci];
                }
                {
                    const int c1 = c0 + 4 * GC;
                    if (c1 + lane < ncols && lane < GC) ci = sg.cidx[c1 + lane];
                }
                const T *p = base + (long long)c0 * ld;
                double2 v[CH];
                if (nc == GC) {
#pragma unroll
                    for (int u = 0; u < CH; u++) v[u] = active ? ldnt16(p + (long long)(u * F) * ld) : make_double2(0.0, 0.0);
                } else {
#pragma unroll
                    for (int u = 0; u < CH; u++) v[u] = (active && u * F + g < nc) ? ldnt16(p + (long long)(u * F) * ld) : make_double2(0.0, 0.0);
                }
#pragma unroll
                for (int u = 0; u < CH; u++) {
#pragma unroll
                    for (int r = 0; r < NR; r++) Ops::fma(acc[r], v[u], Ops::shfl(coef[r], u * F + g));
                }
            }
        }
        // add the F column groups (butterfly: every lane ends with the total of its row pair)
#pragma unroll
        for (int off = LPG; off < 64; off <<= 1) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                acc[r].x += __shfl_xor(acc[r].x, off);
                acc[r].y += __shfl_xor(acc[r].y, off);
            }
        }
    }
    const int row0 = lane * Ops::RPL;
    __shared__ double2 red[NR][4][64];
#pragma unroll
    for (int r = 0; r < NR; r++) red[r][wave][lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < NR; r++) {
            double2 a = red[r][0][lane], b = red[r][1][lane], c = red[r][2][lane], d = red[r][3][lane];
            double2 sum = make_double2(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y);
            store_rows<Ops>(out + r * out_stride, tl, row0, tl.nrows, sum);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// phase A / A2: many rows (cut in chunks of 64*RPL), few columns; each wave owns whole row chunks.
// ------------------------------------------------------------------------------------------------
template <typename Ops, int CH, int NR>
__global__ __launch_bounds__(256) void tile_gemv_tall(const GTile *__restrict__ tiles, const GSeg *__restrict__ segs,
                                                      const typename Ops::T *W, typename Ops::T *out, // W and out alias (disjoint regions)
                                                      long long w_stride, long long out_stride, long long panel_stride) {
    // panel_stride: 0 when all right-hand sides share the panel (phase A: V data); the distance between the
    // per-right-hand-side panels otherwise (phase A2: the partial sums live in the coefficient workspace)
    typedef typename Ops::T T;
    constexpr int TM = 64 * Ops::RPL;
    const GTile tl = tiles[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = lane * Ops::RPL;
    // a tile holds one panel per batch (segment); the row chunks of all of them are dealt round-robin to the 4 waves
    int flat = 0;
    for (int s = 0; s < tl.nseg; s++) {
        const GSeg sg = segs[tl.seg_begin + s];
        GTile ot = tl;
        if (sg.oidx) ot.omap = sg.oidx;
        const int nrows = sg.nrows_t;
        const int nq = (nrows + TM - 1) / TM;
        const int ncols = sg.ncols;
        // all row chunks of a panel share the columns, hence the coefficients: with at most 128 columns (always in
        // phase A, where the columns are one source tile) they are fetched once per panel, not per chunk
        const bool hoisted = ncols <= 128;
        T coef1[NR], coef2[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) coef1[r] = coef2[r] = Ops::zero();
        if (hoisted && lane < ncols) {
            const long long ci = sg.cidx[lane];
#pragma unroll
            for (int r = 0; r < NR; r++) coef1[r] = W[r * w_stride + ci];
        }
        if (hoisted && 64 + lane < ncols) {
            const long long ci = sg.cidx[64 + lane];
#pragma unroll
            for (int r = 0; r < NR; r++) coef2[r] = W[r * w_stride + ci];
        }
        for (int q = (wave + 4 - (flat & 3)) & 3; q < nq; q += 4) {
            const int rows_here = min(TM, nrows - q * TM);
            const long long ld = (q == nq - 1) ? sg.ld_last : sg.ld_full;
            const bool active = row0 < rows_here;
            const T *base = (const T *)sg.panel + (long long)q * sg.chunk_stride + row0;
            // where the lane's rows go: fetched now, used after the chunk has been streamed
            const int rr = q * TM + row0;
            long long oi0 = -1, oi1 = -1;
            if (rr < nrows) oi0 = ot.omap ? (long long)ot.omap[rr] : ot.out_begin + rr;
            if (Ops::RPL == 2 && rr + 1 < nrows) oi1 = ot.omap ? (long long)ot.omap[rr + 1] : ot.out_begin + rr + 1;
            double2 acc[NR];
#pragma unroll
            for (int r = 0; r < NR; r++) acc[r] = make_double2(0.0, 0.0);
            for (int c0 = 0; c0 < ncols; c0 += 64) {
                const int nc = min(64, ncols - c0);
                T coef[NR];
#pragma unroll
                for (int r = 0; r < NR; r++) coef[r] = c0 == 0 ? coef1[r] : coef2[r];
                if (!hoisted && lane < nc) {
                    const long long ci = sg.cidx[c0 + lane];
#pragma unroll
                    for (int r = 0; r < NR; r++) coef[r] = W[r * w_stride + ci];
                }
                for (int cc = 0; cc < nc; cc += CH) {
                    const T *p = base + (long long)(c0 + cc) * ld;
                    if (NR > 1 && panel_stride != 0) {
                        for (int u = 0; cc + u < nc && u < CH; u++) {
#pragma unroll
                            for (int r = 0; r < NR; r++) {
                                double2 v = active ? *(const double2 *)(p + u * ld + r * panel_stride) : make_double2(0.0, 0.0);
                                Ops::fma(acc[r], v, Ops::bcast(coef[r], cc + u));
                            }
                        }
                    } else if (cc + CH <= nc) {
                        double2 v[CH];
#pragma unroll
                        for (int u = 0; u < CH; u++) v[u] = active ? ldnt16(p + u * ld) : make_double2(0.0, 0.0);
#pragma unroll
                        for (int u = 0; u < CH; u++) {
#pragma unroll
                            for (int r = 0; r < NR; r++) Ops::fma(acc[r], v[u], Ops::bcast(coef[r], cc + u));
                        }
                    } else { // tail group: same unrolled batch, loads predicated (coefficients beyond nc are zero)
                        double2 v[CH];
#pragma unroll
                        for (int u = 0; u < CH; u++) v[u] = (active && cc + u < nc) ? ldnt16(p + u * ld) : make_double2(0.0, 0.0);
#pragma unroll
                        for (int u = 0; u < CH; u++) {
#pragma unroll
                            for (int r = 0; r < NR; r++) Ops::fma(acc[r], v[u], Ops::bcast(coef[r], (cc + u) & 63));
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (Ops::RPL == 2) {
                    double *o = (double *)(out + r * out_stride);
                    if (oi0 >= 0) o[oi0] = acc[r].x;
                    if (oi1 >= 0) o[oi1] = acc[r].y;
                } else {
                    double2 *o = (double2 *)(out + r * out_stride);
                    if (oi0 >= 0) o[oi0] = acc[r];
                }
            }
        }
        flat += nq;
    }
}

// ------------------------------------------------------------------------------------------------
// one-triangle storage of a symmetric operator ('S' with UPLO): every stored off-diagonal
// leaf is used twice, as A_ts (y_t += A_ts x_s) and transposed (y_s += A_ts^T x_t), from ONE pass over its
// panels.  Besides the plain product a lane also multiplies what it loaded with the x of its own rows; the
// per-lane partials of 16 columns are summed across the 64 lanes by a transpose-reduce (17 cross-lane
// exchanges for 16 sums instead of 96) in a fixed order, so the product stays bitwise reproducible.
// ------------------------------------------------------------------------------------------------
template <typename Ops, int HALF, int BIT>
struct TransposeReduce {
    // in: d[0 .. 2*HALF) per lane; out: d[0 .. HALF) with lanes whose bit BIT is set keeping the upper half
    static __device__ __forceinline__ void step(typename Ops::T *d, int lane) {
        const bool up = (lane & BIT) != 0;
#pragma unroll
        for (int u = 0; u < HALF; u++) {
            const typename Ops::T keep = Ops::sel(up, d[u + HALF], d[u]);
            const typename Ops::T send = Ops::sel(up, d[u], d[u + HALF]);
            d[u] = Ops::add(keep, Ops::shfl_xor(send, BIT));
        }
    }
};
// 16 per-lane partials -> lane L ends with the 64-lane total of partial (L >> 2) & 15 in d[0]
template <typename Ops>
__device__ __forceinline__ void lane_transpose_reduce16(typename Ops::T *d, int lane) {
    TransposeReduce<Ops, 8, 32>::step(d, lane);
    TransposeReduce<Ops, 4, 16>::step(d, lane);
    TransposeReduce<Ops, 2, 8>::step(d, lane);
    TransposeReduce<Ops, 1, 4>::step(d, lane);
    d[0] = Ops::add(d[0], Ops::shfl_xor(d[0], 2));
    d[0] = Ops::add(d[0], Ops::shfl_xor(d[0], 1));
}

// phase B, fused with the transposed use of its columns: out (cluster numbering) = panels * W[cidx], and for every
// column c with zidx[c] >= 0:  Wz[zidx[c]] = sum_i panel[c][i] * x[xoff + i]   (x = W[0 : n), cluster numbering)
template <typename Ops>
__global__ __launch_bounds__(256) void tile_gemv_wide_sym(const GTile *__restrict__ tiles, const GSeg *__restrict__ segs,
                                                          const typename Ops::T *W, typename Ops::T *Wz, typename Ops::T *__restrict__ out, int conj_t) {
    // conj_t: Hermitian operator ('H'): the second use of a leaf is its conjugate transpose
    typedef typename Ops::T T;
    constexpr int CH = 16;
    const GTile tl = tiles[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = lane * Ops::RPL;
    const bool active = row0 < tl.nrows;
    double2 xl = make_double2(0.0, 0.0); // x of the lane's rows
    if (active) {
        if constexpr (Ops::RPL == 2) {
            const double *x = (const double *)W + tl.xoff + row0;
            xl.x = x[0];
            if (row0 + 1 < tl.nrows) xl.y = x[1];
        } else {
            xl = ((const double2 *)W)[tl.xoff + row0];
        }
    }
    double2 acc = make_double2(0.0, 0.0);
    for (int s = 0; s < tl.nseg; s++) {
        const GSeg sg = segs[tl.seg_begin + s];
        const T *base = (const T *)sg.panel + row0;
        const long long ld = sg.ld_last;
        const int ncols = sg.ncols;
        int ci = 0; // coefficient index of the next batch, fetched one batch ahead (see tile_gemv_wide)
        if (wave * CH + lane < ncols && lane < CH) ci = sg.cidx[wave * CH + lane];
        for (int c0 = wave * CH; c0 < ncols; c0 += 4 * CH) {
            const int nc = min(CH, ncols - c0);
            T coef = Ops::zero();
            if (lane < nc) coef = W[ci];
            {
                const int c1 = c0 + 4 * CH;
                if (c1 + lane < ncols && lane < CH) ci = sg.cidx[c1 + lane];
            }
            const T *p = base + (long long)c0 * ld;
            double2 v[CH];
            if (nc == CH) {
#pragma unroll
                for (int u = 0; u < CH; u++) v[u] = active ? ldnt16(p + u * ld) : make_double2(0.0, 0.0);
            } else {
#pragma unroll
                for (int u = 0; u < CH; u++) v[u] = (active && u < nc) ? ldnt16(p + u * ld) : make_double2(0.0, 0.0);
            }
            T d[CH];
#pragma unroll
            for (int u = 0; u < CH; u++) {
                Ops::fma(acc, v[u], Ops::bcast(coef, u));
                d[u] = Ops::zero();
                if (Ops::RPL == 1 && conj_t) v[u].y = -v[u].y;
                Ops::tacc(d[u], v[u], xl);
            }
            lane_transpose_reduce16<Ops>(d, lane);
            const int u = lane >> 2;
            if ((lane & 3) == 0 && u < nc) {
                const int zi = sg.zidx[c0 + u];
                if (zi >= 0) Wz[zi] = d[0];
            }
        }
    }
    __shared__ double2 red[4][64];
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        double2 a = red[0][lane], b = red[1][lane], c = red[2][lane], e = red[3][lane];
        double2 sum = make_double2(((a.x + b.x) + c.x) + e.x, ((a.y + b.y) + c.y) + e.y);
        store_rows<Ops>(out, tl, row0, tl.nrows, sum);
    }
}

// transposed use of the phase-A panels: ycl[out_begin + j] += sum_rows panel[row][j] * W[zidx[row]] for the
// tl.nrows source positions j of the tile, over all segments (one per batch).  Wave w owns columns [w*CG, (w+1)*CG).
template <typename Ops>
__global__ __launch_bounds__(256) void tile_gemv_tall_transposed(const GTile *__restrict__ tiles, const GSeg *__restrict__ segs,
                                                                 const typename Ops::T *__restrict__ W, typename Ops::T *ycl, int conj_t) {
    typedef typename Ops::T T;
    constexpr int TM = 64 * Ops::RPL;
    constexpr int CG = TM / 4; // columns per wave: 32 (real) / 16 (complex)
    const GTile tl = tiles[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = lane * Ops::RPL;
    const int cs = tl.nrows; // source positions of the tile = columns of its panels
    // narrow tiles (at most half the maximal tile) give every wave 16 columns instead of 32 / 16, so that all four work
    const int cgw = (cs <= TM / 2 && CG > 16) ? CG / 2 : CG;
    const int j0 = wave * cgw;
    if (j0 >= cs) return;
    T acc[CG];
#pragma unroll
    for (int u = 0; u < CG; u++) acc[u] = Ops::zero();
    for (int s = 0; s < tl.nseg; s++) {
        const GSeg sg = segs[tl.seg_begin + s];
        const int nr = sg.nrows_t;
        const int nq = (nr + TM - 1) / TM;
        for (int q = 0; q < nq; q++) {
            const int rows_here = min(TM, nr - q * TM);
            const long long ld = (q == nq - 1) ? sg.ld_last : sg.ld_full;
            const bool active = row0 < rows_here;
            double2 z = make_double2(0.0, 0.0); // coefficients of the lane's rows
            bool second = true;
            if (active) {
                const int *zi = sg.zidx + q * TM + row0;
                if constexpr (Ops::RPL == 2) {
                    z.x = ((const double *)W)[zi[0]];
                    second = row0 + 1 < rows_here;
                    if (second) z.y = ((const double *)W)[zi[1]];
                } else {
                    z = ((const double2 *)W)[zi[0]];
                }
            }
            const T *base = (const T *)sg.panel + (long long)q * sg.chunk_stride + row0 + (long long)j0 * ld;
#pragma unroll
            for (int cc = 0; cc < CG; cc += 16) {
                if (cc < cgw) {
                    double2 v[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) v[u] = (active && j0 + cc + u < cs) ? ldnt16(base + (long long)(cc + u) * ld) : make_double2(0.0, 0.0);
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        if (Ops::RPL == 2 && !second) v[u].y = 0.0; // the padding row of an odd last chunk is never written
                        if (Ops::RPL == 1 && conj_t) v[u].y = -v[u].y;
                        Ops::tacc(acc[cc + u], v[u], z);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int cc = 0; cc < CG; cc += 16) {
        if (cc < cgw) {
            lane_transpose_reduce16<Ops>(acc + cc, lane);
            const int j = j0 + cc + (lane >> 2);
            if ((lane & 3) == 0 && j < j0 + cgw && j < cs) {
                T *o = ycl + tl.out_begin + j;
                *o = Ops::add(*o, acc[cc]);
            }
        }
    }
}

// last pass of the one-triangle product: y[map(i)] = ycl[i] + sum of the transposed dense-leaf results that land
// on row i (W[woff + i - tile offset], entries in table order); one workgroup per row tile
template <typename T>
__global__ __launch_bounds__(128) void finish_sym_kernel(const T *__restrict__ ycl, const T *__restrict__ W, const int *__restrict__ zd_ptr,
                                                         const long long *__restrict__ zd_woff, const int *__restrict__ tile_rows,
                                                         const int *__restrict__ perm, T *__restrict__ y) {
    const int r = blockIdx.x, i = threadIdx.x;
    const int off = tile_rows[2 * r], size = tile_rows[2 * r + 1];
    if (i >= size) return;
    T a = ycl[off + i];
    for (int e = zd_ptr[r]; e < zd_ptr[r + 1]; e++) {
        const T v = W[zd_woff[e] + i];
        if constexpr (sizeof(T) == 8) a = a + v;
        else { a.x += v.x; a.y += v.y; }
    }
    y[perm ? (long long)perm[off + i] : (long long)(off + i)] = a;
}

// W[r][i] = X[r][perm[i]] for the nr right-hand sides of one sweep
template <typename T>
__global__ void gather_x_kernel(const T *__restrict__ x, long long x_stride, const int *__restrict__ perm, T *__restrict__ W, long long w_stride, int n, int nr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long src = perm ? perm[i] : i;
    for (int r = 0; r < nr; r++) W[r * w_stride + i] = x[r * x_stride + src];
}

// y[r][map(i)] = sum_s ypart[r][s][i], slices added in order (deterministic); map = perm (user numbering) or identity
template <typename T>
__global__ void reduce_y_kernel(const T *__restrict__ ypart, long long stride, int nslices, int n, const int *__restrict__ perm, T *__restrict__ y,
                                long long y_stride, int nr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long dst = perm ? perm[i] : i;
    for (int r = 0; r < nr; r++) {
        const T *yp = ypart + (long long)r * nslices * stride;
        T acc = yp[i];
        for (int s = 1; s < nslices; s++) {
            T v = yp[(long long)s * stride + i];
            if constexpr (sizeof(T) == 8) acc = acc + v;
            else { acc.x += v.x; acc.y += v.y; }
        }
        y[r * y_stride + dst] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// native generators evaluated on device (cluster-ordered SoA coordinates: x[0:n) y[0:n) z[0:n))
// Same operation order as the CPU restatement (fma chain, sqrt, one division).
// ------------------------------------------------------------------------------------------------
struct DevGen {
    int kind, dim;
    double param;
    const double *tc; // 3*nt
    const double *sc; // 3*ns
    int nt, ns;
};

__device__ __forceinline__ double gen_dist(const DevGen &g, int i, int j) {
    double s = 0;
    for (int k = 0; k < g.dim; k++) {
        double t = g.tc[(long long)k * g.nt + i] - g.sc[(long long)k * g.ns + j];
        s = ::fma(t, t, s);
    }
    return sqrt(s);
}
__device__ __forceinline__ void gen_eval(const DevGen &g, int i, int j, double &out) {
    double r = gen_dist(g, i, j);
    if (g.kind == 0) out = 1.0 / (g.param + r);
    else out = r > 0 ? 1.0 / (4 * M_PI * r) : 0.0;
}
__device__ __forceinline__ void gen_eval(const DevGen &g, int i, int j, double2 &out) {
    double r = gen_dist(g, i, j);
    if (g.kind == 2) {
        if (r > 0) {
            double s, c;
            sincos(g.param * r, &s, &c);
            double q = 1.0 / (4 * M_PI * r);
            out = make_double2(c * q, s * q);
        } else out = make_double2(0.0, 0.0);
    } else {
        double v;
        gen_eval(g, i, j, v);
        out = make_double2(v, 0.0);
    }
}

// ------------------------------------------------------------------------------------------------
// pack: scatter the per-leaf panels of the temporary arena into the tile-major panels
// ------------------------------------------------------------------------------------------------
struct PackArgs {
    const DevBlock *blocks;
    const int *item_block, *item_tile;
    const int *tile_off, *tile_size;
    const int *tile_n;         // b_ncols (phase B) / a_nrows (phase A) of the tile
    const long long *tile_pbase, *tile_ibase;
    void *panel;
    int *index;                // cidxB / oidxA
    int *index2;               // one-triangle storage: zidxB / tidxA (null otherwise)
    const void *arena;
    int vec_rows, tile_max;
    int col_off;               // first source position covered by the H-matrix (coefficient indices are relative to it)
    int eval_dense;            // dense leaves are evaluated from gen instead of copied
    DevGen gen;
};

// REVERSE = false: arena -> tile panels (pack).  REVERSE = true: tile panels -> arena (bulk unpack, used by the
// recompression pass).
template <typename T, bool REVERSE>
__global__ __launch_bounds__(256) void pack_u_kernel(PackArgs a) {
    const int it = blockIdx.x;
    const DevBlock b = a.blocks[a.item_block[it]];
    const int r = a.item_tile[it];
    const int toff = a.tile_off[r], ts = a.tile_size[r];
    const int ld = (ts + a.vec_rows - 1) / a.vec_rows * a.vec_rows;
    const int ncols = b.rank >= 0 ? b.rank : b.n;
    T *dst = (T *)a.panel + a.tile_pbase[r] + (long long)b.ucol * ld;
    int *cidx = a.index + a.tile_ibase[r] + b.ucol;
    T *src = (T *)const_cast<void *>(a.arena) + b.tmp_u + (toff - b.t_off);
    const int total = ncols * ld;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        int k = e / ld, i = e - k * ld;
        if (REVERSE) {
            if (i < ts) src[(long long)k * b.m + i] = dst[(long long)k * ld + i];
            continue;
        }
        T v;
        if (i >= ts) v = T{};
        else if (b.rank < 0 && a.eval_dense) gen_eval(a.gen, toff + i, b.s_off + k, v);
        else v = src[(long long)k * b.m + i];
        dst[(long long)k * ld + i] = v;
        if (i == 0) {
            cidx[k] = b.rank >= 0 ? (int)(b.tpos + k) : b.s_off - a.col_off + k;
            // transposed use: partial (or final) slot of this column for this row tile; diagonal leaves are applied once
            if (a.index2) a.index2[a.tile_ibase[r] + b.ucol + k] = b.t_off == b.s_off ? -1 : (int)(b.z_obase + (long long)(r - b.z_tile0) * b.z_ostride + k);
        }
    }
}

template <typename T, bool REVERSE>
__global__ __launch_bounds__(256) void pack_v_kernel(PackArgs a) {
    const int it = blockIdx.x;
    const DevBlock b = a.blocks[a.item_block[it]];
    const int c = a.item_tile[it];
    const int coff = a.tile_off[c], cs = a.tile_size[c];
    const int TM = a.tile_max;
    const int nrows = a.tile_n[c];
    const int nq = (nrows + TM - 1) / TM;
    const int rem = nrows - (nq - 1) * TM;
    const int ld_last = (rem + a.vec_rows - 1) / a.vec_rows * a.vec_rows;
    T *dst = (T *)a.panel + a.tile_pbase[c];
    int *oidx = a.index + a.tile_ibase[c] + b.vcol;
    T *src = (T *)const_cast<void *>(a.arena) + b.tmp_v + (coff - b.s_off);
    const int p = c - b.v_tile0;
    const int total = b.rank * cs;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        int j = e / b.rank, k = e - j * b.rank;
        int rr = b.vcol + k, q = rr / TM, rl = rr - q * TM;
        int ld = (q == nq - 1) ? ld_last : TM;
        const long long di = (long long)q * cs * TM + (long long)j * ld + rl;
        if (REVERSE) { src[(long long)k * b.n + j] = dst[di]; continue; }
        dst[di] = src[(long long)k * b.n + j];
        if (j == 0) {
            oidx[k] = (int)(b.v_obase + (long long)p * b.v_ostride + k);
            if (a.index2) a.index2[a.tile_ibase[c] + b.vcol + k] = (int)(b.zfin + k); // coefficient of the row in the transposed use
        }
    }
}

// zero the padding rows of the last row chunk of phase-A panels (ld_last > rem) is not needed: the
// padding rows are never stored.  Phase-B panels pad inside pack_u (i >= ts -> 0).

// unpack one leaf for introspection (inverse of pack)
template <typename T>
__global__ void unpack_u_kernel(PackArgs a, T *out) {
    const DevBlock b = a.blocks[0];
    const int r = a.item_tile[blockIdx.x];
    const int toff = a.tile_off[r], ts = a.tile_size[r];
    const int ld = (ts + a.vec_rows - 1) / a.vec_rows * a.vec_rows;
    const int ncols = b.rank >= 0 ? b.rank : b.n;
    const T *src = (const T *)a.panel + a.tile_pbase[r] + (long long)b.ucol * ld;
    for (int e = threadIdx.x; e < ncols * ts; e += blockDim.x) {
        int k = e / ts, i = e - k * ts;
        out[(long long)k * b.m + (toff - b.t_off) + i] = src[(long long)k * ld + i];
    }
}
template <typename T>
__global__ void unpack_v_kernel(PackArgs a, T *out) {
    const DevBlock b = a.blocks[0];
    const int c = a.item_tile[blockIdx.x];
    const int coff = a.tile_off[c], cs = a.tile_size[c];
    const int TM = a.tile_max;
    const int nrows = a.tile_n[c];
    const int nq = (nrows + TM - 1) / TM;
    const int rem = nrows - (nq - 1) * TM;
    const int ld_last = (rem + a.vec_rows - 1) / a.vec_rows * a.vec_rows;
    const T *src = (const T *)a.panel + a.tile_pbase[c];
    for (int e = threadIdx.x; e < b.rank * cs; e += blockDim.x) {
        int j = e / b.rank, k = e - j * b.rank;
        int rr = b.vcol + k, q = rr / TM, rl = rr - q * TM;
        int ld = (q == nq - 1) ? ld_last : TM;
        out[(long long)((coff - b.s_off) + j) * b.rank + k] = src[(long long)q * cs * TM + (long long)j * ld + rl];
    }
}

// ------------------------------------------------------------------------------------------------
// host-side drivers
// ------------------------------------------------------------------------------------------------
int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

static int g_device = 0;
void device_select(int dev) {
    HIP_OK(hipSetDevice(dev));
    g_device = dev;
}

std::string device_name() {
    if (device_count() == 0) return "";
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, g_device) != hipSuccess) return "";
    return std::string(p.gcnArchName) + ":" + p.name;
}

static void require_device() {
    HM_CHECK(device_count() > 0, "no HIP device available: libhtool_mi355x has no CPU fallback (HIP path required)");
    HIP_OK(hipSetDevice(g_device));
}

static hipError_t dev_malloc(void **p, size_t bytes); // hipMalloc that drops the cached workspace and retries when memory is short

template <typename V>
static V *upload(const std::vector<V> &h, size_t *bytes = nullptr) {
    V *d = nullptr;
    size_t n = std::max<size_t>(h.size(), 1) * sizeof(V);
    HIP_OK(dev_malloc((void **)&d, n));
    if (!h.empty()) HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(V), hipMemcpyHostToDevice));
    if (bytes) *bytes += n;
    return d;
}

// ------------------------------------------------------------------------------------------------
// device memory helpers
// ------------------------------------------------------------------------------------------------
// Large temporary buffers (the ACA arena of a build, its staging buffer, the arena of a recompression or of a bulk
// download) are kept in a process-wide cache instead of being freed: on this runtime the first allocation after a large
// hipFree waits until the driver has scrubbed the freed memory (~30 ms per GB: 2.4 s after a 1 M-point build), and
// build -> recompression -> next build would pay that every time.  The cache holds one buffer per slot; it is dropped
// by htool_release_workspace(), and automatically when an allocation fails.
class WorkspaceCache {
public:
    enum { ARENA = 0, FLAGS = 1, STAGE = 2, NSLOT = 3 };
    // a buffer of at least `bytes` on the current device, or nullptr when the slot is in use (concurrent builds): the
    // caller then allocates privately
    void *acquire(int slot, size_t bytes) {
        std::lock_guard<std::mutex> lock(mu_);
        Slot &s = slots_[slot];
        if (s.busy) return nullptr;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (s.p && (s.bytes < bytes || s.device != dev)) { (void)hipFree(s.p); s.p = nullptr; s.bytes = 0; }
        if (!s.p) {
            const size_t want = std::max<size_t>(bytes, 1);
            if (hipMalloc(&s.p, want) != hipSuccess) { (void)hipGetLastError(); s.p = nullptr; return nullptr; }
            s.bytes = want;
            s.device = dev;
        }
        s.busy = true;
        return s.p;
    }
    void give_back(int slot) {
        std::lock_guard<std::mutex> lock(mu_);
        slots_[slot].busy = false;
    }
    size_t release_all() { // frees what is not in use; returns the bytes freed
        std::lock_guard<std::mutex> lock(mu_);
        size_t freed = 0;
        for (Slot &s : slots_)
            if (s.p && !s.busy) { (void)hipFree(s.p); freed += s.bytes; s.p = nullptr; s.bytes = 0; }
        return freed;
    }
    size_t slot_bytes(int slot) { // size of the idle buffer cached in a slot (0: none, or in use)
        std::lock_guard<std::mutex> lock(mu_);
        return slots_[slot].busy ? 0 : slots_[slot].bytes;
    }
    size_t cached_bytes() {
        std::lock_guard<std::mutex> lock(mu_);
        size_t b = 0;
        for (Slot &s : slots_) b += s.bytes;
        return b;
    }

private:
    struct Slot { void *p = nullptr; size_t bytes = 0; int device = -1; bool busy = false; };
    Slot slots_[NSLOT];
    std::mutex mu_;
};
static WorkspaceCache g_workspace;

size_t device_release_workspace() { return g_workspace.release_all(); }
size_t device_workspace_bytes() { return g_workspace.cached_bytes(); }

// hipMalloc that gives the cached workspace back to the driver and retries once when memory is short
static hipError_t dev_malloc(void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
        (void)hipGetLastError();
        if (g_workspace.release_all() > 0) e = hipMalloc(p, bytes);
    }
    return e;
}

// one leased workspace buffer: from the cache when the slot is free, private otherwise; returned / freed on scope exit
struct WorkspaceLease {
    int slot;
    void *p = nullptr;
    bool cached = false;
    explicit WorkspaceLease(int s) : slot(s) {}
    WorkspaceLease(const WorkspaceLease &) = delete;
    WorkspaceLease &operator=(const WorkspaceLease &) = delete;
    void *get(size_t bytes) { // (re)size: the previous content is lost
        drop();
        p = g_workspace.acquire(slot, bytes);
        cached = p != nullptr;
        if (!p) HIP_OK(dev_malloc(&p, std::max<size_t>(bytes, 1)));
        return p;
    }
    void drop() {
        if (!p) return;
        if (cached) g_workspace.give_back(slot);
        else (void)hipFree(p);
        p = nullptr;
    }
    ~WorkspaceLease() { drop(); }
};

// scope guard for temporary device buffers (freed on every exit path, exceptions included)
struct TempPool {
    std::vector<void *> ptrs;
    template <typename V>
    V *up(const std::vector<V> &h) {
        V *d = upload(h);
        ptrs.push_back((void *)d);
        return d;
    }
    void *alloc(size_t bytes) {
        void *d = nullptr;
        HIP_OK(dev_malloc(&d, std::max<size_t>(bytes, 1)));
        ptrs.push_back(d);
        return d;
    }
    void release(void *p) { // free one buffer early
        for (auto &q : ptrs) if (q == p) { (void)hipFree(q); q = nullptr; }
    }
    ~TempPool() { for (void *p : ptrs) if (p) (void)hipFree(p); }
};

static DevBlock to_dev(const BlockRec &b, const HMatrix &H) {
    DevBlock d;
    d.tmp_u = b.tmp_u; d.tmp_v = b.tmp_v; d.tpos = b.tpos; d.v_obase = b.v_obase;
    d.t_off = b.t_off; d.m = b.m; d.s_off = b.s_off; d.n = b.n; d.rank = b.rank; d.cap = b.cap;
    d.ucol = b.ucol; d.vcol = b.vcol; d.v_ostride = b.v_ostride;
    d.v_tile0 = H.ctiles.node_tile_begin[b.s_node];
    d.status = b.status; d.pad_ = 0;
    d.z_obase = b.z_obase; d.zfin = b.zfin; d.z_ostride = b.z_ostride;
    d.z_tile0 = H.rtiles.node_tile_begin[b.t_node];
    return d;
}

struct DeviceBuilder {
    HMatrix &H;
    DeviceHMatrix *D;
    int vec_rows;
    int *d_rt_off = nullptr, *d_rt_size = nullptr, *d_ct_off = nullptr, *d_ct_size = nullptr;
    DevGen gen{};
    bool have_gen = false;
    std::vector<void *> deferred_free;

    explicit DeviceBuilder(HMatrix &h) : H(h) {
        require_device();
        D = new DeviceHMatrix;
        H.dev = D;
        D->device = g_device;
        D->is_complex = H.is_complex;
        D->esize = H.is_complex ? 16 : 8;
        vec_rows = H.is_complex ? 1 : 2;
        D->n_source = H.col_size;
        D->n_target = H.tc->n_points;
        D->row_off = H.row_off;
        D->row_size = H.row_size;
        HIP_OK(hipStreamCreate(&D->stream));
        for (auto &slot : D->pev) for (auto &e : slot) HIP_OK(hipEventCreate(&e));
        d_rt_off = upload(H.rtiles.off);
        d_rt_size = upload(H.rtiles.size);
        d_ct_off = upload(H.ctiles.off);
        d_ct_size = upload(H.ctiles.size);
    }
    // attach to an existing device H-matrix (recompression re-packs its batches in place)
    DeviceBuilder(HMatrix &h, DeviceHMatrix *existing) : H(h), D(existing) {
        require_device();
        HIP_OK(hipSetDevice(D->device));
        vec_rows = H.is_complex ? 1 : 2;
        d_rt_off = upload(H.rtiles.off);
        d_rt_size = upload(H.rtiles.size);
        d_ct_off = upload(H.ctiles.off);
        d_ct_size = upload(H.ctiles.size);
    }
    ~DeviceBuilder() {
        for (void *q : deferred_free) (void)hipFree(q);
        (void)hipFree(d_rt_off); (void)hipFree(d_rt_size); (void)hipFree(d_ct_off); (void)hipFree(d_ct_size);
    }

    // pack one batch whose leaf panels sit in d_arena (device)
    // replace_index >= 0: the new panels take the place of an existing batch (whose buffers are freed)
    template <typename T>
    void pack_batch(const std::vector<int64_t> &batch_blocks, const void *d_arena, bool eval_dense, int replace_index = -1) {
        BatchLayout L;
        L.batch_id = replace_index >= 0 ? replace_index : (int)D->batches.size();
        double tl0 = wall_seconds();
        compute_batch_layout(H, batch_blocks, vec_rows, L);
        double tl1 = wall_seconds();
        for (int64_t bi : batch_blocks) H.blocks[bi].batch = L.batch_id;
        DevBatch B;
        size_t szB = std::max<int64_t>(L.panelB_elems, 1) * sizeof(T), szA = std::max<int64_t>(L.panelA_elems, 1) * sizeof(T);
        TempPool owned; // the batch's own buffers, handed over to D at the end
        B.panelB = owned.alloc(szB);
        B.panelA = owned.alloc(szA);
        B.cidxB = (int *)owned.alloc(std::max<int64_t>(L.cidxB_elems, 1) * sizeof(int));
        B.oidxA = (int *)owned.alloc(std::max<int64_t>(L.oidxA_elems, 1) * sizeof(int));
        B.bytes = szB + szA + (L.cidxB_elems + L.oidxA_elems) * sizeof(int);
        if (H.one_triangle) {
            B.zidxB = (int *)owned.alloc(std::max<int64_t>(L.cidxB_elems, 1) * sizeof(int));
            B.tidxA = (int *)owned.alloc(std::max<int64_t>(L.oidxA_elems, 1) * sizeof(int));
            B.bytes += (L.cidxB_elems + L.oidxA_elems) * sizeof(int);
        }
        std::vector<DevBlock> hb(batch_blocks.size());
        for (size_t q = 0; q < batch_blocks.size(); q++) hb[q] = to_dev(H.blocks[batch_blocks[q]], H);
        TempPool tmp;
        DevBlock *d_blocks = tmp.up(hb);
        int *d_ub = tmp.up(L.u_item_block), *d_ut = tmp.up(L.u_item_tile), *d_vb = tmp.up(L.v_item_block), *d_vt = tmp.up(L.v_item_tile);
        int *d_bn = tmp.up(L.b_ncols), *d_an = tmp.up(L.a_nrows);
        std::vector<long long> t1(L.b_pbase.begin(), L.b_pbase.end()), t2(L.b_cbase.begin(), L.b_cbase.end());
        std::vector<long long> t3(L.a_pbase.begin(), L.a_pbase.end()), t4(L.a_obase.begin(), L.a_obase.end());
        long long *d_bp = tmp.up(t1), *d_bc = tmp.up(t2), *d_ap = tmp.up(t3), *d_ao = tmp.up(t4);
        PackArgs a;
        a.blocks = d_blocks; a.arena = d_arena; a.vec_rows = vec_rows; a.tile_max = H.tile_max; a.col_off = H.col_off;
        a.eval_dense = eval_dense ? 1 : 0; a.gen = gen;
        // U / dense side
        a.item_block = d_ub; a.item_tile = d_ut; a.tile_off = d_rt_off; a.tile_size = d_rt_size; a.tile_n = d_bn;
        a.tile_pbase = d_bp; a.tile_ibase = d_bc; a.panel = B.panelB; a.index = B.cidxB; a.index2 = B.zidxB;
        if (!L.u_item_block.empty()) hipLaunchKernelGGL((pack_u_kernel<T, false>), dim3((unsigned)L.u_item_block.size()), dim3(256), 0, D->stream, a);
        // V side
        a.item_block = d_vb; a.item_tile = d_vt; a.tile_off = d_ct_off; a.tile_size = d_ct_size; a.tile_n = d_an;
        a.tile_pbase = d_ap; a.tile_ibase = d_ao; a.panel = B.panelA; a.index = B.oidxA; a.index2 = B.tidxA;
        if (!L.v_item_block.empty()) hipLaunchKernelGGL((pack_v_kernel<T, false>), dim3((unsigned)L.v_item_block.size()), dim3(256), 0, D->stream, a);
        HIP_OK(hipGetLastError());
        double tl2 = wall_seconds();
        HIP_OK(hipStreamSynchronize(D->stream));
        log_message(LOG_DEBUG, strprintf("pack batch %d: layout %.3f s, alloc+upload %.3f s, kernels %.3f s, panels %.2f GB", L.batch_id, tl1 - tl0, tl2 - tl1, wall_seconds() - tl2, (szA + szB) / 1e9));
        BatchTables bt;
        bt.b_ncols.swap(L.b_ncols); bt.a_nrows.swap(L.a_nrows);
        bt.b_pbase.swap(L.b_pbase); bt.b_cbase.swap(L.b_cbase); bt.a_pbase.swap(L.a_pbase); bt.a_obase.swap(L.a_obase);
        bt.reduces.swap(L.reduces);
        bt.z_reduces.swap(L.z_reduces); bt.zd_tile.swap(L.zd_tile); bt.zd_woff.swap(L.zd_woff);
        bt.r_end = H.r_elems;
        owned.ptrs.clear(); // success: ownership moves to the device H-matrix
        if (replace_index >= 0) {
            DevBatch &old = D->batches[replace_index];
            // With memory to spare the old buffers are released when the builder goes away, i.e. after the tables have
            // been re-assembled (an allocation issued right after a large free waits for the driver to scrub the freed
            // memory); when memory is tight they go now.
            size_t free_now = 0, total_now = 0;
            HIP_OK(hipMemGetInfo(&free_now, &total_now));
            const bool defer = free_now > 2 * old.bytes + ((size_t)16 << 30);
            for (void *q : {old.panelB, old.panelA, (void *)old.cidxB, (void *)old.oidxA, (void *)old.zidxB, (void *)old.tidxA}) {
                if (!q) continue;
                if (defer) deferred_free.push_back(q);
                else (void)hipFree(q);
            }
            old = B;
            D->tabs[replace_index] = std::move(bt);
        } else {
            D->batches.push_back(B);
            D->tabs.push_back(std::move(bt));
        }
    }

    // bulk unpack: copy the panels of the given leaves of batch `bidx` back into an arena (tmp_u / tmp_v of the
    // blocks say where); the inverse of pack_batch, driven by the batch's stored tables
    template <typename T>
    void unpack_batch(const std::vector<int64_t> &batch_blocks, int bidx, void *d_arena) {
        const BatchTables &bt = D->tabs[bidx];
        std::vector<DevBlock> hb(batch_blocks.size());
        std::vector<int> ub, ut, vb, vt;
        for (size_t q = 0; q < batch_blocks.size(); q++) {
            const BlockRec &b = H.blocks[batch_blocks[q]];
            hb[q] = to_dev(b, H);
            for (int r = H.rtiles.node_tile_begin[b.t_node]; r < H.rtiles.node_tile_end[b.t_node]; r++) { ub.push_back((int)q); ut.push_back(r); }
            if (b.rank >= 0)
                for (int c = H.ctiles.node_tile_begin[b.s_node]; c < H.ctiles.node_tile_end[b.s_node]; c++) { vb.push_back((int)q); vt.push_back(c); }
        }
        TempPool tmp;
        DevBlock *d_blocks = tmp.up(hb);
        int *d_ub = tmp.up(ub), *d_ut = tmp.up(ut), *d_vb = tmp.up(vb), *d_vt = tmp.up(vt), *d_an = tmp.up(bt.a_nrows);
        std::vector<long long> t1(bt.b_pbase.begin(), bt.b_pbase.end()), t3(bt.a_pbase.begin(), bt.a_pbase.end());
        long long *d_bp = tmp.up(t1), *d_ap = tmp.up(t3);
        PackArgs a;
        std::memset(&a, 0, sizeof(a));
        a.blocks = d_blocks; a.arena = d_arena; a.vec_rows = vec_rows; a.tile_max = H.tile_max; a.col_off = H.col_off;
        a.item_block = d_ub; a.item_tile = d_ut; a.tile_off = d_rt_off; a.tile_size = d_rt_size; a.tile_pbase = d_bp; a.tile_ibase = d_bp;
        a.panel = D->batches[bidx].panelB; a.index = D->batches[bidx].cidxB;
        if (!ub.empty()) hipLaunchKernelGGL((pack_u_kernel<T, true>), dim3((unsigned)ub.size()), dim3(256), 0, D->stream, a);
        a.item_block = d_vb; a.item_tile = d_vt; a.tile_off = d_ct_off; a.tile_size = d_ct_size; a.tile_n = d_an; a.tile_pbase = d_ap; a.tile_ibase = d_ap;
        a.panel = D->batches[bidx].panelA; a.index = D->batches[bidx].oidxA;
        if (!vb.empty()) hipLaunchKernelGGL((pack_v_kernel<T, true>), dim3((unsigned)vb.size()), dim3(256), 0, D->stream, a);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(D->stream));
    }

    // drop W and the product tables (they are rebuilt by assemble())
    void free_product_tables() {
        for (void *p : {(void *)D->segs, (void *)D->tilesB_user, (void *)D->tilesB_cluster, (void *)D->tilesA, (void *)D->tilesA2, (void *)D->tilesB_split,
                        (void *)D->perm_s, (void *)D->perm_t, (void *)D->iota, (void *)D->ones_idx, D->W, D->x_tmp, D->y_tmp, D->ypart,
                        (void *)D->tilesAT, (void *)D->tilesZ, (void *)D->zd_ptr, (void *)D->zd_woff, (void *)D->zd_rows, D->ycl})
            if (p) (void)hipFree(p);
        D->tilesAT = D->tilesZ = nullptr; D->zd_ptr = D->zd_rows = nullptr; D->zd_woff = nullptr; D->ycl = nullptr;
        D->nAT = D->nZ = D->n_zd_tiles = 0;
        D->segs = nullptr; D->tilesB_user = D->tilesB_cluster = D->tilesA = D->tilesA2 = D->tilesB_split = nullptr;
        D->perm_s = D->perm_t = D->iota = D->ones_idx = nullptr;
        D->W = D->x_tmp = D->y_tmp = D->ypart = nullptr;
        D->nB = D->nA = D->nA2 = D->nB_split = 0;
        D->splitB = 1;
        D->rhs_cap = 0;
        D->table_bytes = 0;
    }

    // build W, the permutation tables and the tile/segment tables of the three product phases
    template <typename T>
    void assemble() {
        const ClusterTree &Tt = *H.tc, &Ss = *H.sc;
        const int TM = H.tile_max, Ns = H.col_size;
        std::vector<BatchTables> &tabs = D->tabs;
        const long long r_start = (Ns + 1 + 1) / 2 * 2;
        D->W_elems = (r_start + H.r_elems + 2 + 1) / 2 * 2; // even: every right-hand-side copy stays 16-byte aligned
        HM_CHECK(D->W_elems < (1LL << 31), "coefficient workspace exceeds the 32-bit index range of the panel index arrays");
        HIP_OK(dev_malloc(&D->W, D->W_elems * sizeof(T)));
        HIP_OK(hipMemset(D->W, 0, D->W_elems * sizeof(T)));
        T one;
        std::memset(&one, 0, sizeof(T));
        *(double *)&one = 1.0;
        HIP_OK(hipMemcpy((char *)D->W + (size_t)Ns * sizeof(T), &one, sizeof(T), hipMemcpyHostToDevice));
        D->rhs_cap = 1;
        D->perm_s = upload(std::vector<int>(Ss.perm.begin() + H.col_off, Ss.perm.begin() + H.col_off + H.col_size), &D->table_bytes);
        D->perm_t = upload(Tt.perm, &D->table_bytes);
        std::vector<int> io(Ns);
        std::iota(io.begin(), io.end(), 0);
        D->iota = upload(io, &D->table_bytes);
        int maxP = 1;
        for (auto &bt : tabs) for (auto &r : bt.reduces) maxP = std::max(maxP, r.ncols);
        for (auto &bt : tabs) for (auto &r : bt.z_reduces) maxP = std::max(maxP, r.ncols);
        D->one_triangle = H.one_triangle;
        D->conj_transposed = H.one_triangle && H.is_complex && H.params.symmetry == 'H';
        std::vector<int> ones(maxP, Ns);
        D->ones_idx = upload(ones, &D->table_bytes);
        HIP_OK(dev_malloc(&D->x_tmp, (size_t)std::max(Ns, 1) * sizeof(T)));
        HIP_OK(dev_malloc(&D->y_tmp, (size_t)std::max(Tt.n_points, 1) * sizeof(T)));

        std::vector<GSeg> segs;
        std::vector<GTile> tB, tBc, tA, tA2;
        std::vector<double> wB, wA, wA2; // work, for heavy-first ordering
        const int nrt = H.rtiles.count(), nct = H.ctiles.count();
        for (int r = 0; r < nrt; r++) {
            GTile t;
            t.seg_begin = (long long)segs.size();
            t.nseg = 0;
            t.nrows = H.rtiles.size[r];
            int ld = (t.nrows + vec_rows - 1) / vec_rows * vec_rows;
            double work = 0;
            for (size_t b = 0; b < tabs.size(); b++) {
                int nc = tabs[b].b_ncols[r];
                if (nc == 0) continue;
                GSeg s;
                s.panel = (const char *)D->batches[b].panelB + (size_t)tabs[b].b_pbase[r] * sizeof(T);
                s.cidx = D->batches[b].cidxB + tabs[b].b_cbase[r];
                s.ncols = nc; s.ld_full = ld; s.ld_last = ld; s.nrows_t = 0; s.chunk_stride = 0;
                if (H.one_triangle) s.zidx = D->batches[b].zidxB + tabs[b].b_cbase[r];
                segs.push_back(s);
                t.nseg++;
                work += (double)nc * ld;
            }
            t.omap = D->perm_t + H.rtiles.off[r];
            t.out_begin = 0;
            t.xoff = H.rtiles.off[r];
            tB.push_back(t);
            GTile tc = t;
            tc.omap = nullptr;
            tc.out_begin = H.rtiles.off[r] - H.row_off;
            tBc.push_back(tc);
            wB.push_back(work);
        }
        const int target_items = 256 * 16; // workgroups wanted per launch (256 CUs)
        long long total_chunks = 0;
        for (size_t b = 0; b < tabs.size(); b++)
            for (int c = 0; c < nct; c++) total_chunks += (tabs[b].a_nrows[c] + TM - 1) / TM;
        const int qmax = (int)std::max<long long>(4, ((total_chunks + target_items - 1) / target_items + 3) / 4 * 4);
        // phase A tiles.  Many source tiles (large operators): ONE workgroup per source tile streams the panels of all
        // batches (a segment each), so that its four waves stay busy even when a batch holds only a chunk or two of the
        // tile.  Few source tiles: one workgroup per (batch, piece of at most qmax row chunks) for parallelism.
        const bool merge_batches = nct >= target_items / 2;
        auto tall_segment = [&](size_t b, int c, int nr) {
            const int nq = (nr + TM - 1) / TM, rem = nr - (nq - 1) * TM;
            GSeg s;
            s.panel = (const char *)D->batches[b].panelA + (size_t)tabs[b].a_pbase[c] * sizeof(T);
            s.cidx = D->iota + (H.ctiles.off[c] - H.col_off);
            s.ncols = H.ctiles.size[c]; s.ld_full = TM; s.ld_last = (rem + vec_rows - 1) / vec_rows * vec_rows; s.nrows_t = nr;
            s.chunk_stride = (long long)H.ctiles.size[c] * TM;
            s.oidx = D->batches[b].oidxA + tabs[b].a_obase[c];
            return s;
        };
        if (merge_batches) {
            for (int c = 0; c < nct; c++) {
                GTile t;
                t.seg_begin = (long long)segs.size(); t.nseg = 0; t.nrows = 0; t.omap = nullptr; t.out_begin = 0;
                double work = 0;
                for (size_t b = 0; b < tabs.size(); b++) {
                    const int nr = tabs[b].a_nrows[c];
                    if (nr == 0) continue;
                    segs.push_back(tall_segment(b, c, nr));
                    t.nseg++;
                    t.nrows += nr;
                    work += (double)nr * H.ctiles.size[c];
                }
                if (t.nseg) { tA.push_back(t); wA.push_back(work); }
            }
        } else {
            for (size_t b = 0; b < tabs.size(); b++)
                for (int c = 0; c < nct; c++) {
                    const int nr = tabs[b].a_nrows[c];
                    if (nr == 0) continue;
                    const GSeg s = tall_segment(b, c, nr);
                    const int nq = (nr + TM - 1) / TM;
                    // cut tall tiles into pieces of at most qmax row chunks (independent outputs, no reduction)
                    for (int q0 = 0; q0 < nq; q0 += qmax) {
                        const int q1 = std::min(nq, q0 + qmax);
                        GSeg sp = s;
                        sp.panel = (const char *)s.panel + (size_t)q0 * (size_t)s.chunk_stride * sizeof(T);
                        if (q1 < nq) sp.ld_last = TM;
                        sp.nrows_t = (q1 < nq ? q1 * TM : nr) - q0 * TM;
                        sp.oidx = s.oidx + (long long)q0 * TM;
                        GTile t;
                        t.seg_begin = (long long)segs.size(); t.nseg = 1;
                        t.nrows = sp.nrows_t;
                        t.omap = nullptr; t.out_begin = 0;
                        segs.push_back(sp);
                        tA.push_back(t);
                        wA.push_back((double)t.nrows * s.ncols);
                    }
                }
        }
        for (size_t b = 0; b < tabs.size(); b++)
            for (auto &r : tabs[b].reduces) {
                GSeg s;
                s.panel = (const char *)D->W + (size_t)r.w_panel * sizeof(T);
                s.cidx = D->ones_idx; s.ncols = r.ncols; s.ld_full = r.ld; s.ld_last = r.ld; s.nrows_t = r.nrows; s.chunk_stride = TM;
                GTile t;
                t.seg_begin = (long long)segs.size(); t.nseg = 1; t.nrows = r.nrows; t.omap = nullptr; t.out_begin = r.out_base;
                segs.push_back(s);
                tA2.push_back(t);
                wA2.push_back((double)r.nrows * r.ncols);
            }
        auto sort_heavy = [](std::vector<GTile> &t, const std::vector<double> &w, std::vector<GTile> *twin) {
            std::vector<int> idx(t.size());
            std::iota(idx.begin(), idx.end(), 0);
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return w[a] > w[b]; });
            std::vector<GTile> o(t.size()), o2(t.size());
            for (size_t i = 0; i < idx.size(); i++) { o[i] = t[idx[i]]; if (twin) o2[i] = (*twin)[idx[i]]; }
            t.swap(o);
            if (twin) twin->swap(o2);
        };
        // phase B column slices for small operators
        std::vector<GTile> tBs;
        std::vector<double> wBs;
        D->splitB = std::max(1, std::min(8, (target_items + std::max(nrt, 1) - 1) / std::max(nrt, 1)));
        if (D->splitB > 1) {
            const int S = D->splitB;
            D->ypart_stride = ((long long)H.row_size + 1) / 2 * 2;
            HIP_OK(dev_malloc(&D->ypart, (size_t)S * D->ypart_stride * sizeof(T)));
            D->table_bytes += (size_t)S * D->ypart_stride * sizeof(T);
            for (int r = 0; r < nrt; r++) {
                const GTile &t0 = tBc[r];
                long long C = 0;
                for (int q = 0; q < t0.nseg; q++) C += segs[t0.seg_begin + q].ncols;
                // slice boundaries in units of 16 columns (the kernel's unrolled chunk)
                const long long per = ((C + S - 1) / S + 15) / 16 * 16;
                for (int sl = 0; sl < S; sl++) {
                    const long long lo = std::min<long long>(C, sl * per), hi = std::min<long long>(C, (sl + 1) * per);
                    GTile t = t0;
                    t.seg_begin = (long long)segs.size();
                    t.nseg = 0;
                    t.omap = nullptr;
                    t.out_begin = (long long)sl * D->ypart_stride + (H.rtiles.off[r] - H.row_off);
                    long long pos = 0;
                    double work = 0;
                    for (int q = 0; q < t0.nseg; q++) {
                        const GSeg sg = segs[t0.seg_begin + q];
                        const long long a = std::max(lo, pos), bnd = std::min(hi, pos + sg.ncols);
                        if (a < bnd) {
                            GSeg sp = sg;
                            sp.panel = (const char *)sg.panel + (size_t)(a - pos) * (size_t)sg.ld_last * sizeof(T);
                            sp.cidx = sg.cidx + (a - pos);
                            if (sg.zidx) sp.zidx = sg.zidx + (a - pos);
                            sp.ncols = (int)(bnd - a);
                            segs.push_back(sp);
                            t.nseg++;
                            work += (double)sp.ncols * sg.ld_last;
                        }
                        pos += sg.ncols;
                    }
                    tBs.push_back(t);
                    wBs.push_back(work);
                }
            }
            sort_heavy(tBs, wBs, nullptr);
        }
        sort_heavy(tB, wB, &tBc);
        // classes of the wide kernel (columns per wave instruction F = 1, 2, 4), heavy-first inside each class
        auto tile_class = [&](const GTile &t) {
            if (H.one_triangle) return 0; // one kernel (tile_gemv_wide_sym) for every row tile
            int lanes = (t.nrows + vec_rows - 1) / vec_rows;
            return lanes <= 16 ? 2 : (lanes <= 32 ? 1 : 0);
        };
        auto by_class = [&](std::vector<GTile> &t, std::vector<GTile> *twin, int *cnt) {
            std::vector<GTile> o, o2;
            for (int c = 0; c < 3; c++) {
                cnt[c] = 0;
                for (size_t i = 0; i < t.size(); i++) if (tile_class(t[i]) == c) { o.push_back(t[i]); if (twin) o2.push_back((*twin)[i]); cnt[c]++; }
            }
            t.swap(o);
            if (twin) twin->swap(o2);
        };
        by_class(tB, &tBc, D->cntB);
        by_class(tBs, nullptr, D->cntBs);
        sort_heavy(tA, wA, nullptr);
        sort_heavy(tA2, wA2, nullptr);
        if (H.one_triangle) {
            // transposed use of the phase-A panels: one tile per source tile, one segment per batch (a tile owns its rows of y)
            std::vector<GTile> tAT, tZ;
            std::vector<double> wAT, wZ;
            for (int c = 0; c < nct; c++) {
                GTile t;
                t.seg_begin = (long long)segs.size(); t.nseg = 0;
                t.nrows = H.ctiles.size[c];
                t.omap = nullptr; t.out_begin = H.ctiles.off[c];
                double work = 0;
                for (size_t b = 0; b < tabs.size(); b++) {
                    const int nr = tabs[b].a_nrows[c];
                    if (nr == 0) continue;
                    const int nq = (nr + TM - 1) / TM, rem = nr - (nq - 1) * TM;
                    GSeg s;
                    s.panel = (const char *)D->batches[b].panelA + (size_t)tabs[b].a_pbase[c] * sizeof(T);
                    s.cidx = D->iota;
                    s.ncols = H.ctiles.size[c]; s.ld_full = TM; s.ld_last = (rem + vec_rows - 1) / vec_rows * vec_rows;
                    s.nrows_t = nr;
                    s.chunk_stride = (long long)H.ctiles.size[c] * TM;
                    s.zidx = D->batches[b].tidxA + tabs[b].a_obase[c];
                    segs.push_back(s);
                    t.nseg++;
                    work += (double)nr * s.ncols;
                }
                if (t.nseg) { tAT.push_back(t); wAT.push_back(work); }
            }
            for (size_t b = 0; b < tabs.size(); b++)
                for (auto &r : tabs[b].z_reduces) {
                    GSeg s;
                    s.panel = (const char *)D->W + (size_t)r.w_panel * sizeof(T);
                    s.cidx = D->ones_idx; s.ncols = r.ncols; s.ld_full = r.ld; s.ld_last = r.ld; s.nrows_t = r.nrows; s.chunk_stride = TM;
                    GTile t;
                    t.seg_begin = (long long)segs.size(); t.nseg = 1; t.nrows = r.nrows; t.omap = nullptr; t.out_begin = r.out_base;
                    segs.push_back(s);
                    tZ.push_back(t);
                    wZ.push_back((double)r.nrows * r.ncols);
                }
            sort_heavy(tAT, wAT, nullptr);
            sort_heavy(tZ, wZ, nullptr);
            D->tilesAT = upload(tAT, &D->table_bytes);
            D->tilesZ = upload(tZ, &D->table_bytes);
            D->nAT = (int)tAT.size(); D->nZ = (int)tZ.size();
            // transposed dense leaves: per row tile of y the W offsets to add (batch order, then leaf order)
            std::vector<int> zptr(nrt + 1, 0), rows(2 * (size_t)nrt);
            for (auto &bt : tabs) for (int c : bt.zd_tile) zptr[c + 1]++;
            for (int r = 0; r < nrt; r++) { zptr[r + 1] += zptr[r]; rows[2 * r] = H.rtiles.off[r]; rows[2 * r + 1] = H.rtiles.size[r]; }
            std::vector<long long> zw((size_t)zptr[nrt]);
            std::vector<int> fill(zptr.begin(), zptr.end() - 1);
            for (auto &bt : tabs) for (size_t e = 0; e < bt.zd_tile.size(); e++) zw[fill[bt.zd_tile[e]]++] = bt.zd_woff[e];
            D->zd_ptr = upload(zptr, &D->table_bytes);
            D->zd_woff = upload(zw, &D->table_bytes);
            D->zd_rows = upload(rows, &D->table_bytes);
            D->n_zd_tiles = nrt;
            HIP_OK(dev_malloc(&D->ycl, (size_t)std::max(Tt.n_points, 1) * sizeof(T)));
            D->table_bytes += (size_t)Tt.n_points * sizeof(T);
        }
        D->segs = upload(segs, &D->table_bytes);
        D->tilesB_user = upload(tB, &D->table_bytes);
        D->tilesB_cluster = upload(tBc, &D->table_bytes);
        D->tilesA = upload(tA, &D->table_bytes);
        D->tilesA2 = upload(tA2, &D->table_bytes);
        D->tilesB_split = upload(tBs, &D->table_bytes);
        D->nB_split = (int)tBs.size();
        D->nB = (int)tB.size(); D->nA = (int)tA.size(); D->nA2 = (int)tA2.size();
    }
};

// ------------------------------------------------------------------------------------------------
void device_build_from_host(HMatrix &H, const void *arena, int64_t arena_elems) {
    DeviceBuilder db(H);
    size_t es = H.is_complex ? 16 : 8;
    void *d_arena = nullptr;
    HIP_OK(dev_malloc(&d_arena, std::max<int64_t>(arena_elems, 1) * es));
    if (arena_elems) HIP_OK(hipMemcpy(d_arena, arena, arena_elems * es, hipMemcpyHostToDevice));
    std::vector<int64_t> all;
    for (size_t i = 0; i < H.blocks.size(); i++) if (H.blocks[i].rank != 0) all.push_back((int64_t)i);
    if (H.is_complex) db.pack_batch<double2>(all, d_arena, false);
    else db.pack_batch<double>(all, d_arena, false);
    HIP_OK(hipFree(d_arena));
    if (H.is_complex) db.assemble<double2>();
    else db.assemble<double>();
    H.n_batches = (int)db.D->batches.size();
}

template <typename Ops, int NR>
static void launch_sweep(DeviceHMatrix *D, const void *x_dev, long long x_stride, void *y_dev, long long y_stride, int numbering, hipStream_t st) {
    typedef typename Ops::T T;
    T *W = (T *)D->W;
    const int Ns = D->n_source;
    const long long ws = D->W_elems;
    // numbering: 0 user in / user out, 1 cluster / cluster, 2 user / cluster, 3 cluster / user
    const bool in_user = numbering == 0 || numbering == 2, out_user = numbering == 0 || numbering == 3;
    hipEvent_t *ev = D->pev[D->nprod % DeviceHMatrix::RING];
    const bool timing = D->phase_timing; // five event records cost ~19 us per product: off unless asked for
    if (timing) HIP_OK(hipEventRecord(ev[0], st));
    if (Ns) hipLaunchKernelGGL(gather_x_kernel<T>, dim3((Ns + 255) / 256), dim3(256), 0, st, (const T *)x_dev, x_stride, in_user ? D->perm_s : (const int *)nullptr, W, ws, Ns, NR);
    if (timing) HIP_OK(hipEventRecord(ev[1], st));
    if (D->nA) hipLaunchKernelGGL((tile_gemv_tall<Ops, 16, NR>), dim3(D->nA), dim3(256), 0, st, D->tilesA, D->segs, (const T *)W, W, ws, ws, 0LL);
    if (timing) HIP_OK(hipEventRecord(ev[2], st));
    if (D->nA2) hipLaunchKernelGGL((tile_gemv_tall<Ops, 16, NR>), dim3(D->nA2), dim3(256), 0, st, D->tilesA2, D->segs, (const T *)W, W, ws, ws, ws);
    if (timing) HIP_OK(hipEventRecord(ev[3], st));
    auto launch_wide = [&](const GTile *tiles, const int *cnt, T *out, long long out_stride) {
        const GTile *t = tiles;
        if (cnt[0]) hipLaunchKernelGGL((tile_gemv_wide<Ops, 16, NR, 1>), dim3(cnt[0]), dim3(256), 0, st, t, D->segs, (const T *)W, out, ws, out_stride);
        t += cnt[0];
        if (cnt[1]) hipLaunchKernelGGL((tile_gemv_wide<Ops, 16, NR, 2>), dim3(cnt[1]), dim3(256), 0, st, t, D->segs, (const T *)W, out, ws, out_stride);
        t += cnt[1];
        if (cnt[2]) hipLaunchKernelGGL((tile_gemv_wide<Ops, 16, NR, 4>), dim3(cnt[2]), dim3(256), 0, st, t, D->segs, (const T *)W, out, ws, out_stride);
    };
    if (D->one_triangle) {
        // fused sweep: y (cluster numbering) and the transposed dot products of every column, then the transposed
        // use of the V panels, then the transposed dense results and the scatter to the caller's numbering
        if constexpr (NR == 1) {
            T *ycl = (T *)D->ycl;
            if (D->splitB > 1 && D->nB_split) { // small operator: column slices of the row tiles, summed in slice order
                hipLaunchKernelGGL((tile_gemv_wide_sym<Ops>), dim3(D->nB_split), dim3(256), 0, st, D->tilesB_split, D->segs, (const T *)W, W, (T *)D->ypart, D->conj_transposed ? 1 : 0);
                hipLaunchKernelGGL(reduce_y_kernel<T>, dim3((D->row_size + 255) / 256), dim3(256), 0, st, (const T *)D->ypart, D->ypart_stride, D->splitB, D->row_size,
                                   (const int *)nullptr, ycl, 0LL, 1);
            } else if (D->nB) {
                hipLaunchKernelGGL((tile_gemv_wide_sym<Ops>), dim3(D->nB), dim3(256), 0, st, D->tilesB_cluster, D->segs, (const T *)W, W, ycl, D->conj_transposed ? 1 : 0);
            }
            if (D->nZ) hipLaunchKernelGGL((tile_gemv_tall<Ops, 16, 1>), dim3(D->nZ), dim3(256), 0, st, D->tilesZ, D->segs, (const T *)W, W, ws, ws, 0LL);
            if (D->nAT) hipLaunchKernelGGL((tile_gemv_tall_transposed<Ops>), dim3(D->nAT), dim3(256), 0, st, D->tilesAT, D->segs, (const T *)W, ycl, D->conj_transposed ? 1 : 0);
            if (D->n_zd_tiles) hipLaunchKernelGGL(finish_sym_kernel<T>, dim3(D->n_zd_tiles), dim3(128), 0, st, (const T *)ycl, (const T *)W, D->zd_ptr, D->zd_woff, D->zd_rows,
                                                  out_user ? D->perm_t : (const int *)nullptr, (T *)y_dev);
        } else {
            throw Error("one-triangle storage: products are swept one right-hand side at a time");
        }
    } else if (D->splitB > 1 && D->nB_split) {
        const long long ps = (long long)D->splitB * D->ypart_stride;
        launch_wide(D->tilesB_split, D->cntBs, (T *)D->ypart, ps);
        hipLaunchKernelGGL(reduce_y_kernel<T>, dim3((D->row_size + 255) / 256), dim3(256), 0, st, (const T *)D->ypart, D->ypart_stride, D->splitB, D->row_size,
                           out_user ? D->perm_t + D->row_off : (const int *)nullptr, (T *)y_dev, y_stride, NR);
    } else if (D->nB) {
        launch_wide(out_user ? D->tilesB_user : D->tilesB_cluster, D->cntB, (T *)y_dev, y_stride);
    }
    if (timing) HIP_OK(hipEventRecord(ev[4], st));
    HIP_OK(hipGetLastError());
    if (timing) D->nprod++;
}

// make room for nr coefficient workspaces (and partial-y slabs); W[r][n_source] = 1 for every r
template <typename T>
static void ensure_rhs_capacity(DeviceHMatrix *D, int nr) {
    if (nr <= D->rhs_cap) return;
    void *nW = nullptr;
    HIP_OK(dev_malloc(&nW, (size_t)nr * D->W_elems * sizeof(T)));
    HIP_OK(hipMemset(nW, 0, (size_t)nr * D->W_elems * sizeof(T)));
    T one;
    std::memset(&one, 0, sizeof(T));
    *(double *)&one = 1.0;
    for (int r = 0; r < nr; r++) HIP_OK(hipMemcpy((char *)nW + ((size_t)r * D->W_elems + D->n_source) * sizeof(T), &one, sizeof(T), hipMemcpyHostToDevice));
    // A2 segments point into W: relocate them
    if (D->W && D->nA2) {
        std::vector<GTile> t((size_t)D->nA2);
        HIP_OK(hipMemcpy(t.data(), D->tilesA2, t.size() * sizeof(GTile), hipMemcpyDeviceToHost));
        for (auto &x : t) {
            GSeg sg;
            HIP_OK(hipMemcpy(&sg, D->segs + x.seg_begin, sizeof(GSeg), hipMemcpyDeviceToHost));
            sg.panel = (const char *)nW + ((const char *)sg.panel - (const char *)D->W);
            HIP_OK(hipMemcpy(D->segs + x.seg_begin, &sg, sizeof(GSeg), hipMemcpyHostToDevice));
        }
    }
    if (D->W) (void)hipFree(D->W);
    D->W = nW;
    if (D->splitB > 1) {
        if (D->ypart) (void)hipFree(D->ypart);
        HIP_OK(dev_malloc(&D->ypart, (size_t)nr * D->splitB * D->ypart_stride * sizeof(T)));
    }
    D->rhs_cap = nr;
}

template <typename Ops>
static void launch_product(DeviceHMatrix *D, const void *X, long long x_stride, void *Y, long long y_stride, int mu, int numbering, hipStream_t st) {
    typedef typename Ops::T T;
    int done = 0;
    while (done < mu) {
        const int left = mu - done;
        const T *x = (const T *)X + (long long)done * x_stride;
        T *y = (T *)Y + (long long)done * y_stride;
        if (D->one_triangle) { launch_sweep<Ops, 1>(D, x, x_stride, y, y_stride, numbering, st); done += 1; continue; }
        if (left >= 8) { launch_sweep<Ops, 8>(D, x, x_stride, y, y_stride, numbering, st); done += 8; }
        else if (left >= 4) { launch_sweep<Ops, 4>(D, x, x_stride, y, y_stride, numbering, st); done += 4; }
        else if (left >= 2) { launch_sweep<Ops, 2>(D, x, x_stride, y, y_stride, numbering, st); done += 2; }
        else { launch_sweep<Ops, 1>(D, x, x_stride, y, y_stride, numbering, st); done += 1; }
    }
}

// Y = H X for mu right-hand sides stored with the given strides (elements); mu = 1 is the matvec
void device_matmat_device(const HMatrix &H, const void *X, long long x_stride, void *Y, long long y_stride, int mu, int numbering, void *stream) {
    DeviceHMatrix *D = H.dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    HIP_OK(hipSetDevice(D->device));
    hipStream_t st = stream ? (hipStream_t)stream : D->stream;
    const int need = D->one_triangle ? 1 : (mu >= 8 ? 8 : mu >= 4 ? 4 : mu >= 2 ? 2 : 1);
    if (need > D->rhs_cap) {
        HIP_OK(hipStreamSynchronize(st));
        if (D->is_complex) ensure_rhs_capacity<double2>(D, need);
        else ensure_rhs_capacity<double>(D, need);
    }
    if (D->is_complex) launch_product<CplxOps>(D, X, x_stride, Y, y_stride, mu, numbering, st);
    else launch_product<RealOps>(D, X, x_stride, Y, y_stride, mu, numbering, st);
}

void device_matvec_device(const HMatrix &H, const void *x_dev, void *y_dev, int numbering, void *stream) {
    device_matmat_device(H, x_dev, 0, y_dev, 0, 1, numbering, stream);
}

// host API convention: a matrix built on the whole target (source) cluster takes/returns user numbering on
// that side; one built on a partition works on its local slice in cluster order
static int host_numbering(const HMatrix &H) {
    const bool in_user = H.s_root == 0 && !H.local_numbering, out_user = H.t_root == 0 && !H.local_numbering;
    return in_user ? (out_user ? 0 : 2) : (out_user ? 3 : 1);
}

void device_matvec_host(const HMatrix &H, const void *x, void *y) {
    DeviceHMatrix *D = H.dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    HIP_OK(hipSetDevice(D->device));
    const size_t es = D->esize;
    HIP_OK(hipMemcpyAsync(D->x_tmp, x, (size_t)D->n_source * es, hipMemcpyHostToDevice, D->stream));
    // a matrix built on the whole target cluster answers in user numbering; one built on a partition
    // answers with its local rows in cluster order
    const bool whole = H.t_root == 0; // (then row_size == n_target)
    device_matvec_device(H, D->x_tmp, D->y_tmp, host_numbering(H), D->stream);
    HIP_OK(hipMemcpyAsync(y, D->y_tmp, (size_t)(whole ? D->n_target : D->row_size) * es, hipMemcpyDeviceToHost, D->stream));
    HIP_OK(hipStreamSynchronize(D->stream));
}

// Y = H X on host buffers, X column-major n_source x mu (user numbering), all right-hand sides in one go
void device_matmat_host(const HMatrix &H, const void *X, int mu, void *Y) {
    DeviceHMatrix *D = H.dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    HIP_OK(hipSetDevice(D->device));
    const size_t es = D->esize;
    const bool whole = H.t_root == 0;
    const size_t nin = (size_t)D->n_source, nout = (size_t)(whole ? D->n_target : D->row_size);
    void *dX = nullptr, *dY = nullptr;
    HIP_OK(dev_malloc(&dX, std::max<size_t>(nin * mu, 1) * es));
    HIP_OK(dev_malloc(&dY, std::max<size_t>(nout * mu, 1) * es));
    HIP_OK(hipMemcpyAsync(dX, X, nin * mu * es, hipMemcpyHostToDevice, D->stream));
    device_matmat_device(H, dX, (long long)nin, dY, (long long)nout, mu, host_numbering(H), D->stream);
    HIP_OK(hipMemcpyAsync(Y, dY, nout * mu * es, hipMemcpyDeviceToHost, D->stream));
    HIP_OK(hipStreamSynchronize(D->stream));
    (void)hipFree(dX);
    (void)hipFree(dY);
}

void device_set_phase_timing(const HMatrix &H, bool on) {
    HM_CHECK(H.dev != nullptr, "H-matrix has no device data");
    H.dev->phase_timing = on;
}

// average duration (microseconds) of the four launches over the completed products still in the ring:
// out[0] gather/copy of x, out[1] phase A, out[2] phase A2, out[3] phase B.  Returns the number averaged.
int device_phase_times(const HMatrix &H, double *out4) {
    DeviceHMatrix *D = H.dev;
    for (int i = 0; i < 4; i++) out4[i] = 0;
    if (!D) return 0;
    (void)hipSetDevice(D->device);
    int cnt = 0;
    const long long lo = std::max<long long>(0, D->nprod - DeviceHMatrix::RING);
    for (long long p = lo; p < D->nprod; p++) {
        hipEvent_t *ev = D->pev[p % DeviceHMatrix::RING];
        if (hipEventQuery(ev[4]) != hipSuccess) { (void)hipGetLastError(); continue; }
        float ms[4];
        bool ok = true;
        for (int i = 0; i < 4; i++) if (hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]) != hipSuccess) { ok = false; (void)hipGetLastError(); }
        if (!ok) continue;
        for (int i = 0; i < 4; i++) out4[i] += ms[i] * 1e3;
        cnt++;
    }
    if (cnt) for (int i = 0; i < 4; i++) out4[i] /= cnt;
    return cnt;
}

double device_last_product_us(const HMatrix &H) {
    DeviceHMatrix *D = H.dev;
    if (!D || D->nprod == 0) return -1;
    hipEvent_t *ev = D->pev[(D->nprod - 1) % DeviceHMatrix::RING];
    float ms = 0;
    if (hipEventQuery(ev[4]) == hipSuccess && hipEventElapsedTime(&ms, ev[0], ev[4]) == hipSuccess) return ms * 1e3;
    (void)hipGetLastError();
    return -1;
}

int64_t device_resident_bytes(const HMatrix &H) {
    if (!H.dev) return 0;
    size_t b = H.dev->table_bytes + (size_t)H.dev->W_elems * H.dev->esize;
    for (auto &B : H.dev->batches) b += B.bytes;
    return (int64_t)b;
}

void device_free(DeviceHMatrix *D) {
    if (!D) return;
    (void)hipSetDevice(D->device);
    for (auto &B : D->batches) {
        (void)hipFree(B.panelB); (void)hipFree(B.panelA); (void)hipFree(B.cidxB); (void)hipFree(B.oidxA);
        if (B.zidxB) (void)hipFree(B.zidxB);
        if (B.tidxA) (void)hipFree(B.tidxA);
    }
    for (void *p : {(void *)D->tilesAT, (void *)D->tilesZ, (void *)D->zd_ptr, (void *)D->zd_woff, (void *)D->zd_rows, D->ycl})
        if (p) (void)hipFree(p);
    for (void *p : {(void *)D->segs, (void *)D->tilesB_user, (void *)D->tilesB_cluster, (void *)D->tilesA, (void *)D->tilesA2, (void *)D->perm_s,
                    (void *)D->perm_t, (void *)D->iota, (void *)D->ones_idx, D->W, D->x_tmp, D->y_tmp, (void *)D->tcoord, (void *)D->scoord, (void *)D->tilesB_split, D->ypart})
        if (p) (void)hipFree(p);
    for (auto &slot : D->pev) for (auto &e : slot) if (e) (void)hipEventDestroy(e);
    if (D->stream) (void)hipStreamDestroy(D->stream);
    delete D;
}

// deep copy: new panels, tables re-assembled against the new buffers (hmatrix.hpp:48 __deepcopy__)
void device_clone(const HMatrix &src, HMatrix &dst) {
    const DeviceHMatrix *S = src.dev;
    HM_CHECK(S != nullptr, "H-matrix has no device data");
    HIP_OK(hipSetDevice(S->device));
    // Re-pack is not possible (the arena is gone), so copy buffers and relocate pointers in the tables.
    DeviceHMatrix *D = new DeviceHMatrix(*S);
    dst.dev = D;
    D->stream = nullptr;
    D->nprod = 0;
    HIP_OK(hipStreamCreate(&D->stream));
    for (auto &slot : D->pev) for (auto &e : slot) { e = nullptr; HIP_OK(hipEventCreate(&e)); }
    auto dup = [](const void *p, size_t bytes) -> void * {
        void *q = nullptr;
        HIP_OK(dev_malloc(&q, std::max<size_t>(bytes, 1)));
        if (p && bytes) HIP_OK(hipMemcpy(q, p, bytes, hipMemcpyDeviceToDevice));
        return q;
    };
    struct Range { const char *old_lo, *old_hi; char *neu; };
    std::vector<Range> map;
    const size_t es = S->esize;
    // sizes of per-batch buffers are recomputed from the stored byte count split: keep it simple by
    // querying the allocation sizes through hipMemPtrGetInfo
    auto dup_alloc = [&](const void *p) -> void * {
        if (!p) return nullptr;
        size_t sz = 0;
        HIP_OK(hipMemPtrGetInfo(const_cast<void *>(p), &sz));
        void *q = dup(p, sz);
        map.push_back({(const char *)p, (const char *)p + sz, (char *)q});
        return q;
    };
    for (size_t b = 0; b < S->batches.size(); b++) {
        D->batches[b].panelB = dup_alloc(S->batches[b].panelB);
        D->batches[b].panelA = dup_alloc(S->batches[b].panelA);
        D->batches[b].cidxB = (int *)dup_alloc(S->batches[b].cidxB);
        D->batches[b].oidxA = (int *)dup_alloc(S->batches[b].oidxA);
        D->batches[b].zidxB = (int *)dup_alloc(S->batches[b].zidxB);
        D->batches[b].tidxA = (int *)dup_alloc(S->batches[b].tidxA);
    }
    D->zd_ptr = (int *)dup_alloc(S->zd_ptr);
    D->zd_woff = (long long *)dup_alloc(S->zd_woff);
    D->zd_rows = (int *)dup_alloc(S->zd_rows);
    D->ycl = dup_alloc(S->ycl);
    D->W = dup_alloc(S->W);
    D->perm_s = (int *)dup_alloc(S->perm_s);
    D->perm_t = (int *)dup_alloc(S->perm_t);
    D->iota = (int *)dup_alloc(S->iota);
    D->ones_idx = (int *)dup_alloc(S->ones_idx);
    D->x_tmp = dup_alloc(S->x_tmp);
    D->y_tmp = dup_alloc(S->y_tmp);
    D->tcoord = (double *)dup_alloc(S->tcoord);
    D->scoord = (double *)dup_alloc(S->scoord);
    auto reloc = [&](const void *p) -> const void * {
        if (!p) return nullptr;
        for (auto &r : map) if ((const char *)p >= r.old_lo && (const char *)p < r.old_hi) return r.neu + ((const char *)p - r.old_lo);
        throw Error("device_clone: dangling table pointer");
    };
    // segment table
    size_t seg_bytes = 0;
    HIP_OK(hipMemPtrGetInfo(S->segs, &seg_bytes));
    std::vector<GSeg> segs(seg_bytes / sizeof(GSeg));
    HIP_OK(hipMemcpy(segs.data(), S->segs, segs.size() * sizeof(GSeg), hipMemcpyDeviceToHost));
    int nseg_used = 0;
    auto fix_tiles = [&](const GTile *d_src, int n, GTile **d_dst) {
        std::vector<GTile> t((size_t)std::max(n, 0));
        if (n) HIP_OK(hipMemcpy(t.data(), d_src, (size_t)n * sizeof(GTile), hipMemcpyDeviceToHost));
        for (auto &x : t) { x.omap = (const int *)reloc(x.omap); nseg_used = std::max<long long>(nseg_used, x.seg_begin + x.nseg); }
        *d_dst = upload(t);
    };
    fix_tiles(S->tilesB_user, S->nB, &D->tilesB_user);
    fix_tiles(S->tilesB_cluster, S->nB, &D->tilesB_cluster);
    fix_tiles(S->tilesA, S->nA, &D->tilesA);
    fix_tiles(S->tilesA2, S->nA2, &D->tilesA2);
    fix_tiles(S->tilesB_split, S->nB_split, &D->tilesB_split);
    fix_tiles(S->tilesAT, S->nAT, &D->tilesAT);
    fix_tiles(S->tilesZ, S->nZ, &D->tilesZ);
    if (S->ypart) D->ypart = dup_alloc(S->ypart);
    segs.resize((size_t)nseg_used);
    for (auto &s : segs) { s.panel = reloc(s.panel); s.cidx = (const int *)reloc(s.cidx); s.zidx = (const int *)reloc(s.zidx); s.oidx = (const int *)reloc(s.oidx); }
    D->segs = upload(segs);
    (void)es;
}

// panels of one leaf, copied back to the host (introspection / parity tests)
void device_leaf_panels(const HMatrix &H, int64_t leaf, void *A, void *Bout) {
    const DeviceHMatrix *D = H.dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    HM_CHECK(leaf >= 0 && leaf < (int64_t)H.blocks.size(), "leaf index out of range");
    HIP_OK(hipSetDevice(D->device));
    const BlockRec &b = H.blocks[leaf];
    if (b.rank == 0) return;
    const size_t es = D->esize;
    const int vec_rows = H.is_complex ? 1 : 2;
    const BatchTables &L = D->tabs[b.batch];
    std::vector<DevBlock> hb{to_dev(b, H)};
    TempPool tmp;
    DevBlock *d_b = tmp.up(hb);
    PackArgs a;
    std::memset(&a, 0, sizeof(a));
    a.blocks = d_b; a.vec_rows = vec_rows; a.tile_max = H.tile_max;
    std::vector<int> ut, vt;
    for (int r = H.rtiles.node_tile_begin[b.t_node]; r < H.rtiles.node_tile_end[b.t_node]; r++) ut.push_back(r);
    if (b.rank > 0) for (int c = H.ctiles.node_tile_begin[b.s_node]; c < H.ctiles.node_tile_end[b.s_node]; c++) vt.push_back(c);
    int *d_ut = tmp.up(ut), *d_vt = tmp.up(vt);
    int *d_rto = tmp.up(H.rtiles.off), *d_rts = tmp.up(H.rtiles.size), *d_cto = tmp.up(H.ctiles.off), *d_cts = tmp.up(H.ctiles.size);
    std::vector<long long> t1(L.b_pbase.begin(), L.b_pbase.end()), t3(L.a_pbase.begin(), L.a_pbase.end());
    long long *d_bp = tmp.up(t1), *d_ap = tmp.up(t3);
    int *d_an = tmp.up(L.a_nrows);
    const int ncolsU = b.rank >= 0 ? b.rank : b.n;
    void *d_outA = tmp.alloc((size_t)b.m * ncolsU * es), *d_outB = nullptr;
    a.item_tile = d_ut; a.tile_off = d_rto; a.tile_size = d_rts; a.tile_pbase = d_bp; a.panel = D->batches[b.batch].panelB;
    if (H.is_complex) hipLaunchKernelGGL(unpack_u_kernel<double2>, dim3((unsigned)ut.size()), dim3(256), 0, D->stream, a, (double2 *)d_outA);
    else hipLaunchKernelGGL(unpack_u_kernel<double>, dim3((unsigned)ut.size()), dim3(256), 0, D->stream, a, (double *)d_outA);
    if (b.rank > 0) {
        d_outB = tmp.alloc((size_t)b.n * b.rank * es);
        a.item_tile = d_vt; a.tile_off = d_cto; a.tile_size = d_cts; a.tile_pbase = d_ap; a.tile_n = d_an; a.panel = D->batches[b.batch].panelA;
        if (H.is_complex) hipLaunchKernelGGL(unpack_v_kernel<double2>, dim3((unsigned)vt.size()), dim3(256), 0, D->stream, a, (double2 *)d_outB);
        else hipLaunchKernelGGL(unpack_v_kernel<double>, dim3((unsigned)vt.size()), dim3(256), 0, D->stream, a, (double *)d_outB);
    }
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(D->stream));
    HIP_OK(hipMemcpy(A, d_outA, (size_t)b.m * ncolsU * es, hipMemcpyDeviceToHost));
    if (b.rank > 0) HIP_OK(hipMemcpy(Bout, d_outB, (size_t)b.n * b.rank * es, hipMemcpyDeviceToHost));
}

} // namespace hm

#include "device_build.inc"
#include "device_recompress.inc"
