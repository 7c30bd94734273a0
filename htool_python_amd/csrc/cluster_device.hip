// cluster_device.hip -- the geometric cluster tree of cluster.cpp, built on the GPU level by level.
//
// Replaces htool::ClusterTreeBuilder<double>::create_cluster_tree as called from
// src/htool/clustering/cluster_tree_builder.hpp:19-23,39,56 (algorithm: SURVEY.md Appendix A.2).
// Same definition as the host builder (cluster.cpp) down to the last bit, so that the permutation and the node table
// do not depend on where the tree was built:
//   * sums over the points of a node (weighted mean, covariance) are formed in blocks of SUM_BLOCK consecutive points
//     of the cluster order, every block a plain running sum from zero, the block sums added in block order -- here ONE
//     WAVE per block: the 64 lanes load 64 points and form their terms side by side, the terms go through a small LDS
//     slab, and lane j adds up accumulator j in point order (4 chains for a mean, 9 for a covariance);
//   * the split direction (cyclic Jacobi on the 3 x 3 covariance, or the longest bounding-box edge) is one thread per node,
//     statement for statement the host's arithmetic (this file is compiled with -ffp-contract=off like the host code);
//   * the points of a node are ordered by a STABLE sort of their projections: nodes of at most LDS_CAP points by one
//     workgroup each, entirely in LDS (bitonic network on (projection, position) pairs: distinct keys, so any network
//     gives the stable order), larger nodes by a segmented least-significant-digit radix sort over the 64 key bits
//     (8 passes of 8 bits; tiles never straddle a node; ranks inside a tile by wave ballots in position order);
//   * radii are maxima (atomicMax on the bit pattern of non-negative doubles): the same in any order.
// A level costs 7 launches (LDS sort) or about 40 (radix sort) and one 24-byte read-back; 1 M points: about 20 levels.
#include "device_internal.hpp"

#include <algorithm>
#include <cstring>
#include <mutex>

namespace hm {

int device_current();

namespace {

typedef unsigned long long u64;
constexpr int SUM_BLOCK = 4096; // as in cluster.cpp
constexpr int TILE = 2048;      // elements per radix tile (256 threads x 8)
constexpr int LDS_CAP = 4096;   // largest node sorted by one workgroup in LDS (64 KB of pairs)

struct CtInfo {
    int lvl_lo, lvl_cnt, n_nodes, n_blk, n_tile, max_size;
};

struct CtView {
    int N, d, nc, max_leaf, P;
    const double *pts, *wts, *rad;
    int *perm;
    u64 *kA, *kB;
    int *vA, *vB;
    int *n_off, *n_size, *n_depth, *n_parent, *n_first, *n_nchild, *n_part;
    double *n_cx, *n_cy, *n_cz, *n_rad;
    int *blk_first, *tile_first; // per node of the current level (count + 1 entries)
    double *dir;                 // 3 per node of the current level
    int *ok, *csize;             // split accepted; piece sizes (pieces per node)
    double *part;                // per block: 9 partial sums
    unsigned *hist;              // per radix tile: 256 digit counts, then scatter bases
    CtInfo *info;
};

// ---- small device helpers ----------------------------------------------------------------------------------------
__device__ inline int find_owner(const int *first, int cnt, int b) { // largest q in [0, cnt) with first[q] <= b
    int lo = 0, hi = cnt;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (first[mid] <= b) lo = mid;
        else hi = mid;
    }
    return lo;
}

// order-preserving map double -> u64 (-0.0 counts as +0.0: the host compares values, not bit patterns)
__device__ inline u64 enc_key(double s) {
    u64 b = (u64)__double_as_longlong(s);
    if ((b << 1) == 0) b = 0;
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double dec_key(u64 k) {
    const u64 b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__device__ inline double projection(const CtView &v, int u, const double c[3], const double dir[3]) {
    double s = 0;
    for (int k = 0; k < v.d; k++) s += (v.pts[(size_t)u * v.d + k] - c[k]) * dir[k];
    return s;
}

// exclusive scan over the 1024 threads of a workgroup; sh has 1024 entries
__device__ inline int block_excl_scan(int x, int *total, int *sh) {
    const int tid = threadIdx.x;
    sh[tid] = x;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int t = tid >= o ? sh[tid - o] : 0;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    const int incl = sh[tid];
    *total = sh[1023];
    __syncthreads();
    return incl - x;
}

// blocks of SUM_BLOCK points and radix tiles of the nodes [lo, lo + cnt): prefix arrays, totals, the largest node
__device__ void scan_level(const CtView &v, int lo, int cnt, int *sh) {
    const int tid = threadIdx.x, chunk = (cnt + 1023) / 1024;
    const int q0 = min(cnt, tid * chunk), q1 = min(cnt, q0 + chunk);
    int nb = 0, nt = 0, mx = 0;
    for (int q = q0; q < q1; q++) {
        const int sz = v.n_size[lo + q];
        nb += (sz + SUM_BLOCK - 1) / SUM_BLOCK;
        nt += (sz + TILE - 1) / TILE;
        mx = max(mx, sz);
    }
    int tb, tt;
    int pb = block_excl_scan(nb, &tb, sh);
    int pt = block_excl_scan(nt, &tt, sh);
    for (int q = q0; q < q1; q++) {
        const int sz = v.n_size[lo + q];
        v.blk_first[q] = pb;
        v.tile_first[q] = pt;
        pb += (sz + SUM_BLOCK - 1) / SUM_BLOCK;
        pt += (sz + TILE - 1) / TILE;
    }
    sh[tid] = mx;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) sh[tid] = max(sh[tid], sh[tid + o]);
        __syncthreads();
    }
    if (tid == 0) {
        v.blk_first[cnt] = tb;
        v.tile_first[cnt] = tt;
        v.info->n_blk = tb;
        v.info->n_tile = tt;
        v.info->max_size = sh[0];
    }
    __syncthreads();
}

// ---- node table ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ct_init(CtView v, int iota) {
    __shared__ int sh[1024];
    if (iota)
        for (int i = threadIdx.x; i < v.N; i += 1024) v.perm[i] = i; // (one workgroup: 1 M entries take a few microseconds of 4-byte stores)
    if (threadIdx.x == 0) {
        v.n_off[0] = 0; v.n_size[0] = v.N; v.n_depth[0] = 0; v.n_parent[0] = -1; v.n_first[0] = -1; v.n_nchild[0] = 0;
        v.n_part[0] = v.P == 1 ? 0 : -1;
        v.info->lvl_lo = 0; v.info->lvl_cnt = 1; v.info->n_nodes = 1;
    }
    __syncthreads();
    scan_level(v, 0, 1, sh);
}

// the accepted nodes of the current level get their children (ids in level order, children consecutive); the children
// become the current level
__global__ __launch_bounds__(1024) void ct_assign_children(CtView v, int pieces, int partition_level) {
    __shared__ int sh[1024];
    const int lo = v.info->lvl_lo, cnt = v.info->lvl_cnt, nn = v.info->n_nodes;
    __syncthreads();
    const int tid = threadIdx.x, chunk = (cnt + 1023) / 1024;
    const int q0 = min(cnt, tid * chunk), q1 = min(cnt, q0 + chunk);
    int mine = 0;
    for (int q = q0; q < q1; q++) mine += v.ok[q] != 0;
    int total;
    int base = block_excl_scan(mine, &total, sh);
    for (int q = q0; q < q1; q++) {
        if (!v.ok[q]) continue;
        const int id = lo + q, first = nn + base * pieces;
        base++;
        v.n_first[id] = first;
        v.n_nchild[id] = pieces;
        int o = v.n_off[id];
        for (int p = 0; p < pieces; p++) {
            const int ch = first + p, sz = v.csize[(size_t)q * pieces + p];
            v.n_off[ch] = o; v.n_size[ch] = sz; v.n_depth[ch] = v.n_depth[id] + 1; v.n_parent[ch] = id;
            v.n_first[ch] = -1; v.n_nchild[ch] = 0; v.n_part[ch] = partition_level ? p : v.n_part[id];
            o += sz;
        }
    }
    __threadfence_block();
    __syncthreads();
    const int ncnt = total * pieces;
    if (tid == 0) { v.info->lvl_lo = nn; v.info->lvl_cnt = ncnt; v.info->n_nodes = nn + ncnt; }
    scan_level(v, nn, ncnt, sh);
}

// ---- blocked sums --------------------------------------------------------------------------------------------------
// one wave per block of SUM_BLOCK points.  COV = false: {sum w, sum w x_k}; true: the 9 entries sum w t_p t_q, t = x - centre
template <int NA, bool COV>
__global__ __launch_bounds__(64) void ct_block_sums(CtView v, int lo, int cnt) {
    __shared__ double sh[NA][65];
    const int B = blockIdx.x, lane = threadIdx.x;
    if (B >= v.blk_first[cnt]) return;
    const int q = find_owner(v.blk_first, cnt, B), id = lo + q;
    const int off = v.n_off[id], i0 = off + (B - v.blk_first[q]) * SUM_BLOCK, i1 = min(off + v.n_size[id], i0 + SUM_BLOCK);
    const double c[3] = {v.n_cx[id], v.n_cy[id], v.n_cz[id]};
    const int d = v.d;
    double acc = 0;
    for (int base = i0; base < i1; base += 64) {
        const int n = min(64, i1 - base);
        if (lane < n) {
            const int u = v.perm[base + lane];
            const double w = v.wts ? v.wts[u] : 1.0;
            double x[3] = {0, 0, 0};
            for (int k = 0; k < d; k++) x[k] = v.pts[(size_t)u * d + k];
            if (!COV) {
                sh[0][lane] = w;
                for (int k = 0; k < 3; k++) sh[(1 + k) % NA][lane] = k < d ? w * x[k] : 0.0;
            } else {
                double t[3] = {0, 0, 0};
                for (int k = 0; k < d; k++) t[k] = x[k] - c[k];
                for (int p = 0; p < 3; p++)
                    for (int r = 0; r < 3; r++) sh[(p * 3 + r) % NA][lane] = (p < d && r < d) ? w * t[p] * t[r] : 0.0;
            }
        }
        __syncthreads();
        if (lane < NA) {
            if (n == 64) {
#pragma unroll
                for (int i = 0; i < 64; i += 16) {
                    double t[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) t[j] = sh[lane][i + j];
#pragma unroll
                    for (int j = 0; j < 16; j++) acc += t[j];
                }
            } else {
                for (int i = 0; i < n; i++) acc += sh[lane][i];
            }
        }
        __syncthreads();
    }
    if (lane < NA) v.part[(size_t)B * 9 + lane] = acc;
}

// bounding box of a block (minima and maxima: the same in any order): part[B][0..2] = lo, [3..5] = hi
__global__ __launch_bounds__(64) void ct_block_bbox(CtView v, int lo, int cnt) {
    const int B = blockIdx.x, lane = threadIdx.x;
    if (B >= v.blk_first[cnt]) return;
    const int q = find_owner(v.blk_first, cnt, B), id = lo + q;
    const int off = v.n_off[id], i0 = off + (B - v.blk_first[q]) * SUM_BLOCK, i1 = min(off + v.n_size[id], i0 + SUM_BLOCK);
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int i = i0 + lane; i < i1; i += 64) {
        const int u = v.perm[i];
        for (int k = 0; k < v.d; k++) {
            const double x = v.pts[(size_t)u * v.d + k];
            mn[k] = fmin(mn[k], x);
            mx[k] = fmax(mx[k], x);
        }
    }
    for (int o = 32; o > 0; o >>= 1)
        for (int k = 0; k < 3; k++) {
            mn[k] = fmin(mn[k], __shfl_xor(mn[k], o));
            mx[k] = fmax(mx[k], __shfl_xor(mx[k], o));
        }
    if (lane == 0)
        for (int k = 0; k < 3; k++) { v.part[(size_t)B * 9 + k] = mn[k]; v.part[(size_t)B * 9 + 3 + k] = mx[k]; }
}

__global__ void ct_node_centre(CtView v, int lo, int cnt) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= cnt) return;
    const int id = lo + q, d = v.d;
    double c[3] = {0, 0, 0}, wsum = 0;
    for (int b = v.blk_first[q]; b < v.blk_first[q + 1]; b++) {
        for (int k = 0; k < d; k++) c[k] += v.part[(size_t)b * 9 + 1 + k];
        wsum += v.part[(size_t)b * 9];
    }
    if (wsum != 0)
        for (int k = 0; k < d; k++) c[k] /= wsum;
    v.n_cx[id] = c[0]; v.n_cy[id] = c[1]; v.n_cz[id] = c[2];
    v.n_rad[id] = 0.0;
}

__global__ __launch_bounds__(64) void ct_block_radius(CtView v, int lo, int cnt) {
    const int B = blockIdx.x, lane = threadIdx.x;
    if (B >= v.blk_first[cnt]) return;
    const int q = find_owner(v.blk_first, cnt, B), id = lo + q;
    const int off = v.n_off[id], i0 = off + (B - v.blk_first[q]) * SUM_BLOCK, i1 = min(off + v.n_size[id], i0 + SUM_BLOCK);
    const double c[3] = {v.n_cx[id], v.n_cy[id], v.n_cz[id]};
    double radius = 0;
    for (int i = i0 + lane; i < i1; i += 64) {
        const int u = v.perm[i];
        double s = 0;
        for (int k = 0; k < v.d; k++) {
            const double t = v.pts[(size_t)u * v.d + k] - c[k];
            s += t * t;
        }
        const double r = __dsqrt_rn(s) + (v.rad ? v.rad[u] : 0.0);
        if (r > radius) radius = r;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(radius, o);
        if (other > radius) radius = other;
    }
    if (lane == 0 && radius > 0) atomicMax((u64 *)&v.n_rad[id], (u64)__double_as_longlong(radius));
}

// dominant eigenvector of a symmetric d x d matrix (d <= 3), cyclic Jacobi: the statements of cluster.cpp:dominant_axis
__device__ void dominant_axis_dev(double a[3][3], int d, double dir[3]) {
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 50; sweep++) {
        double off = 0;
        for (int i = 0; i < d; i++)
            for (int j = i + 1; j < d; j++) off += a[i][j] * a[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < d; p++)
            for (int q = p + 1; q < d; q++) {
                if (fabs(a[p][q]) < 1e-300) continue;
                const double theta = (a[q][q] - a[p][p]) / (2 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + __dsqrt_rn(theta * theta + 1));
                const double cs = 1 / __dsqrt_rn(t * t + 1), sn = t * cs;
                for (int k = 0; k < d; k++) {
                    const double x = a[k][p], y = a[k][q];
                    a[k][p] = cs * x - sn * y;
                    a[k][q] = sn * x + cs * y;
                }
                for (int k = 0; k < d; k++) {
                    const double x = a[p][k], y = a[q][k];
                    a[p][k] = cs * x - sn * y;
                    a[q][k] = sn * x + cs * y;
                }
                for (int k = 0; k < d; k++) {
                    const double x = v[k][p], y = v[k][q];
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
        if (fabs(dir[k]) > 1e-14) {
            if (dir[k] < 0)
                for (int q = 0; q < d; q++) dir[q] = -dir[q];
            break;
        }
}

__global__ void ct_node_axis(CtView v, int lo, int cnt, int pca) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= cnt) return;
    const int d = v.d;
    double dir[3] = {1, 0, 0};
    if (pca) {
        double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int b = v.blk_first[q]; b < v.blk_first[q + 1]; b++)
            for (int p = 0; p < d; p++)
                for (int r = 0; r < d; r++) cov[p][r] += v.part[(size_t)b * 9 + p * 3 + r];
        dominant_axis_dev(cov, d, dir);
    } else {
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
        for (int b = v.blk_first[q]; b < v.blk_first[q + 1]; b++)
            for (int k = 0; k < d; k++) {
                mn[k] = fmin(mn[k], v.part[(size_t)b * 9 + k]);
                mx[k] = fmax(mx[k], v.part[(size_t)b * 9 + 3 + k]);
            }
        int best = 0;
        for (int k = 1; k < d; k++)
            if (mx[k] - mn[k] > mx[best] - mn[best]) best = k;
        for (int k = 0; k < 3; k++) dir[k] = k == best;
    }
    for (int k = 0; k < 3; k++) v.dir[(size_t)q * 3 + k] = dir[k];
}

// ---- piece sizes of a sorted node ----------------------------------------------------------------------------------
// keys: the node's sorted (encoded) projections.  Returns whether every piece has at least max_leaf points.
template <typename KeyAt>
__device__ bool piece_sizes(int sz, int pieces, bool regular, int max_leaf, KeyAt key_at, int *sizes) {
    bool all_large = true;
    if (regular) {
        const int base = sz / pieces;
        for (int p = 0; p < pieces; p++) sizes[p] = p == pieces - 1 ? sz - base * (pieces - 1) : base;
        for (int p = 0; p < pieces; p++) all_large = all_large && sizes[p] >= max_leaf;
        return all_large;
    }
    const double lo = dec_key(key_at(0)), hi = dec_key(key_at(sz - 1)), w = (hi - lo) / pieces;
    int pos = 0;
    for (int p = 0; p < pieces; p++) {
        const double cut = lo + w * (p + 1);
        const int start = pos;
        if (p == pieces - 1) pos = sz;
        else { // first position >= pos whose projection is not below the cut (the keys are sorted)
            int a = pos, b = sz;
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (dec_key(key_at(mid)) < cut) a = mid + 1;
                else b = mid;
            }
            pos = a;
        }
        sizes[p] = pos - start;
        all_large = all_large && sizes[p] >= max_leaf;
    }
    return all_large;
}

// ---- small nodes: one workgroup sorts a node in LDS ------------------------------------------------------------------
__global__ void ct_lds_sort(CtView v, int lo, int cnt, int pieces, int force, int regular) {
    extern __shared__ u64 sm[];
    __shared__ int accept;
    const int q = blockIdx.x, id = lo + q, tid = threadIdx.x, nt = blockDim.x;
    const int sz = v.n_size[id], off = v.n_off[id];
    const bool attempt = sz >= 1 && (force || sz / pieces >= v.max_leaf);
    if (!attempt) {
        if (tid == 0) v.ok[q] = 0;
        return;
    }
    int n2 = 2;
    while (n2 < sz) n2 <<= 1;
    u64 *keys = sm, *vals = sm + n2;
    const double c[3] = {v.n_cx[id], v.n_cy[id], v.n_cz[id]};
    const double dir[3] = {v.dir[(size_t)q * 3], v.dir[(size_t)q * 3 + 1], v.dir[(size_t)q * 3 + 2]};
    for (int i = tid; i < n2; i += nt) {
        if (i < sz) {
            const int u = v.perm[off + i];
            keys[i] = enc_key(projection(v, u, c, dir));
            vals[i] = ((u64)i << 32) | (unsigned)u;
        } else {
            keys[i] = ~0ull;
            vals[i] = ~0ull;
        }
    }
    __syncthreads();
    for (int k2 = 2; k2 <= n2; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += nt) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const u64 ka = keys[i], kb = keys[ixj], va = vals[i], vb = vals[ixj];
                    const bool a_after_b = ka > kb || (ka == kb && va > vb);
                    const bool up = (i & k2) == 0;
                    if (a_after_b == up) {
                        keys[i] = kb; keys[ixj] = ka;
                        vals[i] = vb; vals[ixj] = va;
                    }
                }
            }
            __syncthreads();
        }
    if (tid == 0) {
        const bool large = piece_sizes(sz, pieces, regular != 0, v.max_leaf, [&](int i) { return keys[i]; }, v.csize + (size_t)q * pieces);
        accept = force || large;
        v.ok[q] = accept;
    }
    __syncthreads();
    if (accept)
        for (int i = tid; i < sz; i += nt) v.perm[off + i] = (int)(vals[i] & 0xffffffffu);
}

// ---- large nodes: segmented radix sort over tiles --------------------------------------------------------------------
struct TileRange { int q, start, n; };
__device__ inline TileRange tile_range(const CtView &v, int lo, int cnt, int t) {
    TileRange r;
    r.q = find_owner(v.tile_first, cnt, t);
    const int id = lo + r.q, off = v.n_off[id];
    r.start = off + (t - v.tile_first[r.q]) * TILE;
    r.n = min(TILE, off + v.n_size[id] - r.start);
    return r;
}

__global__ __launch_bounds__(256) void ct_project(CtView v, int lo, int cnt) {
    const int t = blockIdx.x;
    if (t >= v.tile_first[cnt]) return;
    const TileRange r = tile_range(v, lo, cnt, t);
    const int id = lo + r.q;
    const double c[3] = {v.n_cx[id], v.n_cy[id], v.n_cz[id]};
    const double dir[3] = {v.dir[(size_t)r.q * 3], v.dir[(size_t)r.q * 3 + 1], v.dir[(size_t)r.q * 3 + 2]};
    for (int i = threadIdx.x; i < r.n; i += 256) {
        const int u = v.perm[r.start + i];
        v.kA[r.start + i] = enc_key(projection(v, u, c, dir));
        v.vA[r.start + i] = u;
    }
}

__global__ __launch_bounds__(256) void ct_hist(CtView v, int lo, int cnt, const u64 *kin, int shift) {
    __shared__ unsigned h[256];
    const int t = blockIdx.x, tid = threadIdx.x;
    if (t >= v.tile_first[cnt]) return;
    const TileRange r = tile_range(v, lo, cnt, t);
    h[tid] = 0;
    __syncthreads();
    for (int i = tid; i < r.n; i += 256) atomicAdd(&h[(unsigned)(kin[r.start + i] >> shift) & 255u], 1u);
    __syncthreads();
    v.hist[(size_t)t * 256 + tid] = h[tid];
}

// one workgroup per node: digit counts of its tiles -> where each tile's elements of each digit go (absolute positions)
__global__ __launch_bounds__(256) void ct_tile_scan(CtView v, int lo, int cnt) {
    __shared__ unsigned s[256];
    const int q = blockIdx.x, d = threadIdx.x;
    const int t0 = v.tile_first[q], t1 = v.tile_first[q + 1];
    if (t0 == t1) return;
    unsigned tot = 0;
    for (int t = t0; t < t1; t++) {
        const unsigned c = v.hist[(size_t)t * 256 + d];
        v.hist[(size_t)t * 256 + d] = tot;
        tot += c;
    }
    s[d] = tot;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const unsigned x = d >= o ? s[d - o] : 0;
        __syncthreads();
        s[d] += x;
        __syncthreads();
    }
    const unsigned base = s[d] - tot + (unsigned)v.n_off[lo + q];
    for (int t = t0; t < t1; t++) v.hist[(size_t)t * 256 + d] += base;
}

// stable scatter of one tile: wave w owns the elements [512 w, 512 (w + 1)) of the tile, eight rows of 64 in position order
__global__ __launch_bounds__(256) void ct_scatter(CtView v, int lo, int cnt, const u64 *kin, const int *vin, u64 *kout, int *vout, int shift) {
    __shared__ unsigned cntw[4][256];
    const int t = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    if (t >= v.tile_first[cnt]) return;
    const TileRange r = tile_range(v, lo, cnt, t);
    for (int j = 0; j < 4; j++) cntw[j][tid] = 0;
    __syncthreads();
    u64 key[8];
    int val[8];
    unsigned rank[8];
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int row = 0; row < 8; row++) {
        const int i = w * 512 + row * 64 + lane;
        const bool valid = i < r.n;
        key[row] = valid ? kin[r.start + i] : 0;
        val[row] = valid ? vin[r.start + i] : 0;
        const unsigned d = (unsigned)(key[row] >> shift) & 255u;
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const u64 bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const unsigned before = (unsigned)__popcll(peers & lt);
        unsigned prior = 0;
        if (valid) prior = cntw[w][d];
        // (the reads of a wave precede its writes: LDS operations of one wave execute in program order)
        if (valid && before == 0) cntw[w][d] = prior + (unsigned)__popcll(peers);
        rank[row] = prior + before;
    }
    __syncthreads();
    {
        unsigned run = v.hist[(size_t)t * 256 + tid];
        for (int j = 0; j < 4; j++) {
            const unsigned c = cntw[j][tid];
            cntw[j][tid] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int row = 0; row < 8; row++) {
        const int i = w * 512 + row * 64 + lane;
        if (i < r.n) {
            const unsigned d = (unsigned)(key[row] >> shift) & 255u;
            const unsigned dst = cntw[w][d] + rank[row];
            kout[dst] = key[row];
            vout[dst] = val[row];
        }
    }
}

__global__ void ct_split_sizes(CtView v, int lo, int cnt, int pieces, int force, int regular) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= cnt) return;
    const int id = lo + q, sz = v.n_size[id], off = v.n_off[id];
    const bool attempt = sz >= 1 && (force || sz / pieces >= v.max_leaf);
    if (!attempt) { v.ok[q] = 0; return; }
    const u64 *keys = v.kA + off;
    const bool large = piece_sizes(sz, pieces, regular != 0, v.max_leaf, [&](int i) { return keys[i]; }, v.csize + (size_t)q * pieces);
    v.ok[q] = force || large;
}

__global__ __launch_bounds__(256) void ct_commit(CtView v, int lo, int cnt) {
    const int t = blockIdx.x;
    if (t >= v.tile_first[cnt]) return;
    const TileRange r = tile_range(v, lo, cnt, t);
    if (!v.ok[r.q]) return;
    for (int i = threadIdx.x; i < r.n; i += 256) v.perm[r.start + i] = v.vA[r.start + i];
}

// ---- host driver -----------------------------------------------------------------------------------------------------
struct GrowBuf {
    void *p = nullptr;
    size_t cap = 0;
    void *get(size_t bytes) {
        if (bytes > cap) {
            if (p) (void)hipFree(p);
            p = nullptr; cap = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            hipError_t e = hipMalloc(&p, want);
            if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; throw Error(strprintf("cluster tree on the device: out of device memory (%.2f GB requested)", want / 1e9)); }
            cap = want;
        }
        return p;
    }
    size_t release() { const size_t b = cap; if (p) (void)hipFree(p); p = nullptr; cap = 0; return b; }
};

struct CtWorkspace {
    std::mutex mu;
    int device = -1;
    hipStream_t stream = nullptr;
    GrowBuf main, part, hist;
    CtInfo *info_host = nullptr; // pinned
} g_ct;

inline size_t al(size_t b) { return (b + 255) / 256 * 256; }

} // namespace

size_t cluster_device_release_workspace() {
    std::lock_guard<std::mutex> lock(g_ct.mu);
    return g_ct.main.release() + g_ct.part.release() + g_ct.hist.release();
}

ClusterTree *build_cluster_tree_device(const ClusterBuildArgs &a) {
    HM_CHECK(a.dim >= 1 && a.dim <= 3, "cluster tree: spatial dimension must be 1, 2 or 3");
    HM_CHECK(a.n_points > 0, "cluster tree: no points");
    HM_CHECK(a.n_children >= 2, "cluster tree: number_of_children must be >= 2");
    HM_CHECK(a.max_leaf >= 1, "cluster tree on the device: maximal_leaf_size must be >= 1");
    std::lock_guard<std::mutex> lock(g_ct.mu);
    const int dev = device_current();
    HIP_OK(hipSetDevice(dev));
    if (g_ct.device != dev) {
        g_ct.main.release(); g_ct.part.release(); g_ct.hist.release();
        if (g_ct.stream) (void)hipStreamDestroy(g_ct.stream);
        g_ct.stream = nullptr;
        g_ct.device = dev;
    }
    if (!g_ct.stream) HIP_OK(hipStreamCreateWithFlags(&g_ct.stream, hipStreamNonBlocking));
    if (!g_ct.info_host) {
        HIP_OK(hipHostMalloc((void **)&g_ct.info_host, sizeof(CtInfo), hipHostMallocDefault));
        // (a node of LDS_CAP points needs 64 KB of pairs plus the kernel's few static bytes: above the default limit)
        HIP_OK(hipFuncSetAttribute((const void *)ct_lds_sort, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_CAP * 16 + 1024));
    }
    hipStream_t st = g_ct.stream;

    const int N = a.n_points, d = a.dim, nc = a.n_children, P = a.size_of_partition < 1 ? 1 : a.size_of_partition;
    const int pieces_max = std::max(nc, P);
    const size_t max_nodes = 2 * ((size_t)N / a.max_leaf + P) + 4;
    HM_CHECK(max_nodes < ((size_t)1 << 30), "cluster tree on the device: too many nodes");

    std::unique_ptr<ClusterTree> Tp(new ClusterTree);
    ClusterTree &T = *Tp;
    T.n_points = N; T.dim = d; T.max_leaf = a.max_leaf; T.n_children = nc; T.n_partition = P;

    // one slab for everything whose size is known up front
    size_t need = 0;
    auto take = [&](size_t bytes) { const size_t o = need; need += al(bytes); return o; };
    const size_t o_pts = take((size_t)N * d * 8), o_wts = take(a.weights ? (size_t)N * 8 : 0), o_rad = take(a.radii ? (size_t)N * 8 : 0);
    const size_t o_perm = take((size_t)N * 4), o_kA = take((size_t)N * 8), o_kB = take((size_t)N * 8), o_vA = take((size_t)N * 4), o_vB = take((size_t)N * 4);
    size_t o_int[7], o_dbl[4];
    for (auto &o : o_int) o = take(max_nodes * 4);
    for (auto &o : o_dbl) o = take(max_nodes * 8);
    const size_t o_bf = take((max_nodes + 1) * 4), o_tf = take((max_nodes + 1) * 4), o_dir = take(max_nodes * 24), o_ok = take(max_nodes * 4);
    const size_t o_cs = take(std::max(max_nodes * (size_t)nc, (size_t)pieces_max) * 4), o_info = take(sizeof(CtInfo));
    char *base = (char *)g_ct.main.get(need);

    CtView v;
    v.N = N; v.d = d; v.nc = nc; v.max_leaf = a.max_leaf; v.P = P;
    v.pts = (const double *)(base + o_pts);
    v.wts = a.weights ? (const double *)(base + o_wts) : nullptr;
    v.rad = a.radii ? (const double *)(base + o_rad) : nullptr;
    v.perm = (int *)(base + o_perm);
    v.kA = (u64 *)(base + o_kA); v.kB = (u64 *)(base + o_kB); v.vA = (int *)(base + o_vA); v.vB = (int *)(base + o_vB);
    int **ip[7] = {&v.n_off, &v.n_size, &v.n_depth, &v.n_parent, &v.n_first, &v.n_nchild, &v.n_part};
    for (int i = 0; i < 7; i++) *ip[i] = (int *)(base + o_int[i]);
    double **dp[4] = {&v.n_cx, &v.n_cy, &v.n_cz, &v.n_rad};
    for (int i = 0; i < 4; i++) *dp[i] = (double *)(base + o_dbl[i]);
    v.blk_first = (int *)(base + o_bf); v.tile_first = (int *)(base + o_tf); v.dir = (double *)(base + o_dir);
    v.ok = (int *)(base + o_ok); v.csize = (int *)(base + o_cs); v.info = (CtInfo *)(base + o_info);
    v.part = nullptr; v.hist = nullptr;

    HIP_OK(hipMemcpyAsync((void *)v.pts, a.coords, (size_t)N * d * 8, hipMemcpyHostToDevice, st));
    if (a.weights) HIP_OK(hipMemcpyAsync((void *)v.wts, a.weights, (size_t)N * 8, hipMemcpyHostToDevice, st));
    if (a.radii) HIP_OK(hipMemcpyAsync((void *)v.rad, a.radii, (size_t)N * 8, hipMemcpyHostToDevice, st));

    // a partition given by the caller: sizes (and, for a global one, the order of the points) come from the host
    std::vector<int> given_sizes, given_perm;
    if (P > 1 && a.partition) {
        if (a.partition_is_local) {
            int total = 0;
            for (int p = 0; p < P; p++) {
                HM_CHECK(a.partition[2 * p] == total, "Wrong format for partition");
                given_sizes.push_back(a.partition[2 * p + 1]);
                total += a.partition[2 * p + 1];
            }
            HM_CHECK(total == N, "Wrong format for partition");
        } else {
            given_sizes.assign(P, 0);
            for (int i = 0; i < N; i++) {
                HM_CHECK(a.partition[i] >= 0 && a.partition[i] < P, "Wrong format for partition");
                given_sizes[a.partition[i]]++;
            }
            std::vector<int> start(P, 0);
            given_perm.resize(N);
            for (int p = 1; p < P; p++) start[p] = start[p - 1] + given_sizes[p - 1];
            for (int i = 0; i < N; i++) given_perm[start[a.partition[i]]++] = i;
        }
    }

    CtInfo info;
    auto read_info = [&]() {
        HIP_OK(hipMemcpyAsync(g_ct.info_host, v.info, sizeof(CtInfo), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        info = *g_ct.info_host;
        v.part = (double *)g_ct.part.get(((size_t)info.n_blk + 1) * 9 * 8);
    };
    auto grid1 = [](int n, int per) { return dim3((unsigned)std::max(1, (n + per - 1) / per)); };
    // centre and radius of the nodes of the current level
    auto geometry = [&]() {
        if (info.lvl_cnt == 0) return;
        if (info.n_blk > 0) hipLaunchKernelGGL((ct_block_sums<4, false>), dim3(info.n_blk), dim3(64), 0, st, v, info.lvl_lo, info.lvl_cnt);
        hipLaunchKernelGGL(ct_node_centre, grid1(info.lvl_cnt, 256), dim3(256), 0, st, v, info.lvl_lo, info.lvl_cnt);
        if (info.n_blk > 0) hipLaunchKernelGGL(ct_block_radius, dim3(info.n_blk), dim3(64), 0, st, v, info.lvl_lo, info.lvl_cnt);
    };
    const bool pca = a.strategy == 0 || a.strategy == 1, regular = a.strategy == 0 || a.strategy == 2;
    // splits the nodes of the current level (those that may be split) and makes their children the current level
    auto split_level = [&](int pieces, bool force, bool partition_level) {
        const int lo = info.lvl_lo, cnt = info.lvl_cnt;
        if (info.n_blk > 0) {
            if (pca) hipLaunchKernelGGL((ct_block_sums<9, true>), dim3(info.n_blk), dim3(64), 0, st, v, lo, cnt);
            else hipLaunchKernelGGL(ct_block_bbox, dim3(info.n_blk), dim3(64), 0, st, v, lo, cnt);
        }
        hipLaunchKernelGGL(ct_node_axis, grid1(cnt, 64), dim3(64), 0, st, v, lo, cnt, pca ? 1 : 0);
        if (info.max_size <= LDS_CAP) {
            int n2 = 2;
            while (n2 < info.max_size) n2 <<= 1;
            const int threads = n2 <= 128 ? 64 : (n2 <= 1024 ? 256 : 512);
            hipLaunchKernelGGL(ct_lds_sort, dim3(cnt), dim3(threads), (size_t)n2 * 16, st, v, lo, cnt, pieces, force ? 1 : 0, regular ? 1 : 0);
        } else {
            v.hist = (unsigned *)g_ct.hist.get(((size_t)info.n_tile + 1) * 256 * 4);
            hipLaunchKernelGGL(ct_project, dim3(info.n_tile), dim3(256), 0, st, v, lo, cnt);
            for (int pass = 0; pass < 8; pass++) {
                const u64 *kin = pass % 2 ? v.kB : v.kA;
                const int *vin = pass % 2 ? v.vB : v.vA;
                u64 *kout = pass % 2 ? v.kA : v.kB;
                int *vout = pass % 2 ? v.vA : v.vB;
                hipLaunchKernelGGL(ct_hist, dim3(info.n_tile), dim3(256), 0, st, v, lo, cnt, kin, pass * 8);
                hipLaunchKernelGGL(ct_tile_scan, dim3(cnt), dim3(256), 0, st, v, lo, cnt);
                hipLaunchKernelGGL(ct_scatter, dim3(info.n_tile), dim3(256), 0, st, v, lo, cnt, kin, vin, kout, vout, pass * 8);
            }
            hipLaunchKernelGGL(ct_split_sizes, grid1(cnt, 64), dim3(64), 0, st, v, lo, cnt, pieces, force ? 1 : 0, regular ? 1 : 0);
            hipLaunchKernelGGL(ct_commit, dim3(info.n_tile), dim3(256), 0, st, v, lo, cnt);
        }
        hipLaunchKernelGGL(ct_assign_children, dim3(1), dim3(1024), 0, st, v, pieces, partition_level ? 1 : 0);
        HIP_OK(hipGetLastError());
        read_info();
        geometry();
    };

    hipLaunchKernelGGL(ct_init, dim3(1), dim3(1024), 0, st, v, 1);
    HIP_OK(hipGetLastError());
    read_info();
    geometry(); // (the root's sums run over the points in the caller's order, as on the host: a given global partition reorders them afterwards)
    if (!given_perm.empty()) HIP_OK(hipMemcpyAsync(v.perm, given_perm.data(), (size_t)N * 4, hipMemcpyHostToDevice, st));
    if (P > 1) {
        if (!given_sizes.empty()) {
            const int one = 1;
            HIP_OK(hipMemcpyAsync(v.csize, given_sizes.data(), (size_t)P * 4, hipMemcpyHostToDevice, st));
            HIP_OK(hipMemcpyAsync(v.ok, &one, 4, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(ct_assign_children, dim3(1), dim3(1024), 0, st, v, P, 1);
            HIP_OK(hipGetLastError());
            read_info();
            geometry();
        } else {
            split_level(P, true, true);
        }
    }
    while (info.lvl_cnt > 0 && info.max_size / nc >= a.max_leaf) split_level(nc, false, false);
    HIP_OK(hipStreamSynchronize(st));

    // the tables come back to the host: the block tree and the layout are built from them
    const int nn = info.n_nodes;
    T.perm.resize(N);
    HIP_OK(hipMemcpyAsync(T.perm.data(), v.perm, (size_t)N * 4, hipMemcpyDeviceToHost, st));
    std::vector<int> *iv[7] = {&T.offset, &T.size, &T.depth, &T.parent, &T.first_child, &T.n_child, &T.partition};
    for (int i = 0; i < 7; i++) {
        iv[i]->resize(nn);
        HIP_OK(hipMemcpyAsync(iv[i]->data(), *ip[i], (size_t)nn * 4, hipMemcpyDeviceToHost, st));
    }
    std::vector<double> *dv[4] = {&T.cx, &T.cy, &T.cz, &T.radius};
    for (int i = 0; i < 4; i++) {
        dv[i]->resize(nn);
        HIP_OK(hipMemcpyAsync(dv[i]->data(), *dp[i], (size_t)nn * 8, hipMemcpyDeviceToHost, st));
    }
    HIP_OK(hipStreamSynchronize(st));
    if (P == 1) T.part_nodes.push_back(0);
    else
        for (int p = 0; p < P; p++) T.part_nodes.push_back(1 + p);
    return Tp.release();
}

} // namespace hm

// library warm-up (device.hip: device_warm_up): the first launch of a kernel of this translation unit loads its code object
namespace hm {
__global__ void warm_kernel_cluster() {}
void warm_up_cluster() { hipLaunchKernelGGL(warm_kernel_cluster, dim3(1), dim3(64), 0, 0); }
} // namespace hm
