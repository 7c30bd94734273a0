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
// One translation unit, in pieces: product_kernels.inc (the kernels above), pack_kernels.inc (per-leaf factors
// <-> tile panels, device-evaluated generators), device_memory.inc (workspace cache of the large temporary
// buffers), device_tables.inc (batch packing and table assembly), this file (host drivers of the product, copy,
// introspection), device_build.inc (device ACA and the native build), device_recompress.inc (SVD recompression).
#include "device_internal.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <exception>
#include <array>
#include <map>
#include <mutex>
#include <numeric>
#include <thread>
#include <type_traits>

namespace hm {

#include "product_kernels.inc"
#include "product_mfma.inc"
#include "pack_kernels.inc"

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

int device_current() { return g_device; }

// the leaf records of a device-resident build come to the host on first use
void HMatrix::materialise_blocks() const {
    std::lock_guard<std::mutex> lock(blocks_mu);
    if (!blocks_lazy) return;
    DeviceHMatrix *D = dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    HIP_OK(hipSetDevice(D->device));
    size_t total = 0;
    for (const LeafTable &lt : D->leaf_tables) total += lt.n;
    blocks_.clear();
    blocks_.resize(total);
    size_t base = 0;
    for (LeafTable &lt : D->leaf_tables) {
        std::unique_ptr<DevBlock[]> hb(new DevBlock[std::max<size_t>(lt.n, 1)]);
        std::unique_ptr<int2[]> hn(new int2[std::max<size_t>(lt.n, 1)]);
        if (lt.n) {
            HIP_OK(hipMemcpy(hb.get(), lt.blocks, lt.n * sizeof(DevBlock), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(hn.get(), lt.nodes, lt.n * sizeof(int2), hipMemcpyDeviceToHost));
        }
        const DevBlock *pb = hb.get();
        const int2 *pn = hn.get();
        BlockRec *out = blocks_.data() + base;
        const int batch = lt.batch;
        parallel_for((long long)lt.n, [&](long long q) {
            const DevBlock &d = pb[q];
            BlockRec b;
            b.t_node = pn[q].x; b.s_node = pn[q].y;
            b.t_off = d.t_off; b.m = d.m; b.s_off = d.s_off; b.n = d.n; b.rank = d.rank; b.cap = d.cap;
            b.batch = d.rank == 0 ? -1 : batch;
            b.tmp_u = d.tmp_u; b.tmp_v = d.tmp_v; b.ucol = d.ucol; b.vcol = d.vcol; b.tpos = d.tpos; b.v_obase = d.v_obase; b.v_ostride = d.v_ostride;
            b.status = d.status; b.z_obase = d.z_obase; b.z_ostride = d.z_ostride; b.zfin = d.zfin;
            out[q] = b;
        });
        base += lt.n;
        (void)hipFree(lt.blocks);
        (void)hipFree(lt.nodes);
        lt.blocks = nullptr; lt.nodes = nullptr; lt.n = 0;
    }
    D->leaf_tables.clear();
    blocks_lazy = false;
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


#include "device_memory.inc"
#include "device_tables.inc"

// ------------------------------------------------------------------------------------------------
void device_build_from_host(HMatrix &H, const void *arena, int64_t arena_elems) {
    DeviceBuilder db(H);
    size_t es = H.is_complex ? 16 : 8;
    void *d_arena = nullptr;
    HIP_OK(dev_malloc(&d_arena, std::max<int64_t>(arena_elems, 1) * es));
    if (arena_elems) HIP_OK(hipMemcpy(d_arena, arena, arena_elems * es, hipMemcpyHostToDevice));
    std::vector<int64_t> all;
    for (size_t i = 0; i < H.blocks().size(); i++) if (H.blocks()[i].rank != 0) all.push_back((int64_t)i);
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
    if (D->nA) hipLaunchKernelGGL((tile_gemv_tall_grouped<Ops, 16, NR>), dim3(D->nA), dim3(256), 0, st, D->tilesA, D->segs, (const T *)W, W, ws, ws);
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
        t += cnt[2];
        // tiles of at most 8 lanes (16 real rows: the reference's default leaf size 10): eight columns per wave instruction, batches of 8 loads
        if (cnt[3]) hipLaunchKernelGGL((tile_gemv_wide<Ops, 8, NR, 8>), dim3(cnt[3]), dim3(256), 0, st, t, D->segs, (const T *)W, out, ws, out_stride);
    };
    if (D->one_triangle) {
        // fused sweep: y (cluster numbering) and the transposed dot products of every column, then the transposed
        // use of the V panels, then the transposed dense results and the scatter to the caller's numbering
        if constexpr (NR <= 4) {
            T *ycl = (T *)D->ycl;
            const long long ys = D->ycl_stride;
            const int cj = D->conj_transposed ? 1 : 0;
            auto launch_wide_sym = [&](const GTile *tiles, const int *cnt, T *dst, long long dst_stride) { // one launch per lane-packing class
                const GTile *t = tiles;
                if (cnt[0]) hipLaunchKernelGGL((tile_gemv_wide_sym<Ops, 1, NR>), dim3(cnt[0]), dim3(256), 0, st, t, D->segs, (const T *)W, W, dst, cj, ws, dst_stride, (const T *)W, ws);
                t += cnt[0];
                if (cnt[1]) hipLaunchKernelGGL((tile_gemv_wide_sym<Ops, 2, NR>), dim3(cnt[1]), dim3(256), 0, st, t, D->segs, (const T *)W, W, dst, cj, ws, dst_stride, (const T *)W, ws);
                t += cnt[1];
                // (the fused kernel has no 8-column class: the tiles of classes 2 and 3 are consecutive and both fit its F = 4 form)
                if (cnt[2] + cnt[3]) hipLaunchKernelGGL((tile_gemv_wide_sym<Ops, 4, NR>), dim3(cnt[2] + cnt[3]), dim3(256), 0, st, t, D->segs, (const T *)W, W, dst, cj, ws, dst_stride, (const T *)W, ws);
            };
            if (D->splitB > 1 && D->nB_split) { // small operator: column slices of the row tiles, summed in slice order
                launch_wide_sym(D->tilesB_split, D->cntBs, (T *)D->ypart, (long long)D->splitB * D->ypart_stride);
                hipLaunchKernelGGL(reduce_y_kernel<T>, dim3((D->row_size + 255) / 256), dim3(256), 0, st, (const T *)D->ypart, D->ypart_stride, D->splitB, D->row_size,
                                   (const int *)nullptr, ycl, ys, NR);
            } else if (D->nB) {
                launch_wide_sym(D->tilesB_cluster, D->cntB, ycl, ys);
            }
            if (D->nZ) hipLaunchKernelGGL((tile_gemv_tall<Ops, 16, NR>), dim3(D->nZ), dim3(256), 0, st, D->tilesZ, D->segs, (const T *)W, W, ws, ws, ws);
            if (D->nAT) {
                if constexpr (NR == 1) hipLaunchKernelGGL((tile_gemv_tall_transposed<Ops>), dim3(D->nAT), dim3(256), 0, st, D->tilesAT, D->segs, (const T *)W, ycl, cj);
                else hipLaunchKernelGGL((tile_gemv_tall_transposed_multi<Ops, NR>), dim3(D->nAT), dim3(256), 0, st, D->tilesAT, D->segs, (const T *)W, ycl, cj, ws, ys);
            }
            if (D->n_zd_tiles) hipLaunchKernelGGL(finish_sym_kernel<T>, dim3(D->n_zd_tiles), dim3(128), 0, st, (const T *)ycl, (const T *)W, D->zd_ptr, D->zd_woff, D->zd_rows,
                                                  out_user ? D->perm_t : (const int *)nullptr, (T *)y_dev, NR, ys, ws, y_stride);
        } else {
            throw Error("one-triangle storage: at most four right-hand sides per fused sweep");
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

// y = H^T x (conj: H^H x) for an operator that stores both triangles, with the kernels of the one-triangle product:
//   x -> cluster numbering of the rows;  phase-B panels, transposed dot products only (z = U^T x per leaf, D^T x per dense leaf);
//   sums of the z partials of target nodes spanning several row tiles;  phase-A panels transposed (y += V^T z);
//   dense contributions added and y scattered to the caller's numbering, one workgroup per source tile.
// Every sum has a fixed order: bitwise reproducible like the direct product.  At most four right-hand sides per sweep.
template <typename Ops, int NR>
static void launch_sweep_T(DeviceHMatrix *D, const void *x_dev, long long x_stride, void *y_dev, long long y_stride, int numbering, bool conj, hipStream_t st) {
    typedef typename Ops::T T;
    T *W = (T *)D->W;
    const long long ws = D->W_elems, xs = D->xt_stride, ys = D->ycl_stride;
    T *XT = (T *)D->xt, *ycl = (T *)D->ycl;
    const bool in_user = numbering == 0 || numbering == 2, out_user = numbering == 0 || numbering == 3;
    const int cj = conj ? 1 : 0;
    if (D->row_size) hipLaunchKernelGGL(gather_x_kernel<T>, dim3((D->row_size + 255) / 256), dim3(256), 0, st, (const T *)x_dev, x_stride, in_user ? D->perm_t + D->row_off : (const int *)nullptr, XT, xs, D->row_size, NR);
    const T *Xrows = XT - D->row_off; // (the row tiles carry target positions of the whole cluster tree)
    const bool split = D->splitB > 1 && D->nB_split;
    const GTile *t = split ? D->tilesB_split : D->tilesB_cluster;
    const int *cnt = split ? D->cntBs : D->cntB;
    if (cnt[0]) hipLaunchKernelGGL((tile_gemv_wide_sym<Ops, 1, NR, false>), dim3(cnt[0]), dim3(256), 0, st, t, D->segs, (const T *)W, W, (T *)nullptr, cj, ws, 0LL, Xrows, xs);
    t += cnt[0];
    if (cnt[1]) hipLaunchKernelGGL((tile_gemv_wide_sym<Ops, 2, NR, false>), dim3(cnt[1]), dim3(256), 0, st, t, D->segs, (const T *)W, W, (T *)nullptr, cj, ws, 0LL, Xrows, xs);
    t += cnt[1];
    if (cnt[2] + cnt[3]) hipLaunchKernelGGL((tile_gemv_wide_sym<Ops, 4, NR, false>), dim3(cnt[2] + cnt[3]), dim3(256), 0, st, t, D->segs, (const T *)W, W, (T *)nullptr, cj, ws, 0LL, Xrows, xs);
    if (D->nZ) hipLaunchKernelGGL((tile_gemv_tall<Ops, 16, NR>), dim3(D->nZ), dim3(256), 0, st, D->tilesZ, D->segs, (const T *)W, W, ws, ws, ws);
    HIP_OK(hipMemsetAsync(ycl, 0, (size_t)NR * ys * sizeof(T), st));
    if (D->nAT) {
        if constexpr (NR == 1) hipLaunchKernelGGL((tile_gemv_tall_transposed<Ops>), dim3(D->nAT), dim3(256), 0, st, D->tilesAT, D->segs, (const T *)W, ycl, cj);
        else hipLaunchKernelGGL((tile_gemv_tall_transposed_multi<Ops, NR>), dim3(D->nAT), dim3(256), 0, st, D->tilesAT, D->segs, (const T *)W, ycl, cj, ws, ys);
    }
    if (D->n_zd_tiles) hipLaunchKernelGGL(finish_sym_kernel<T>, dim3(D->n_zd_tiles), dim3(128), 0, st, (const T *)ycl, (const T *)W, D->zd_ptr, D->zd_woff, D->zd_rows,
                                          out_user ? D->perm_s : (const int *)nullptr, (T *)y_dev, NR, ys, ws, y_stride);
    HIP_OK(hipGetLastError());
}

// forget the captured product (its kernel arguments point into tables / workspaces that are about to change)
static void drop_product_graph(DeviceHMatrix *D) {
    if (D->graph.exec) {
        // a replay launched by an earlier call may still be running on the caller's stream (call pattern A, A, A, B): the exec
        // graph is destroyed only once that stream has passed it
        if (D->graph.stream) (void)hipStreamSynchronize(D->graph.stream);
        (void)hipGraphExecDestroy(D->graph.exec);
    }
    D->graph = ProductGraph();
}

// the first segment of every tile of a reduction table points into the coefficient workspace: move it to the new workspace
__global__ void relocate_segments_kernel(const GTile *tiles, int n, GSeg *segs, const char *old_base, const char *new_base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    GSeg &sg = segs[tiles[i].seg_begin];
    sg.panel = new_base + ((const char *)sg.panel - old_base);
}

// make room for nr coefficient workspaces (and partial-y slabs); W[r][n_source] = 1 for every r
template <typename T>
static void ensure_rhs_capacity(DeviceHMatrix *D, int nr) {
    if (nr <= D->rhs_cap) return;
    drop_product_graph(D);
    void *nW = nullptr;
    HIP_OK(dev_malloc(&nW, (size_t)nr * D->W_elems * sizeof(T)));
    HIP_OK(hipMemset(nW, 0, (size_t)nr * D->W_elems * sizeof(T)));
    T one;
    std::memset(&one, 0, sizeof(T));
    *(double *)&one = 1.0;
    {
        std::vector<T> ones((size_t)nr, one);
        HIP_OK(hipMemcpy2D((char *)nW + (size_t)D->n_source * sizeof(T), (size_t)D->W_elems * sizeof(T), ones.data(), sizeof(T), sizeof(T), (size_t)nr, hipMemcpyHostToDevice));
    }
    // the A2 segments, and the panels of the transposed partial sums (one-triangle storage), point into W: relocate them on the
    // device, one launch per table (round 3 read and rewrote one 64-byte record per tile with blocking copies: some 30 000 of
    // them at 1 M points, longer than the build of the operator)
    if (D->W && D->nA2) hipLaunchKernelGGL(relocate_segments_kernel, dim3((unsigned)((D->nA2 + 255) / 256)), dim3(256), 0, 0, D->tilesA2, D->nA2, D->segs, (const char *)D->W, (const char *)nW);
    if (D->W && D->nZ) hipLaunchKernelGGL(relocate_segments_kernel, dim3((unsigned)((D->nZ + 255) / 256)), dim3(256), 0, 0, D->tilesZ, D->nZ, D->segs, (const char *)D->W, (const char *)nW);
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    if (D->W) (void)hipFree(D->W);
    D->W = nW;
    if (D->one_triangle || D->transposable) {
        if (D->ycl) (void)hipFree(D->ycl);
        HIP_OK(dev_malloc(&D->ycl, (size_t)nr * std::max<long long>(D->ycl_stride, 1) * sizeof(T)));
    }
    if (D->transposable) {
        if (D->xt) (void)hipFree(D->xt);
        HIP_OK(dev_malloc(&D->xt, (size_t)nr * std::max<long long>(D->xt_stride, 1) * sizeof(T)));
        HIP_OK(hipMemset(D->xt, 0, (size_t)nr * std::max<long long>(D->xt_stride, 1) * sizeof(T)));
    }
    if (D->splitB > 1) {
        if (D->ypart) (void)hipFree(D->ypart);
        HIP_OK(dev_malloc(&D->ypart, (size_t)nr * D->splitB * D->ypart_stride * sizeof(T)));
    }
    D->rhs_cap = nr;
}

// ---- sixteen right-hand sides per sweep on the matrix cores ----
// the partial sums a reduction pass adds, as index ranges of the coefficient workspace (read back from its tile table)
static std::vector<Reduce16> reduce16_items(DeviceHMatrix *D, const GTile *tiles, int n) {
    std::vector<Reduce16> items;
    if (!n) return items;
    std::vector<GTile> t((size_t)n);
    HIP_OK(hipMemcpy(t.data(), tiles, t.size() * sizeof(GTile), hipMemcpyDeviceToHost));
    std::vector<GSeg> segs((size_t)D->n_segs); // the whole table in one copy (not one blocking copy per tile)
    HIP_OK(hipMemcpy(segs.data(), D->segs, segs.size() * sizeof(GSeg), hipMemcpyDeviceToHost));
    for (const GTile &x : t) {
        HM_CHECK(x.seg_begin >= 0 && x.seg_begin < (long long)segs.size(), "internal error: reduction tile outside the segment table");
        const GSeg &sg = segs[(size_t)x.seg_begin];
        Reduce16 r;
        r.w_panel = ((const char *)sg.panel - (const char *)D->W) / (long long)D->esize;
        r.out_base = x.out_begin;
        r.ld = sg.ld_last; r.nrows = sg.nrows_t; r.ncols = sg.ncols;
        for (r.row0 = 0; r.row0 < r.nrows; r.row0 += 16) items.push_back(r); // one work item per 16 rows
    }
    return items;
}

static void ensure_w16(DeviceHMatrix *D) {
    if (D->W16) return;
    drop_product_graph(D);
    void *w = nullptr;
    HIP_OK(dev_malloc(&w, (size_t)D->W_elems * 16 * D->esize));
    HIP_OK(hipMemset(w, 0, (size_t)D->W_elems * 16 * D->esize));
    const std::vector<Reduce16> items = reduce16_items(D, D->tilesA2, D->nA2); // phase A2
    D->red16 = upload(items);
    D->n_red16 = (int)items.size();
    if (D->one_triangle || D->transposable) { // the tables of the fused / transposed sweeps, 16 wide
        const std::vector<Reduce16> zitems = reduce16_items(D, D->tilesZ, D->nZ);
        D->redz16 = upload(zitems);
        D->n_redz16 = (int)zitems.size();
        HIP_OK(dev_malloc(&D->ycl16, (size_t)std::max<long long>(D->ycl_stride, 1) * 16 * D->esize));
        if (D->transposable) {
            HIP_OK(dev_malloc(&D->xt16, (size_t)std::max<long long>(D->xt_stride, 1) * 16 * D->esize));
            HIP_OK(hipMemset(D->xt16, 0, (size_t)std::max<long long>(D->xt_stride, 1) * 16 * D->esize));
        }
        // the finishing pass is a gather over the destinations: the tile of every position of the accumulator, and the inverse of
        // the permutation the results leave by (target side for one-triangle storage, source side for the transposed product)
        if (D->n_zd_tiles) {
            std::vector<int> rows(2 * (size_t)D->n_zd_tiles);
            HIP_OK(hipMemcpy(rows.data(), D->zd_rows, rows.size() * sizeof(int), hipMemcpyDeviceToHost));
            int npos = 0;
            for (int t = 0; t < D->n_zd_tiles; t++) npos = std::max(npos, rows[2 * t] + rows[2 * t + 1]);
            std::vector<int> tile_of((size_t)npos, 0), perm((size_t)npos), iperm((size_t)npos, 0);
            for (int t = 0; t < D->n_zd_tiles; t++)
                for (int i = 0; i < rows[2 * t + 1]; i++) tile_of[(size_t)rows[2 * t] + i] = t;
            HIP_OK(hipMemcpy(perm.data(), D->one_triangle ? D->perm_t : D->perm_s, perm.size() * sizeof(int), hipMemcpyDeviceToHost));
            bool is_perm = true;
            std::vector<char> seen((size_t)npos, 0);
            for (int p = 0; p < npos; p++) {
                if (perm[p] < 0 || perm[p] >= npos || seen[(size_t)perm[p]]) { is_perm = false; break; }
                seen[(size_t)perm[p]] = 1;
                iperm[(size_t)perm[p]] = p;
            }
            HM_CHECK(is_perm, "internal error: the output numbering of the fused sweep is not a permutation of its positions");
            D->fin_tile_of = upload(tile_of);
            D->fin_iperm = upload(iperm);
            D->fin_npos = npos;
        }
    }
    D->W16 = w;
}

template <typename T>
static void launch_sweep16(DeviceHMatrix *D, const T *x, long long x_stride, T *y, long long y_stride, int nr, int numbering, hipStream_t st) {
    constexpr bool CPLX = sizeof(T) == 16;
    T *W16 = (T *)D->W16;
    const int Ns = D->n_source;
    const bool in_user = numbering == 0 || numbering == 2, out_user = numbering == 0 || numbering == 3;
    hipEvent_t *ev = D->pev[D->nprod % DeviceHMatrix::RING];
    const bool timing = D->phase_timing;
    if (timing) HIP_OK(hipEventRecord(ev[0], st));
    if (Ns) hipLaunchKernelGGL(gather_x16_kernel<T>, dim3((unsigned)((Ns + 63) / 64)), dim3(256), 0, st, x, x_stride, in_user ? D->perm_s : (const int *)nullptr, W16, Ns, nr);
    if (timing) HIP_OK(hipEventRecord(ev[1], st));
    if (D->nA) hipLaunchKernelGGL(tile_gemm_tall16<CPLX>, dim3(D->nA), dim3(256), 0, st, D->tilesA, D->segs, (double *)W16);
    if (timing) HIP_OK(hipEventRecord(ev[2], st));
    if (D->n_red16) hipLaunchKernelGGL(reduce_partials16_kernel<T>, dim3(D->n_red16), dim3(256), 0, st, (const Reduce16 *)D->red16, W16);
    if (timing) HIP_OK(hipEventRecord(ev[3], st));
    if (D->nB) hipLaunchKernelGGL(tile_gemm_wide16<CPLX>, dim3(D->nB), dim3(256), 0, st, out_user ? D->tilesB_user : D->tilesB_cluster, D->segs, (const double *)W16, (double *)y, y_stride, nr);
    if (timing) HIP_OK(hipEventRecord(ev[4], st));
    HIP_OK(hipGetLastError());
    if (timing) D->nprod++;
}

// one-triangle storage, sixteen right-hand sides: the passes of launch_sweep's fused branch, on the matrix cores
template <typename T>
static void launch_sweep16_sym(DeviceHMatrix *D, const T *x, long long x_stride, T *y, long long y_stride, int nr, int numbering, hipStream_t st) {
    constexpr bool CPLX = sizeof(T) == 16;
    T *W16 = (T *)D->W16;
    const int Ns = D->n_source;
    const bool in_user = numbering == 0 || numbering == 2, out_user = numbering == 0 || numbering == 3;
    const int cj = D->conj_transposed ? 1 : 0;
    hipEvent_t *ev = D->pev[D->nprod % DeviceHMatrix::RING];
    const bool timing = D->phase_timing;
    if (timing) HIP_OK(hipEventRecord(ev[0], st));
    if (Ns) hipLaunchKernelGGL(gather_x16_kernel<T>, dim3((unsigned)((Ns + 63) / 64)), dim3(256), 0, st, x, x_stride, in_user ? D->perm_s : (const int *)nullptr, W16, Ns, nr);
    if (timing) HIP_OK(hipEventRecord(ev[1], st));
    if (D->nA) hipLaunchKernelGGL(tile_gemm_tall16<CPLX>, dim3(D->nA), dim3(256), 0, st, D->tilesA, D->segs, (double *)W16);
    if (timing) HIP_OK(hipEventRecord(ev[2], st));
    if (D->n_red16) hipLaunchKernelGGL(reduce_partials16_kernel<T>, dim3(D->n_red16), dim3(256), 0, st, (const Reduce16 *)D->red16, W16);
    if (timing) HIP_OK(hipEventRecord(ev[3], st));
    if (D->nB) hipLaunchKernelGGL((tile_gemm_wide16_sym<CPLX, true>), dim3(D->nB), dim3(256), 0, st, D->tilesB_cluster, D->segs, (const double *)W16, (double *)W16, (double *)D->ycl16,
                                  (const double *)W16, cj);
    if (D->n_redz16) hipLaunchKernelGGL(reduce_partials16_kernel<T>, dim3(D->n_redz16), dim3(256), 0, st, (const Reduce16 *)D->redz16, W16);
    if (D->nAT) hipLaunchKernelGGL(tile_gemm_tall16_transposed<CPLX>, dim3(D->nAT), dim3(256), 0, st, D->tilesAT, D->segs, (const double *)W16, (double *)D->ycl16, cj);
    if (D->fin_npos) hipLaunchKernelGGL(finish_sym16_kernel<T>, dim3((unsigned)((D->fin_npos + 63) / 64)), dim3(256), 0, st, (const T *)D->ycl16, (const T *)W16, D->zd_ptr, D->zd_woff, D->zd_rows,
                                        D->fin_tile_of, out_user ? D->fin_iperm : (const int *)nullptr, y, D->fin_npos, nr, y_stride);
    if (timing) HIP_OK(hipEventRecord(ev[4], st));
    HIP_OK(hipGetLastError());
    if (timing) D->nprod++;
}

// y = H^T x (conj: H^H x) for sixteen right-hand sides (launch_sweep_T on the matrix cores)
template <typename T>
static void launch_sweep16_T(DeviceHMatrix *D, const T *x, long long x_stride, T *y, long long y_stride, int nr, int numbering, bool conj, hipStream_t st) {
    constexpr bool CPLX = sizeof(T) == 16;
    T *W16 = (T *)D->W16, *XT16 = (T *)D->xt16;
    const bool in_user = numbering == 0 || numbering == 2, out_user = numbering == 0 || numbering == 3;
    const int cj = conj ? 1 : 0;
    if (D->row_size) hipLaunchKernelGGL(gather_x16_kernel<T>, dim3((unsigned)((D->row_size + 63) / 64)), dim3(256), 0, st, x, x_stride, in_user ? D->perm_t + D->row_off : (const int *)nullptr, XT16, D->row_size, nr);
    const T *Xrows = XT16 - (long long)D->row_off * 16; // (the row tiles carry target positions of the whole cluster tree)
    if (D->nB) hipLaunchKernelGGL((tile_gemm_wide16_sym<CPLX, false>), dim3(D->nB), dim3(256), 0, st, D->tilesB_cluster, D->segs, (const double *)W16, (double *)W16, (double *)nullptr, (const double *)Xrows, cj);
    if (D->n_redz16) hipLaunchKernelGGL(reduce_partials16_kernel<T>, dim3(D->n_redz16), dim3(256), 0, st, (const Reduce16 *)D->redz16, W16);
    HIP_OK(hipMemsetAsync(D->ycl16, 0, (size_t)std::max<long long>(D->ycl_stride, 1) * 16 * sizeof(T), st));
    if (D->nAT) hipLaunchKernelGGL(tile_gemm_tall16_transposed<CPLX>, dim3(D->nAT), dim3(256), 0, st, D->tilesAT, D->segs, (const double *)W16, (double *)D->ycl16, cj);
    if (D->fin_npos) hipLaunchKernelGGL(finish_sym16_kernel<T>, dim3((unsigned)((D->fin_npos + 63) / 64)), dim3(256), 0, st, (const T *)D->ycl16, (const T *)W16, D->zd_ptr, D->zd_woff, D->zd_rows,
                                        D->fin_tile_of, out_user ? D->fin_iperm : (const int *)nullptr, y, D->fin_npos, nr, y_stride);
    HIP_OK(hipGetLastError());
}

static bool mfma_sweep_enabled() {
    const char *v = getenv("HTOOL_MULTI_RHS_KERNEL"); // "valu": the 8-wide VALU sweeps for every count (A/B comparisons, tests)
    return !(v && std::string(v) == "valu");
}

template <typename Ops>
static void launch_product(DeviceHMatrix *D, const void *X, long long x_stride, void *Y, long long y_stride, int mu, int numbering, hipStream_t st) {
    typedef typename Ops::T T;
    int done = 0;
    while (done < mu) {
        const int left = mu - done;
        const T *x = (const T *)X + (long long)done * x_stride;
        T *y = (T *)Y + (long long)done * y_stride;
        if (D->one_triangle) { // fused sweeps: sixteen right-hand sides on the matrix cores, up to four on the vector units
            if (left > 8 && D->W16 && D->ycl16) {
                const int nr = std::min(left, 16);
                launch_sweep16_sym<T>(D, x, x_stride, y, y_stride, nr, numbering, st);
                done += nr;
                continue;
            }
            if (left >= 4) { launch_sweep<Ops, 4>(D, x, x_stride, y, y_stride, numbering, st); done += 4; }
            else if (left >= 2) { launch_sweep<Ops, 2>(D, x, x_stride, y, y_stride, numbering, st); done += 2; }
            else { launch_sweep<Ops, 1>(D, x, x_stride, y, y_stride, numbering, st); done += 1; }
            continue;
        }
        // more than 8 columns left: one sweep of the panels for up to 16 right-hand sides, on the matrix cores
        if (left > 8 && D->W16) {
            const int nr = std::min(left, 16);
            launch_sweep16<T>(D, x, x_stride, y, y_stride, nr, numbering, st);
            done += nr;
            continue;
        }
        if (left >= 8) { launch_sweep<Ops, 8>(D, x, x_stride, y, y_stride, numbering, st); done += 8; }
        else if (left >= 4) { launch_sweep<Ops, 4>(D, x, x_stride, y, y_stride, numbering, st); done += 4; }
        else if (left >= 2) { launch_sweep<Ops, 2>(D, x, x_stride, y, y_stride, numbering, st); done += 2; }
        else { launch_sweep<Ops, 1>(D, x, x_stride, y, y_stride, numbering, st); done += 1; }
    }
}

// Y = H X for mu right-hand sides stored with the given strides (elements); mu = 1 is the matvec
// make the tables of the transposed product (once per operator): the slots of the transposed dot products are added to the
// coefficient workspace, every index array is written again (the panels are not touched) and the product tables re-assembled
void device_make_transposable(HMatrix &H) {
    DeviceHMatrix *D = H.dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    std::lock_guard<std::recursive_mutex> lock(D->mu);
    if (H.transposable || H.one_triangle) return;
    HIP_OK(hipSetDevice(D->device));
    HIP_OK(hipDeviceSynchronize()); // products in flight (on any stream) still read the tables that are replaced below
    std::vector<std::vector<int64_t>> per_batch(D->batches.size());
    for (size_t i = 0; i < H.blocks().size(); i++) {
        const BlockRec &b = H.blocks()[i];
        if (b.rank == 0) continue;
        HM_CHECK(b.batch >= 0 && b.batch < (int)per_batch.size(), "internal error: leaf without a batch");
        per_batch[b.batch].push_back((int64_t)i);
    }
    DeviceBuilder db(H, D);
    const bool timing = D->phase_timing;
    db.free_product_tables();
    H.transposable = true;
    H.r_elems = 0;
    try {
        for (size_t b = 0; b < per_batch.size(); b++) {
            if (H.is_complex) db.reindex_batch<double2>(per_batch[b], (int)b);
            else db.reindex_batch<double>(per_batch[b], (int)b);
        }
        if (H.is_complex) db.assemble<double2>();
        else db.assemble<double>();
    } catch (...) {
        H.transposable = false; // (the operator is unusable after a failure here: an allocation failed between tables)
        throw;
    }
    D->phase_timing = timing;
}

template <typename Ops>
static void launch_product_T(DeviceHMatrix *D, const void *X, long long x_stride, void *Y, long long y_stride, int mu, int numbering, bool conj, hipStream_t st) {
    typedef typename Ops::T T;
    int done = 0;
    while (done < mu) {
        const int left = mu - done;
        const T *x = (const T *)X + (long long)done * x_stride;
        T *y = (T *)Y + (long long)done * y_stride;
        if (left > 8 && D->W16 && D->xt16) {
            const int nr = std::min(left, 16);
            launch_sweep16_T<T>(D, x, x_stride, y, y_stride, nr, numbering, conj, st);
            done += nr;
            continue;
        }
        if (left >= 4) { launch_sweep_T<Ops, 4>(D, x, x_stride, y, y_stride, numbering, conj, st); done += 4; }
        else if (left >= 2) { launch_sweep_T<Ops, 2>(D, x, x_stride, y, y_stride, numbering, conj, st); done += 2; }
        else { launch_sweep_T<Ops, 1>(D, x, x_stride, y, y_stride, numbering, conj, st); done += 1; }
    }
}

void device_matmat_device(const HMatrix &H, const void *X, long long x_stride, void *Y, long long y_stride, int mu, int numbering, void *stream, char trans) {
    DeviceHMatrix *D = H.dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    HM_CHECK(trans == 'N' || trans == 'T' || trans == 'C', "H-matrix product: trans must be 'N', 'T' or 'C'");
    // products of one handle share its coefficient workspace (and may re-allocate it): callers on several host threads are
    // serialised here (launches only; the kernels of two calls on ONE stream run in order anyway)
    std::lock_guard<std::recursive_mutex> lock(D->mu);
    HIP_OK(hipSetDevice(D->device));
    hipStream_t st = stream ? (hipStream_t)stream : D->stream;
    if (trans == 'C' && !D->is_complex) trans = 'T';
    if (trans != 'N' && D->one_triangle) {
        // one triangle stored: H = H^T ('S') or H = H^H ('H'), the product with the stored operator is the answer
        HM_CHECK((trans == 'T' && !D->conj_transposed) || (trans == 'C' && D->conj_transposed),
                 "H-matrix product: for one-triangle storage only the transposition that leaves the operator unchanged is implemented ('T' for 'S', 'C' for 'H')");
        trans = 'N';
    }
    if (trans != 'N') {
        if (!H.transposable) device_make_transposable(const_cast<HMatrix &>(H)); // (a cache of tables, made under the handle's lock)
        const int need_t = mu >= 4 ? 4 : mu >= 2 ? 2 : 1;
        if (mu > 8 && !D->W16 && mfma_sweep_enabled()) {
            HIP_OK(hipStreamSynchronize(st));
            ensure_w16(D);
        }
        if (need_t > D->rhs_cap) {
            HIP_OK(hipStreamSynchronize(st));
            if (D->is_complex) ensure_rhs_capacity<double2>(D, need_t);
            else ensure_rhs_capacity<double>(D, need_t);
        }
        if (D->is_complex) launch_product_T<CplxOps>(D, X, x_stride, Y, y_stride, mu, numbering, trans == 'C', st);
        else launch_product_T<RealOps>(D, X, x_stride, Y, y_stride, mu, numbering, false, st);
        return;
    }
    const int need = D->one_triangle ? (mu >= 4 ? 4 : mu >= 2 ? 2 : 1) : (mu >= 8 ? 8 : mu >= 4 ? 4 : mu >= 2 ? 2 : 1);
    if (mu > 8 && !D->W16 && mfma_sweep_enabled()) {
        HIP_OK(hipStreamSynchronize(st));
        ensure_w16(D);
    }
    if (need > D->rhs_cap) {
        HIP_OK(hipStreamSynchronize(st));
        if (D->is_complex) ensure_rhs_capacity<double2>(D, need);
        else ensure_rhs_capacity<double>(D, need);
    }
    auto launch = [&]() {
        if (D->is_complex) launch_product<CplxOps>(D, X, x_stride, Y, y_stride, mu, numbering, st);
        else launch_product<RealOps>(D, X, x_stride, Y, y_stride, mu, numbering, st);
    };
    // Small operators (the per-GPU share of a distributed run, Krylov loops on fixed buffers): a product is 4-6 short launches;
    // the second request for the SAME product (buffers, stream) is captured as a hipGraph and replayed afterwards.
    static const bool graphs_on = !(getenv("HTOOL_PRODUCT_GRAPH") && std::string(getenv("HTOOL_PRODUCT_GRAPH")) == "0");
    // Only on a stream the CALLER named: with stream = NULL the kernels run on the handle's own stream and are ordered against
    // the caller's default-stream work by the legacy default stream's implicit synchronisation -- which a graph launch does not take part in.
    const bool graphable = graphs_on && !D->phase_timing && D->row_size <= 300000 && stream != nullptr;
    if (!graphable) { launch(); return; }
    ProductGraph &g = D->graph;
    const bool same = g.x == X && g.y == Y && g.x_stride == x_stride && g.y_stride == y_stride && g.mu == mu && g.numbering == numbering && g.stream == st;
    if (same && g.exec) {
        HIP_OK(hipGraphLaunch(g.exec, st));
        g.replays++;
        return;
    }
    if (!same) {
        drop_product_graph(D);
        g.x = X; g.y = Y; g.x_stride = x_stride; g.y_stride = y_stride; g.mu = mu; g.numbering = numbering; g.stream = st;
        launch();
        return;
    }
    hipGraph_t captured = nullptr;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { // (a stream that cannot be captured: stay eager)
        (void)hipGetLastError();
        launch();
        return;
    }
    try {
        launch();
    } catch (...) {
        (void)hipStreamEndCapture(st, &captured);
        if (captured) (void)hipGraphDestroy(captured);
        throw;
    }
    HIP_OK(hipStreamEndCapture(st, &captured));
    const hipError_t e = hipGraphInstantiate(&g.exec, captured, nullptr, nullptr, 0);
    (void)hipGraphDestroy(captured);
    if (e != hipSuccess) { (void)hipGetLastError(); g.exec = nullptr; g.x = nullptr; launch(); return; }
    HIP_OK(hipGraphLaunch(g.exec, st));
}

void device_matvec_device(const HMatrix &H, const void *x_dev, void *y_dev, int numbering, void *stream, char trans) {
    device_matmat_device(H, x_dev, 0, y_dev, 0, 1, numbering, stream, trans);
}

// host API convention: a matrix built on the whole target (source) cluster takes/returns user numbering on
// that side; one built on a partition works on its local slice in cluster order
static int host_numbering(const HMatrix &H, char trans = 'N') {
    bool in_user = H.s_root == 0 && !H.local_numbering, out_user = H.t_root == 0 && !H.local_numbering;
    if (trans != 'N') std::swap(in_user, out_user); // x lives on the target side then
    return in_user ? (out_user ? 0 : 2) : (out_user ? 3 : 1);
}

void device_matvec_host(const HMatrix &H, const void *x, void *y, char trans) {
    if (trans != 'N') { device_matmat_host(H, x, 1, y, trans); return; }
    DeviceHMatrix *D = H.dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    std::lock_guard<std::recursive_mutex> lock(D->mu); // x_tmp / y_tmp and the stream are per handle: one host product at a time
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
void device_matmat_host(const HMatrix &H, const void *X, int mu, void *Y, char trans) {
    DeviceHMatrix *D = H.dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    std::lock_guard<std::recursive_mutex> lock(D->mu);
    HIP_OK(hipSetDevice(D->device));
    const size_t es = D->esize;
    const bool whole = H.t_root == 0;
    size_t nin = (size_t)D->n_source, nout = (size_t)(whole ? D->n_target : D->row_size);
    if (trans != 'N') std::swap(nin, nout); // x has one entry per row of the operator, y one per column
    TempPool tmp;
    void *dX = tmp.alloc(std::max<size_t>(nin * mu, 1) * es), *dY = tmp.alloc(std::max<size_t>(nout * mu, 1) * es);
    HIP_OK(hipMemcpyAsync(dX, X, nin * mu * es, hipMemcpyHostToDevice, D->stream));
    device_matmat_device(H, dX, (long long)nin, dY, (long long)nout, mu, host_numbering(H, trans), D->stream, trans);
    HIP_OK(hipMemcpyAsync(Y, dY, nout * mu * es, hipMemcpyDeviceToHost, D->stream));
    HIP_OK(hipStreamSynchronize(D->stream));
}

void device_set_phase_timing(const HMatrix &H, bool on) {
    HM_CHECK(H.dev != nullptr, "H-matrix has no device data");
    drop_product_graph(H.dev);
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
    drop_product_graph(D);
    for (auto &B : D->batches) {
        (void)hipFree(B.panelB); (void)hipFree(B.panelA); (void)hipFree(B.cidxB); (void)hipFree(B.oidxA);
        if (B.zidxB) (void)hipFree(B.zidxB);
        if (B.tidxA) (void)hipFree(B.tidxA);
    }
    for (LeafTable &lt : D->leaf_tables) {
        if (lt.blocks) (void)hipFree(lt.blocks);
        if (lt.nodes) (void)hipFree(lt.nodes);
    }
    for (void *p : {(void *)D->tilesAT, (void *)D->tilesZ, (void *)D->zd_ptr, (void *)D->zd_woff, (void *)D->zd_rows, D->ycl, D->W16, D->red16, D->xt, D->redz16, D->ycl16, D->xt16, (void *)D->fin_tile_of, (void *)D->fin_iperm})
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
    // The copy starts EMPTY (no pointer of the source in it) and is owned by a guard until it is complete: when an
    // allocation fails half-way -- out of memory is the realistic case for a multi-GB operator -- only buffers the copy
    // itself allocated are freed, and the source stays intact.
    struct Guard {
        DeviceHMatrix *d;
        ~Guard() { if (d) device_free(d); }
    } guard{new DeviceHMatrix};
    DeviceHMatrix *D = guard.d;
    // scalar and host-side state
    D->device = S->device; D->tabs = S->tabs; D->is_complex = S->is_complex; D->esize = S->esize;
    D->nB = S->nB; D->nA = S->nA; D->nA2 = S->nA2; D->splitB = S->splitB; D->nB_split = S->nB_split;
    D->one_triangle = S->one_triangle; D->conj_transposed = S->conj_transposed; D->nAT = S->nAT; D->nZ = S->nZ; D->n_zd_tiles = S->n_zd_tiles;
    D->transposable = S->transposable; D->xt_stride = S->xt_stride;
    D->ycl_stride = S->ycl_stride; D->ypart_stride = S->ypart_stride; D->W_elems = S->W_elems; D->rhs_cap = S->rhs_cap;
    for (int c = 0; c < 4; c++) { D->cntB[c] = S->cntB[c]; D->cntBs[c] = S->cntBs[c]; }
    D->n_source = S->n_source; D->n_target = S->n_target; D->row_off = S->row_off; D->row_size = S->row_size; D->table_bytes = S->table_bytes;
    D->batches.resize(S->batches.size());
    for (size_t b = 0; b < S->batches.size(); b++) D->batches[b].bytes = S->batches[b].bytes;
    HIP_OK(hipStreamCreate(&D->stream));
    for (auto &slot : D->pev) for (auto &e : slot) HIP_OK(hipEventCreate(&e));
    // Re-pack is not possible (the arena is gone), so copy buffers and relocate pointers in the tables.
    struct Range { const char *old_lo, *old_hi; char *neu; };
    std::vector<Range> map;
    // every buffer is assigned to its owner in the copy the moment it exists, so the guard frees it on failure
    auto dup_into = [&](void **slot, const void *p) {
        if (!p) return;
        size_t sz = 0;
        HIP_OK(hipMemPtrGetInfo(const_cast<void *>(p), &sz));
        {
            const hipError_t e = dev_malloc(slot, std::max<size_t>(sz, 1));
            if (e != hipSuccess) { // (a copy needs the operator's memory a second time: say so instead of a bare HIP error)
                size_t free_b = 0, total_b = 0;
                (void)hipMemGetInfo(&free_b, &total_b);
                (void)hipGetLastError();
                throw Error(strprintf("out of device memory while copying an H-matrix: %.2f GB requested for one of its buffers, %.2f GB free of %.2f GB (%s) -- a deep copy "
                                      "needs the operator's memory a second time", sz / 1e9, free_b / 1e9, total_b / 1e9, hipGetErrorString(e)));
            }
        }
        if (sz) HIP_OK(hipMemcpy(*slot, p, sz, hipMemcpyDeviceToDevice));
        map.push_back({(const char *)p, (const char *)p + sz, (char *)*slot});
    };
    for (size_t b = 0; b < S->batches.size(); b++) {
        dup_into(&D->batches[b].panelB, S->batches[b].panelB);
        dup_into(&D->batches[b].panelA, S->batches[b].panelA);
        dup_into((void **)&D->batches[b].cidxB, S->batches[b].cidxB);
        dup_into((void **)&D->batches[b].oidxA, S->batches[b].oidxA);
        dup_into((void **)&D->batches[b].zidxB, S->batches[b].zidxB);
        dup_into((void **)&D->batches[b].tidxA, S->batches[b].tidxA);
    }
    dup_into((void **)&D->zd_ptr, S->zd_ptr);
    dup_into((void **)&D->zd_woff, S->zd_woff);
    dup_into((void **)&D->zd_rows, S->zd_rows);
    dup_into(&D->ycl, S->ycl);
    dup_into(&D->xt, S->xt);
    dup_into(&D->W, S->W);
    dup_into((void **)&D->perm_s, S->perm_s);
    dup_into((void **)&D->perm_t, S->perm_t);
    dup_into((void **)&D->iota, S->iota);
    dup_into((void **)&D->ones_idx, S->ones_idx);
    dup_into(&D->x_tmp, S->x_tmp);
    dup_into(&D->y_tmp, S->y_tmp);
    dup_into((void **)&D->tcoord, S->tcoord);
    dup_into((void **)&D->scoord, S->scoord);
    dup_into(&D->ypart, S->ypart);
    auto reloc = [&](const void *p) -> const void * {
        if (!p) return nullptr;
        for (auto &r : map) if ((const char *)p >= r.old_lo && (const char *)p < r.old_hi) return r.neu + ((const char *)p - r.old_lo);
        // (an empty segment may point one past the end of its buffer: never dereferenced, keep it at the same place of the copy)
        for (auto &r : map) if ((const char *)p == r.old_hi) return r.neu + (r.old_hi - r.old_lo);
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
    segs.resize((size_t)nseg_used);
    for (auto &s : segs) { s.panel = reloc(s.panel); s.cidx = (const int *)reloc(s.cidx); s.zidx = (const int *)reloc(s.zidx); s.oidx = (const int *)reloc(s.oidx); }
    D->segs = upload(segs);
    D->n_segs = (long long)segs.size();
    // (the 16-wide workspace of the matrix-core sweep is created again by the copy's first such product)
    dst.dev = D;
    guard.d = nullptr;
}

// panels of one leaf, copied back to the host (introspection / parity tests)
void device_leaf_panels(const HMatrix &H, int64_t leaf, void *A, void *Bout) {
    const DeviceHMatrix *D = H.dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    HM_CHECK(leaf >= 0 && leaf < (int64_t)H.blocks().size(), "leaf index out of range");
    HIP_OK(hipSetDevice(D->device));
    const BlockRec &b = H.blocks()[leaf];
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
#include "device_expand.inc"

