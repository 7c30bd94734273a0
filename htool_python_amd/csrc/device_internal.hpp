// device_internal.hpp -- structures shared by the .hip translation units (never seen by g++ files)
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "hmatrix.hpp"

namespace hm {

// A failing HIP call also leaves its code behind as the runtime's sticky "last error": it is cleared here, otherwise the
// next HIP_OK(hipGetLastError()) after an unrelated kernel launch would report it again (cascading failures after one
// out-of-memory condition).
#define HIP_OK(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            (void)hipGetLastError();                                                                         \
            throw hm::Error(hm::strprintf("HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #call)); \
        }                                                                                                    \
    } while (0)

// one column segment of a tile panel (device view); T elements are double (real) or double2 (complex)
struct GSeg {
    const void *panel;     // first element
    const int *cidx;       // coefficient index (into W) of every column
    int ncols;
    int ld_full;           // leading dimension of full row chunks
    int ld_last;           // leading dimension of the last (or only) row chunk
    int nrows_t = 0;       // rows of the panel (tall panels: a tile may hold several, one per batch)
    long long chunk_stride; // elements between consecutive row chunks
    const int *oidx = nullptr;      // tall panels: output index of every row (null: the tile's omap / out_begin)
    const int *zidx = nullptr;      // one-triangle storage: per column, where its transposed dot product goes in W (-1: nowhere);
                            // for the transposed use of a tall panel: per row, the index in W of its coefficient
    int flush_from = 0;     // phase A, grouped source tiles: rows >= this are written out after this tile's panel (the others go on accumulating)
    int pad_ = 0;
};

struct GTile {
    long long seg_begin;
    int nseg;
    int nrows;
    const int *omap;       // output index of row i is omap[i] when non-null ...
    long long out_begin;   // ... and out_begin + i otherwise
    int xoff = 0;          // cluster position of the tile's first row (one-triangle storage: x of the tile = W[xoff + i])
    int pad_ = 0;
};

struct DevBatch {
    void *panelB = nullptr, *panelA = nullptr;
    int *cidxB = nullptr, *oidxA = nullptr;
    int *zidxB = nullptr, *tidxA = nullptr; // one-triangle storage only (same shapes as cidxB / oidxA)
    size_t bytes = 0;
};

// what table assembly / introspection needs to remember about a packed batch (host copies)
struct BatchTables {
    std::vector<int> b_ncols, a_nrows, a_flush;
    std::vector<int64_t> b_pbase, b_cbase, a_pbase, a_obase;
    std::vector<BatchLayout::Reduce> reduces;
    std::vector<BatchLayout::Reduce> z_reduces;
    std::vector<int> zd_tile;
    std::vector<int64_t> zd_woff;
    int64_t r_end = 0;   // end of the part of the R workspace this batch's layout refers to
};

struct DevBlock;
// the leaf records of a batch packed by the device-resident build, kept on the device until the host asks for them
struct LeafTable {
    DevBlock *blocks = nullptr;
    int2 *nodes = nullptr; // (target node, source node) of every leaf
    size_t n = 0;
    int batch = -1;
};

// the launches of one product, captured as a hipGraph the second time the same product (same buffers, same stream) is asked
// for and replayed from then on: a Krylov loop or a distributed step on fixed buffers then costs one graph launch
struct ProductGraph {
    const void *x = nullptr;
    void *y = nullptr;
    long long x_stride = 0, y_stride = 0;
    int mu = 0, numbering = -1;
    hipStream_t stream = nullptr;
    hipGraphExec_t exec = nullptr;
    long long replays = 0;
};

struct DeviceHMatrix {
    ProductGraph graph;
    std::recursive_mutex mu; // serialises the products of this handle issued from several host threads (shared workspace)
    int device = 0;
    std::vector<BatchTables> tabs;
    bool is_complex = false;
    size_t esize = 8;
    std::vector<DevBatch> batches;
    std::vector<LeafTable> leaf_tables;
    GSeg *segs = nullptr;
    long long n_segs = 0;
    GTile *tilesB_user = nullptr, *tilesB_cluster = nullptr, *tilesA = nullptr, *tilesA2 = nullptr;
    int nB = 0, nA = 0, nA2 = 0;
    // small operators: every row tile is cut in splitB column slices (more, smaller workgroups); the slices
    // write partial sums to ypart[slice][row] and reduce_y_kernel adds them in slice order
    int splitB = 1, nB_split = 0;
    // one-triangle storage of a symmetric operator: extra tables and a cluster-numbered accumulator for y
    bool phase_timing = false;      // record the per-phase events of every product (htool_hmatrix_set_phase_timing)
    bool one_triangle = false;
    bool conj_transposed = false;   // 'H': the second use of a stored leaf is its conjugate transpose
    GTile *tilesAT = nullptr, *tilesZ = nullptr; // transposed use of the tall panels; sums of the transposed partials
    int nAT = 0, nZ = 0;
    int *zd_ptr = nullptr;          // per row tile: range of zd_woff entries to add
    long long *zd_woff = nullptr;
    int *zd_rows = nullptr;         // per row tile: offset and size (2 ints)
    int n_zd_tiles = 0;
    void *ycl = nullptr;
    long long ycl_stride = 0;       // elements between the accumulators of consecutive right-hand sides
    // transposed product of an operator that stores both triangles (tables made on first use, HMatrix::transposable): the tables
    // above in their "tmode" form, plus x gathered into the cluster numbering of the rows
    bool transposable = false;
    void *xt = nullptr;
    long long xt_stride = 0;
    int cntB[4] = {0, 0, 0, 0}, cntBs[4] = {0, 0, 0, 0}; // tiles per class of the wide kernel (F = 1, 2, 4, 8 columns per wave instruction)
    GTile *tilesB_split = nullptr;
    void *ypart = nullptr;
    long long ypart_stride = 0;
    int *perm_s = nullptr, *perm_t = nullptr, *iota = nullptr, *ones_idx = nullptr;
    // 16-wide sweeps on the matrix cores (product_mfma.inc): coefficient workspace W16[index][16] and the list of partial
    // sums to reduce between phase A and phase B; both created by the first product that needs them
    void *W16 = nullptr;
    void *red16 = nullptr;
    int n_red16 = 0;
    // ... and of one-triangle storage / the transposed product: the sums of the transposed partials, y accumulated in the
    // W16 layout ([position][16]) and, for the transposed product, x gathered by row position
    void *redz16 = nullptr;
    int n_redz16 = 0;
    void *ycl16 = nullptr, *xt16 = nullptr;
    int *fin_tile_of = nullptr, *fin_iperm = nullptr; // finishing pass as a gather: tile of every position, inverse of the output permutation
    int fin_npos = 0;
    void *W = nullptr;
    long long W_elems = 0;   // elements of ONE coefficient workspace; W holds rhs_cap of them back to back
    int rhs_cap = 0;
    void *x_tmp = nullptr, *y_tmp = nullptr;
    int n_source = 0, n_target = 0, row_off = 0, row_size = 0;
    hipStream_t stream = nullptr;
    // ring of HIP events bracketing the four launches of the most recent products (kernel durations
    // measured on the stream the kernels run on; read back by htool_hmatrix_phase_times)
    static constexpr int RING = 32;
    hipEvent_t pev[RING][5] = {};
    long long nprod = 0;
    size_t table_bytes = 0;
    // cluster-ordered coordinates (SoA x|y|z) of a native generator, kept for dense evaluation / ACA
    double *tcoord = nullptr, *scoord = nullptr;
};

// what the ACA kernels need of an admissible leaf (40 bytes: the queue of a 1 M-point build is 31 MB to upload instead of 86)
struct AcaBlock {
    long long tmp_u, tmp_v; // arena offsets (elements) of the U / V workspace of the leaf
    int t_off, m, s_off, n;
    int cap, pad_;
};

// per-block device descriptor used by the pack / unpack kernels
struct DevBlock {
    long long tmp_u, tmp_v; // arena offsets (elements)
    long long tpos;
    long long v_obase;
    int t_off, m, s_off, n;
    int rank;               // -1 dense
    int cap;
    int ucol, vcol;
    int v_ostride, v_tile0;
    int status, pad_;
    long long z_obase, zfin; // one-triangle storage (see BlockRec)
    int z_ostride, z_tile0;
};

} // namespace hm
