// hmatrix.hpp -- flattened H-matrix: host bookkeeping + handles to the HBM-resident panels.
//
// Data layout in HBM.  All leaves are contiguous index ranges in cluster
// numbering, so the rows [row_off, row_off+row_size) and the source positions [0, n_source) are cut
// into TILES (pieces of cluster-tree leaves, at most TM entries).  The panels of all leaves are
// re-grouped per tile so that every kernel streams contiguous memory:
//   phase B ("wide" tiles, one per row tile): the U rows of every low-rank leaf covering the tile
//     and the rows of every dense leaf of the tile, concatenated column after column; column c has
//     a coefficient index cidx[c] into the coefficient vector W;  y_tile = Panel * W[cidx].
//   phase A ("tall" tiles, one per source tile): the V columns of every low-rank leaf covering the
//     source tile: rows = (leaf, k) pairs, columns = source positions;  t = Panel * x_tile.
//   phase A2 ("tall" tiles): sums the per-source-tile partials of leaves spanning several tiles.
// W = [ x permuted (n_source) | 1.0 | R ] where R holds the t vectors and phase-A partials.
// A panel with nrows x ncols entries is stored as row chunks of TM rows; chunk q is column-major
// [col][ld_q] with ld_q = TM (full chunks) or the remainder rounded up to the vector width.
#pragma once
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "cluster.hpp"
#include "common.hpp"

namespace hm {

// one leaf of the block cluster tree
struct BlockRec {
    int t_node, s_node;
    int t_off, m, s_off, n;
    int rank;  // -1 dense, >= 0 low rank
    int cap;   // rank capacity reserved in the temporary arena (low rank only)
    int batch; // which pack batch holds its panels
    int64_t tmp_u, tmp_v; // element offsets in the temporary arena: U [k][i] (stride m) or D col-major; V [k][j] (stride n)
    int ucol, vcol;       // first column in the phase-B tile panels / first row in the phase-A tile panels
    int64_t tpos;         // index in W of t_b[0]
    int64_t v_obase;      // phase-A output index of (k=0, first source tile)
    int v_ostride;        // phase-A output stride between consecutive source tiles (0: single tile)
    int status;           // device ACA status (0 ok, 1 not compressible, 2 capacity exceeded)
    // one-triangle storage only: where the transposed use of the leaf's phase-B columns is written / read
    int64_t z_obase = 0;  // output index of (column 0, first row tile of the target node) in W
    int z_ostride = 0;    // stride between consecutive row tiles (0: the target node is a single tile)
    int64_t zfin = -1;    // index in W of the reduced transposed coefficients of column 0 (t'_b[0], or the dense leaf's A^T x); -1: none
};

struct TileSet {
    std::vector<int> off, size;         // tiles in increasing offset order
    std::vector<int> node_tile_begin;   // per cluster node: first tile
    std::vector<int> node_tile_end;     // per cluster node: one past last tile
    std::vector<int> leaf_of_tile;      // cluster leaf node containing the tile
    int count() const { return (int)off.size(); }
};

TileSet make_tiles(const ClusterTree &T, int root_node, int tile_max);
// Source tiles are streamed in GROUPS by phase A: a group is the tiles of one cluster node of at most group_positions points
// (or of one leaf).  Every low-rank leaf whose source cluster lies inside a group gets its t = V x completed by ONE
// workgroup; leaves above get one partial sum per group (not per tile).  group[c] = group of tile c, numbered in tile order.
void assign_tile_groups(const ClusterTree &T, int root_node, const TileSet &tiles, int group_positions, std::vector<int> &group);

// host description of what one pack batch needs (computed by layout.cpp, consumed by device.hip)
struct BatchLayout {
    int batch_id = 0;
    std::vector<int64_t> blocks; // indices into HMatrix::blocks that belong to this batch (rank != 0)
    // phase B: per row tile
    std::vector<int> b_ncols;      // columns contributed by this batch
    std::vector<int64_t> b_pbase;  // element offset of the tile's panel in panelB
    std::vector<int64_t> b_cbase;  // offset of the tile's cidx in cidxB
    int64_t panelB_elems = 0, cidxB_elems = 0;
    // phase A: per source tile
    std::vector<int> a_nrows;      // (leaf,k) rows contributed by this batch
    std::vector<int64_t> a_pbase;  // element offset in panelA
    std::vector<int64_t> a_obase;  // offset of the tile's output-index list in oidxA
    std::vector<int> a_flush;      // rows >= this have their sums written out after the tile (0: all of them, the group ends here)
    int64_t panelA_elems = 0, oidxA_elems = 0;
    // phase A2 tiles: (partial panel offset in W, ld, rows, cols, output base in W)
    struct Reduce { int64_t w_panel; int ld; int nrows; int ncols; int64_t out_base; };
    std::vector<Reduce> reduces;
    // one-triangle storage: sums of the per-row-tile transposed partials (run after phase B), and for every
    // stored off-diagonal dense leaf the row tiles of y that receive its transposed contribution
    std::vector<Reduce> z_reduces;
    std::vector<int> zd_tile;        // row tile (of y) receiving a dense leaf's A^T x ...
    std::vector<int64_t> zd_woff;    // ... which starts at this index of W
    // pack work items (block index into `blocks`, tile id): (leaf, row tile) and (low-rank leaf, source tile) pairs in leaf order.
    // u_first / v_first (n_blocks + 1 entries) say where a leaf's items start; the item lists themselves are written here only when
    // host_items is set -- the build expands them on the device from the two prefix arrays (75 MB less to upload at 1 M points)
    bool host_items = true;
    std::vector<int> u_first, v_first;
    std::vector<int> u_item_block, u_item_tile, v_item_block, v_item_tile;
};

struct DeviceHMatrix; // defined in device.hip

struct BuildParams {
    double epsilon = 1e-3, eta = 10;
    char symmetry = 'N', uplo = 'N';
    int reqrank = -1, min_target_depth = 0, min_source_depth = 0, block_tree_consistency = 1;
    int aca_confirm_steps = 0;  // built-in ACA: further steps that must pass the stopping test as well (aca_stop.hpp); 0 = the reference's rule
    int store_one_triangle = 0; // 'S'/'H' on one cluster tree: keep the UPLO triangle only and apply stored leaves transposed too
    int (*compress)(void *, int, int, const int *, const int *, double, const void **, const void **, int *) = nullptr;
    void *compress_ctx = nullptr;
    int compress_borrows = 0; // the hook's U / V stay valid until the build returns: copied once, at the end of the host phase
    void (*dense_blocks)(void *, int, const int *, const int *, const int *, const int *, void **) = nullptr;
    void *dense_blocks_ctx = nullptr;
};

struct Generator {
    bool is_complex = false;
    bool native = false;
    // callback flavour
    void (*fn)(void *, int, int, const int *, const int *, void *) = nullptr;
    void *ctx = nullptr;
    // native flavour
    int kind = 0, dim = 3;
    double param = 0;
    std::vector<double> tpts, spts; // point-major copies, user numbering
    int n_target = 0, n_source = 0;
};

struct HMatrix {
    const ClusterTree *tc = nullptr, *sc = nullptr;
    int t_root = 0;                 // target node the matrix was built on (root or a partition)
    int row_off = 0, row_size = 0;  // rows covered, cluster numbering
    int s_root = 0;                 // source node (root, or a partition for block-diagonal / local operators)
    int col_off = 0, col_size = 0;  // columns covered, cluster numbering
    bool local_numbering = false;   // built as a local block: host products use cluster order on both sides
    bool one_triangle = false;      // symmetric operator stored as its UPLO triangle (off-diagonal leaves are applied twice)
    bool transposable = false;      // the tables of the transposed product exist (made by the first product with trans = 'T' / 'C')
    bool is_complex = false;
    BuildParams params;
    int tile_max = 128;
    TileSet rtiles, ctiles;
    std::vector<int> ctile_group;   // group of every source tile (phase A streams a group per workgroup)
    // The leaf records.  A device-resident build (device_build2.inc) leaves them on the GPU: blocks() fetches them when something
    // on the host asks (introspection, recompression, save, the tables of the transposed product); leaf_count() never does.
    mutable std::vector<BlockRec> blocks_;
    mutable bool blocks_lazy = false;
    mutable std::mutex blocks_mu;
    int64_t n_leaves_lazy = 0;
    std::vector<BlockRec> &blocks() const { if (blocks_lazy) materialise_blocks(); return blocks_; }
    void materialise_blocks() const; // device.hip
    size_t leaf_count() const { return blocks_lazy ? (size_t)n_leaves_lazy : blocks_.size(); }
    int64_t r_elems = 0;            // size of region R of W
    double build_seconds = 0;
    int n_batches = 0;
    DeviceHMatrix *dev = nullptr;   // HBM-resident part (device.hip)
    ~HMatrix();
};

// ---- block tree (blocktree.cpp) ----
void build_block_tree(const ClusterTree &T, const ClusterTree &S, const BuildParams &P, int t_root, int s_root,
                      std::vector<BlockRec> &adm, std::vector<BlockRec> &dns);
// sub-blocks of an admissible block whose compression failed (re-visit ignoring its own admissibility)
void split_failed_block(const ClusterTree &T, const ClusterTree &S, const BuildParams &P, const BlockRec &b,
                        std::vector<BlockRec> &adm, std::vector<BlockRec> &dns);

// ---- layout (layout.cpp) ----
// assigns ucol/vcol/tpos/... of the given blocks (one batch) and fills the BatchLayout.
void compute_batch_layout(HMatrix &H, const std::vector<int64_t> &batch_blocks, int vec_rows, BatchLayout &L);
// the part of it that needs per-node sums only (layout.cpp)
struct NodeLayout {
    std::vector<int> tbase, sbase;      // first phase-B column / phase-A row of a node's own leaves
    std::vector<int64_t> tb, pbse;      // per source node: its t vector and its panel of per-group partial sums in R (-1: none), relative to r_start
    std::vector<int> ldp;
    std::vector<int64_t> zf, zp;        // per target node (one-triangle storage / transposed products): final slots, per-tile partials (-1: none)
    std::vector<int> zld;
    int64_t r_start = 0;                // W index of R[0]
};
void compute_node_layout(HMatrix &H, const std::vector<int> &Klr, const std::vector<int> &Kdn, const std::vector<int> &Ks, int vec_rows, BatchLayout &L, NodeLayout &NL);

// ---- host build for callback generators (build_host.cpp) ----
// runs ACA / the custom compressor / the dense fill on the calling thread; fills a host arena laid out
// like the device temporary arena and sets rank/tmp_u/tmp_v of every block.
template <typename T>
void host_fill_blocks(const Generator &g, HMatrix &H, std::vector<T> &arena);

// ---- device side (device.hip) ----
int device_count();
void device_select(int dev);
std::string device_name();
double device_warm_up(); // loads the code objects, creates the build streams (once per device); returns the seconds it took
// whole build for a callback generator: upload the host arena, pack, assemble the product tables
void device_build_from_host(HMatrix &H, const void *arena, int64_t arena_elems);
// whole build for a native generator: device ACA + dense evaluation + pack
void device_build_native(HMatrix &H, const Generator &g);
// trans: 'N' y = H x, 'T' y = H^T x, 'C' y = H^H x (x then has one entry per ROW of H, y one per column; "in" / "out" of the
// numbering refer to x / y).  The tables of the transposed product are made by the first call that asks for it.
void device_matvec_host(const HMatrix &H, const void *x, void *y, char trans = 'N');                  // user numbering
void device_matvec_device(const HMatrix &H, const void *x_dev, void *y_dev, int numbering, void *stream, char trans = 'N');
void device_matmat_device(const HMatrix &H, const void *X, long long x_stride, void *Y, long long y_stride, int mu, int numbering, void *stream, char trans = 'N');
void device_matmat_host(const HMatrix &H, const void *X, int mu, void *Y, char trans = 'N');
void device_make_transposable(HMatrix &H);
// dense(H), leaf by leaf, into a device matrix (cluster numbering of the rows / columns the operator covers): device_expand.inc
void device_expand_to_dense(HMatrix &H, void *out_dev, long long ld, void *stream);
int64_t device_recompress(HMatrix &H, double eps);
void device_clone(const HMatrix &src, HMatrix &dst);
void device_leaf_panels(const HMatrix &H, int64_t leaf, void *A, void *B);
int64_t device_leaf_panels_bulk(const HMatrix &H, int64_t n, const int64_t *ids, int64_t *offs, void *out);
void device_unpack_leaves(const HMatrix &H, int64_t n, const int64_t *ids, const int64_t *u_offs, const int64_t *v_offs, void *d_arena); // (hierarchical LU: leaves into its own arena)
int64_t device_resident_bytes(const HMatrix &H);
double device_last_product_us(const HMatrix &H);
int device_phase_times(const HMatrix &H, double *out4);
void device_set_phase_timing(const HMatrix &H, bool on);
void device_free(DeviceHMatrix *d);
size_t device_release_workspace(); // frees the cached temporary buffers (build arena ...); returns the bytes freed
size_t device_workspace_bytes();
// unit-test entries of the scan / sort primitives of the device-resident build (device_scan.inc)
void device_debug_positions(const int *counts_dev, long long n, long long *pos_dev, long long totals[2]);
void device_debug_sort_pairs(unsigned *keys_dev, unsigned *vals_dev, long long n, int bits);

} // namespace hm
