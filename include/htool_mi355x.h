/* htool_mi355x.h -- C ABI of libhtool_mi355x.so, the MI355X (gfx950) H-matrix engine.
 *
 * This is the drop-in boundary for the H-matrix build + matvec hot path of
 * htool-ddm/htool_python.  Every entry point replaces a call the reference's pybind11
 * translation unit (src/htool/main.cpp) makes into the un-vendored C++ core lib/htool; the
 * replaced call site is cited (path:line relative to the reference repository) next to each
 * declaration.  Plain pointers and sizes only; no torch / pybind11 / STL types cross this line.
 *
 * Conventions (all visible at the reference's binding layer):
 *   - coordinates are point-major, dim doubles per point (cluster_tree_builder.hpp:19-23)
 *   - permutation[i] = user index of the point at cluster position i
 *   - generators are called with USER indices and write column-major blocks
 *     (hmatrix/interfaces/virtual_generator.hpp:16-25)
 *   - products take and return USER-numbered vectors (hmatrix/hmatrix.hpp:101-117)
 *   - complex data is interleaved (re, im) double pairs; "void*" data pointers are double* or
 *     double(*)[2] according to the object's is_complex flag
 *   - every function returning int returns 0 on success, non-zero on error; the message is
 *     available from htool_last_error() (thread-local).  The shim rethrows it as RuntimeError.
 *   - the compute path is HIP only: if no gfx950 device is usable, build/product calls FAIL
 *     (there is no CPU fallback inside the library).
 *   - products on ONE handle are not re-entrant (they share the handle's coefficient workspace and stream
 *     ordering is the caller's); different handles are independent.  The reference gives no stronger guarantee.
 */
#ifndef HTOOL_MI355X_H
#define HTOOL_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct htool_cluster htool_cluster;         /* a node of a cluster tree; the root owns the tree */
typedef struct htool_generator htool_generator;     /* entry generator A(i,j) */
typedef struct htool_hmatrix htool_hmatrix;         /* flattened H-matrix resident in HBM */
typedef struct htool_distributed htool_distributed; /* row-partitioned operator (one rank per GPU) */

/* ---- errors, device, logging -------------------------------------------------------------- */
const char *htool_last_error(void);
int htool_device_count(void);          /* number of usable HIP devices (0 on a GPU-less box) */
int htool_set_device(int device);      /* select the HIP device for objects created afterwards; the first call for a device also warms
                                           the library up on it: code objects of all kernels loaded, build streams created (HTOOL_WARM_UP=0:
                                           not done, the first cluster tree / build of the process then pays for it) */
double htool_last_warm_up_seconds(void); /* what the warm-up of the most recent htool_set_device took (0: already warm) */
const char *htool_device_name(void);   /* e.g. "gfx950:..." or "" */
void htool_set_num_threads(int n);     /* OpenMP threads of the host-side tree construction (0: leave as is) */
/* Builds and recompressions keep their large temporary device buffers (the ACA arena ...) in a process-wide cache for
 * the next call instead of freeing them (a large hipFree makes the next allocation wait for the driver to scrub the
 * memory).  This gives the cache back to the driver; returns the number of bytes released.  The library does it by
 * itself when one of its allocations fails. */
int64_t htool_release_workspace(void);

/* replaces PythonLoggerWriter / htool::Logger (misc/logger.hpp:10-37, main.cpp:42).
 * levels: 0 CRITICAL, 1 ERROR, 2 WARNING, 3 DEBUG, 4 INFO (order of logger.hpp:17-32) */
typedef void (*htool_log_sink)(int level, const char *message);
void htool_set_log_sink(htool_log_sink sink);
void htool_test_logger(void); /* misc/testing.hpp:5-11 */

/* ---- cluster tree (clustering/cluster_tree_builder.hpp:19-67, cluster_node.hpp:18-26) ------- */
enum { HTOOL_PCA_REGULAR = 0, HTOOL_PCA_GEOMETRIC = 1, HTOOL_BBOX_REGULAR = 2, HTOOL_BBOX_GEOMETRIC = 3 }; /* main.cpp:54-57 */

/* partition: NULL | n labels (partition_is_local=0, cluster_tree_builder.hpp:32-39)
 *                 | size_of_partition (offset,size) pairs (partition_is_local=1, :49-56) */
int htool_cluster_create(const double *coordinates, int n_points, int dim, const double *radii, const double *weights,
                         int number_of_children, int size_of_partition, const int *partition, int partition_is_local,
                         int maximal_leaf_size, int strategy, htool_cluster **out);
void htool_cluster_destroy(htool_cluster *root);
int htool_cluster_size(const htool_cluster *c);                                   /* cluster_node.hpp:18 */
int htool_cluster_offset(const htool_cluster *c);                                 /* :19 */
int htool_cluster_maximal_leaf_size(const htool_cluster *c);                      /* :20 */
const int *htool_cluster_permutation(const htool_cluster *c, int *n);             /* :21-25, borrowed, whole tree */
const htool_cluster *htool_cluster_on_partition(const htool_cluster *c, int p);   /* :26, borrowed */
int htool_cluster_dimension(const htool_cluster *c);
/* introspection used by plotting and tests: node table of the whole tree.
 * ints per node: offset,size,depth,parent,first_child,n_children,partition; doubles: cx,cy,cz,radius */
int htool_cluster_node_count(const htool_cluster *c);
int htool_cluster_node_id(const htool_cluster *c);
void htool_cluster_nodes(const htool_cluster *c, int *ints7, double *doubles4);
int htool_cluster_number_of_children(const htool_cluster *c);
/* the inverse of htool_cluster_permutation + htool_cluster_nodes: a cluster tree from its tables (read_cluster_from,
 * clustering/utility.hpp:10 -- the reference reads the two CSV files lib/htool writes; the host language parses the files
 * and hands the tables over).  The tables are validated (permutation, children tiling their parents). */
int htool_cluster_create_from_tables(int n_points, int dim, int maximal_leaf_size, int number_of_children, const int *permutation, int n_nodes,
                                     const int *ints7, const double *doubles4, htool_cluster **out);

/* ---- generators (hmatrix/interfaces/virtual_generator.hpp:16-25) ----------------------------- */
/* callback flavour: out is column-major M x N, rows/cols in user numbering.  Only ever invoked on
 * the thread that entered the library (ctx is typically a Python object needing the GIL). */
typedef void (*htool_copy_submatrix_fn)(void *ctx, int M, int N, const int *rows, const int *cols, void *out);
int htool_generator_create_callback(int is_complex, htool_copy_submatrix_fn fn, void *ctx, htool_generator **out);

/* native flavour: evaluated on the device.  kinds: */
enum {
    HTOOL_KERNEL_INV_DELTA = 0, /* 1/(param + |x-y|)            (example/define_generators.py:14-17, param=0.1) */
    HTOOL_KERNEL_LAPLACE   = 1, /* 1/(4 pi |x-y|), 0 at x==y     */
    HTOOL_KERNEL_HELMHOLTZ = 2  /* exp(i param |x-y|)/(4 pi |x-y|), 0 at x==y; complex */
};
int htool_generator_create_native(int kind, int dim, const double *target_points, int n_target, const double *source_points,
                                  int n_source, double param, htool_generator **out);
int htool_generator_is_complex(const htool_generator *g);
void htool_generator_destroy(htool_generator *g);

/* ---- builder hooks ---------------------------------------------------------------------------- */
/* custom compressor (hmatrix/interfaces/virtual_low_rank_generator.hpp:25-45): return 1 and set
 * U (M x rank, column-major), V (rank x N, column-major) on success, 0 when not worthwhile.
 * The library copies U and V before the next call, or -- htool_build_params.compress_borrows = 1 -- borrows them until
 * the build returns. */
typedef int (*htool_compress_fn)(void *ctx, int M, int N, const int *rows, const int *cols, double epsilon,
                                 const void **U, const void **V, int *rank);
/* batched dense fill (hmatrix/interfaces/virtual_dense_blocks_generator.hpp:21-35): block i is
 * M[i] x N[i] column-major at ptrs[i]; its rows are permutation_t[row_offsets[i] ...], same for cols. */
typedef void (*htool_dense_blocks_fn)(void *ctx, int n_blocks, const int *M, const int *N, const int *row_offsets,
                                      const int *col_offsets, void **ptrs);

/* parameters of HMatrixTreeBuilder (hmatrix/hmatrix_tree_builder.hpp:23-43) */
typedef struct htool_build_params {
    double epsilon;
    double eta;
    char symmetry; /* 'N','S','H' */
    char uplo;     /* 'N','L','U' */
    int reqrank;   /* -1: epsilon-driven */
    int minimal_target_depth;
    int minimal_source_depth;
    int block_tree_consistency;
    htool_compress_fn compress;         /* NULL: built-in partial-pivot ACA */
    void *compress_ctx;
    htool_dense_blocks_fn dense_blocks; /* NULL: generator */
    void *dense_blocks_ctx;
    /* copy-or-borrow of the compress hook's factors (virtual_low_rank_generator.hpp:33-42, allow_copy).  0 (default): the
     * library copies U and V before it calls the hook again (the hook may reuse its buffers).  1: the hook guarantees that
     * every U / V it returned stays valid until htool_hmatrix_build returns; the library then reads them only once, when it
     * ships the panels to HBM at the end of the host phase -- no intermediate host copy.  Either way nothing is referenced
     * after the build: the panels live in device memory. */
    int compress_borrows;
    /* 1 (default): a symmetric or Hermitian ('S' / 'H', UPLO 'L'/'U') operator built on one cluster tree without
     * partition restriction keeps the UPLO triangle only, like the reference, and the product applies
     * every stored off-diagonal leaf a second time, (conjugate) transposed, in the same pass over its panels.
     * 0: both triangles are stored and the product is a single plain sweep (faster for many right-hand sides). */
    int store_one_triangle;
    /* Built-in ACA only (an extension; 0 = the reference's stopping rule |u_k| |v_k| <= epsilon |A_k|_F, tested on one term):
     * c > 0 asks for c further pivot steps that pass the test as well before a leaf is accepted; the confirming terms are
     * dropped, so leaves on which the test was right keep exactly the factors of c = 0, and a leaf changes only when a
     * confirming step finds a large term (nearly collinear clouds, where partial pivoting can stop far too early:
     * profiles/r03_fuzz_sheet_case_61_322.txt).  Costs one more pivot step per leaf and unit of c.  0 .. 8. */
    int aca_confirm_steps;
    /* 1: the index tables of the transposed product y = H^T x / H^H x (htool_hmatrix_matvec with trans = 'T' / 'C'; the trans argument
     * of lu_solve and of the local-operator hooks, hmatrix/hmatrix.hpp:64-78) are laid out and written by the build itself: + 0.5 % of
     * panel memory, one more coefficient slot per panel column.  0 (default): they are made by the first transposed product
     * (about 0.5 s at 10^6 points, once).  Ignored for one-triangle storage (which carries them anyway). */
    int transposed_products;
} htool_build_params;
void htool_build_params_default(htool_build_params *p);

/* ---- H-matrix (hmatrix/hmatrix_tree_builder.hpp:36, hmatrix/hmatrix.hpp:31-138) ---------------- */
/* HMatrixTreeBuilder.build(generator, target, source, target_partition_number, partition_number_for_symmetry).
 * The cluster roots must outlive the H-matrix (as in the reference, which stores references). */
int htool_hmatrix_build(const htool_generator *g, const htool_cluster *target_root, const htool_cluster *source_root,
                        const htool_build_params *params, int target_partition_number, int partition_number_for_symmetry,
                        htool_hmatrix **out);
/* Same, restricted to the columns of one partition of the source tree (source_partition_number >= 0): the
 * (partition p x partition q) sub-operator behind DefaultLocalApproximationBuilder and
 * block_diagonal_hmatrix (distributed_operator/utility.hpp:31,34-41).  On a side restricted to a partition the
 * host-pointer products work on that partition's slice in cluster order. */
int htool_hmatrix_build_local(const htool_generator *g, const htool_cluster *target_root, const htool_cluster *source_root,
                              const htool_build_params *params, int target_partition_number, int source_partition_number,
                              htool_hmatrix **out);
void htool_hmatrix_destroy(htool_hmatrix *h);
int htool_hmatrix_clone(const htool_hmatrix *h, htool_hmatrix **out); /* __deepcopy__, hmatrix.hpp:48 */
/* htool::recompression (hmatrix.hpp:96-99): SVD recompression of the low-rank leaves, on the device, in place
 * (panels re-packed; leaf ranks shrink).  epsilon <= 0 uses the builder's epsilon.  Applies for
 * epsilon >= 1e-7 (tighter tolerances are left unchanged, a WARNING is logged).  n_reduced: leaves whose rank decreased. */
int htool_hmatrix_recompress(htool_hmatrix *h, double epsilon, int64_t *n_reduced);
int htool_hmatrix_is_complex(const htool_hmatrix *h);
int htool_hmatrix_nb_rows(const htool_hmatrix *h); /* hmatrix.hpp:31: size of the target cluster it was built on */
int htool_hmatrix_nb_cols(const htool_hmatrix *h);
const htool_cluster *htool_hmatrix_target_cluster(const htool_hmatrix *h); /* hmatrix.hpp:55, borrowed */
const htool_cluster *htool_hmatrix_source_cluster(const htool_hmatrix *h); /* hmatrix.hpp:56, borrowed */

/* y = alpha * op(H) x + beta * y, host pointers, USER numbering; replaces
 * htool::add_hmatrix_vector_product (hmatrix.hpp:113).  trans: 'N' (op(H) = H), 'T' (H^T) or 'C' (H^H; 'T' for a real
 * operator) -- the value the reference passes through from lu_solve / the local-operator hooks (hmatrix.hpp:64-78).  For
 * 'T' / 'C' x has one entry per row of H and y one per column.  The first transposed product of an operator makes its extra
 * index tables (a few per cent of the panels; the panels themselves are used as they are); for one-triangle storage only
 * the transposition that leaves the operator unchanged is accepted ('T' for symmetry 'S', 'C' for 'H'). */
int htool_hmatrix_matvec(const htool_hmatrix *h, char trans, const void *alpha, const void *x, const void *beta, void *y);
/* Y = alpha H X + beta Y, X column-major n_cols x mu; replaces add_hmatrix_matrix_product (hmatrix.hpp:134) */
int htool_hmatrix_matmat(const htool_hmatrix *h, char trans, const void *alpha, const void *X, int mu, const void *beta, void *Y);
/* device-pointer variants for GPU-resident loops (Krylov, distributed bench): y = H x with
 * x (n_cols) and y (rows of this H-matrix) device buffers.  numbering: 0 = user in and out,
 * 1 = cluster in and out (x is the whole permuted source vector, y the local row slice),
 * 2 = user in, cluster (local row slice) out, 3 = cluster in, user out.  stream = hipStream_t. */
int htool_hmatrix_matvec_device(const htool_hmatrix *h, const void *x_dev, void *y_dev, int numbering, void *stream);
/* Y = H X on device buffers: mu right-hand sides, column c of X at X_dev + c*ldx elements (same for Y).
 * All columns are multiplied in sweeps of up to 8 right-hand sides per pass over the panels. */
int htool_hmatrix_matmat_device(const htool_hmatrix *h, const void *X_dev, int64_t ldx, void *Y_dev, int64_t ldy, int mu, int numbering, void *stream);
/* the same with trans as in htool_hmatrix_matvec; for 'T' / 'C' the "in" side of the numbering is the target side (x has
 * one entry per row of this H-matrix: its local row slice when cluster-numbered), the "out" side the source side */
int htool_hmatrix_matmat_device_trans(const htool_hmatrix *h, char trans, const void *X_dev, int64_t ldx, void *Y_dev, int64_t ldy, int mu, int numbering, void *stream);

/* dense expansion, column-major nb_rows x nb_cols (hmatrix.hpp:32-46): written leaf by leaf on the device (every panel read once),
 * downloaded, permuted to the caller's numbering */
int htool_hmatrix_to_dense(const htool_hmatrix *h, void *out, int user_numbering);

/* H-LU / H-Cholesky (hmatrix.hpp:58-94): hierarchical factorisations are NOT part of this engine.  So that code written for the
 * reference still runs -- in particular the one-level DDM preconditioner, which factorises the rank's diagonal block
 * (example/use_ddm_solver.py:48-63) -- these entries factorise a DENSE copy of the operator:
 *   - whole-cluster operators of at most 20000 unknowns: dense(H) downloaded, factorised on the host (partial pivoting);
 *   - larger operators and partition-built blocks (block_diagonal_hmatrix): dense(H) written ON THE DEVICE, leaf by leaf, and
 *     factorised there by the dense solver library (rocSOLVER, loaded at run time); an LU request for a real operator with
 *     symmetry 'S' tries Cholesky first and falls back to pivoting if the matrix is not positive definite;
 *     the limit is the memory of the dense copy (62500 unknowns = 31 GB).  HTOOL_DENSE_FACTOR=device / host forces one path.
 * A WARNING is logged either way.  kind: 1 = LU, 2 = Cholesky; B is n x mu column-major in user numbering (cluster order of
 * the block for partition-built operators), overwritten by the solution. */
int htool_hmatrix_lu_factorization(htool_hmatrix *h);
int htool_hmatrix_cholesky_factorization(htool_hmatrix *h, char uplo);
int htool_hmatrix_factor_solve(const htool_hmatrix *h, int kind, char trans, void *B, int mu);
/* extensions of the device path: LU of (H + shift I); solves on DEVICE right-hand sides in the operator's cluster numbering
 * (column c at B_dev + c * ldb), enqueued on stream -- what a Krylov loop applies as a preconditioner; the dense expansion by
 * itself (column-major, leading dimension ld >= rows, cluster numbering of the rows and columns the operator covers) */
int htool_hmatrix_lu_factorization_shifted(htool_hmatrix *h, double shift);
int htool_hmatrix_factor_solve_device(const htool_hmatrix *h, int kind, char trans, void *B_dev, int64_t ldb, int mu, void *stream);
int htool_hmatrix_to_dense_device(const htool_hmatrix *h, void *out_dev, int64_t ld, void *stream);

/* host-only introspection (needs no GPU): the two work queues of the block cluster tree BEFORE the
 * leaves are filled -- admissible (to be compressed) and inadmissible (dense) -- as 4 ints per entry
 * {t_off, m, s_off, n}.  Call with NULL arrays to get the counts.  (hmatrix_tree_builder.hpp:36) */
int htool_block_tree_queues(const htool_cluster *target_root, const htool_cluster *source_root, const htool_build_params *params,
                            int target_partition_number, int64_t *n_admissible, int64_t *n_dense, int *admissible4, int *dense4);
/* host-only: the row/column tiles the panels of an H-matrix on these clusters would be grouped by
 * (tile_max rows at most; 2 ints per tile {offset, size}); returns the tile count */
int htool_cluster_tiles(const htool_cluster *root, int partition_number, int tile_max, int *out2, int cap);

/* flattened leaf list: 5 ints per leaf {t_off, m, s_off, n, rank}, rank -1 = dense
 * (matplotlib/hmatrix.hpp:13-24,65-67) */
int64_t htool_hmatrix_leaf_count(const htool_hmatrix *h);
void htool_hmatrix_leaves(const htool_hmatrix *h, int *out5);
/* panels of one leaf copied from HBM to host: dense -> A (m x n col-major); low rank -> A = U (m x r
 * col-major), B = V (r x n col-major).  Used by parity tests and plotting; not a hot path. */
int htool_hmatrix_leaf_panels(const htool_hmatrix *h, int64_t leaf, void *A, void *B);

/* the same for n leaves in one call: leaf q's dense block or U (column-major) starts at element offsets2[2q] of
 * out, its V at offsets2[2q+1] stored step by step (rank rows of n_cols entries).  Pass out = NULL to obtain the
 * offsets and the total element count first. */
int htool_hmatrix_leaf_panels_bulk(const htool_hmatrix *h, int64_t n, const int64_t *leaf_ids, int64_t *offsets2, void *out,
                                   int64_t *n_elements);

/* Rebuild an H-matrix from its leaves (checkpoint / resume): leaves5 = n_leaves x (t_off, m, s_off, n,
 * rank) as returned by htool_hmatrix_leaves, offsets2 / data as returned by htool_hmatrix_leaf_panels_bulk for ALL
 * leaves (dense block or U at offsets2[2i], V at offsets2[2i+1], elements of the coefficient type).  The clusters must
 * be the trees the leaves were computed on (same offsets and sizes); params carries epsilon / eta / symmetry / UPLO for
 * the information entries and store_one_triangle = 1 when the leaves hold one triangle of a symmetric operator. */
int htool_hmatrix_build_from_leaves(const htool_cluster *target_root, const htool_cluster *source_root, const htool_build_params *params,
                                    int is_complex, int target_partition_number, int64_t n_leaves, const int *leaves5,
                                    const int64_t *offsets2, const void *data, int64_t n_elements, htool_hmatrix **out);
/* 1 when the handle stores one triangle of a symmetric / Hermitian operator */
int htool_hmatrix_is_one_triangle(const htool_hmatrix *h);

/* get_tree_parameters / get_local_information (hmatrix.hpp:50-52): "key=value\n" lines copied
 * into buf (truncated to cap); returns needed size. which: 0 tree parameters, 1 local information */
int htool_hmatrix_info(const htool_hmatrix *h, int which, char *buf, int cap);
/* numeric statistics for roofline accounting: out[0]=dense elements, out[1]=low-rank
 * elements sum r(m+n), out[2]=n dense leaves, out[3]=n low-rank leaves, out[4]=sum of ranks,
 * out[5]=bytes resident in HBM, out[6]=build seconds*1e6, out[7]=max rank */
void htool_hmatrix_stats(const htool_hmatrix *h, int64_t *out8);
/* Per-phase timing of the products of this handle (HIP events around every launch, recorded on the stream the kernels
 * run on).  Off by default: the five event records cost about 19 us per product, a quarter of the time of a
 * 10 000-point product.  The two queries below report nothing while it is off. */
int htool_hmatrix_set_phase_timing(htool_hmatrix *h, int on);
/* time of the kernels of the last product in microseconds (HIP events), -1 if none */
double htool_hmatrix_last_product_us(const htool_hmatrix *h);
/* average duration in microseconds of the four launches of a product (HIP events recorded on the
 * stream the kernels ran on) over the completed products of the last 32: out4[0] x gather/copy,
 * out4[1] phase A (V panels), out4[2] phase A2 (partial sums), out4[3] phase B (U and dense panels).
 * Returns how many products were averaged.  Call after synchronising the stream. */
int htool_hmatrix_phase_times(const htool_hmatrix *h, double *out4);

/* ---- distributed operator (distributed_operator/utility.hpp:25-32, distributed_operator.hpp:18-65) */
/* Communicator: rank / size plus (a) an RCCL communicator owned or wrapped by the library -- the opaque handle the
 * GPU-resident exchange uses (one process per GPU, RCCL over xGMI; replaces the MPI_Comm the reference extracts from
 * mpi4py, misc/wrapper_mpi.hpp:28-55) -- and / or (b) a host-buffer all-gather callback supplied by the host language
 * (mpi4py stand-in over torch.distributed / gloo), which serves the replicated-vector API on boxes where several ranks
 * share a GPU.  A communicator made by htool_comm_init_rccl / htool_comm_wrap_rccl has both: its allgatherv stages host
 * buffers through device memory and RCCL. */
typedef struct htool_comm {
    int rank, size;
    void *ctx;
    int (*allgatherv)(void *ctx, const void *send, int64_t send_bytes, void *recv, const int64_t *recv_bytes, const int64_t *displs);
    void *rccl; /* opaque; NULL for a host-only communicator */
    /* (c) the exchange step of the GPU-resident product (htool_distributed_matvec_device): all-gather of EQUAL slices of
     * `bytes` bytes on DEVICE buffers, enqueued on `stream` (hipStream_t) -- recv_dev receives size * bytes, rank p's slice at
     * p * bytes.  htool_comm_init_rccl / _wrap_rccl install ncclAllGather here.  NULL on a communicator that has only (b):
     * the library then stages the slices through pinned host memory and allgatherv (device -> host, stream synchronised,
     * allgatherv, host -> device): slower, but the product runs the same code path -- counts, displacements, padded slices,
     * compaction, local product -- with several ranks sharing ONE GPU (tests, rehearsals).  A host language may also install
     * its own (the ctypes tests do, to play rank r of P in a single process). */
    int (*allgather_device)(void *ctx, const void *send_dev, void *recv_dev, int64_t bytes, void *stream);
    /* (d) the exchange step of the TRANSPOSED GPU-resident product (htool_distributed_matmat_device_trans): reduce-scatter of
     * doubles on device buffers, enqueued on `stream` -- send_dev holds size * count doubles, recv_dev receives the SUM over
     * the ranks of everybody's chunk number `rank` (count doubles).  htool_comm_init_rccl / _wrap_rccl install
     * ncclReduceScatter.  NULL: staged through the host and allgatherv by the library, as for (c). */
    int (*reduce_scatter_device)(void *ctx, const void *send_dev, void *recv_dev, int64_t count, void *stream);
} htool_comm;
/* RCCL bootstrap (the NCCL pattern): ONE rank obtains a unique id, the host language broadcasts its 128 bytes to the
 * other ranks by its own means, then every rank calls htool_comm_init_rccl on the device it selected with
 * htool_set_device (collective: returns when all `size` ranks have called it). */
#define HTOOL_RCCL_UNIQUE_ID_BYTES 128
int htool_rccl_get_unique_id(void *id128);
int htool_comm_init_rccl(const void *id128, int rank, int size, htool_comm *out);
/* the same around a communicator the caller already has (ncclComm_t); it is not destroyed by htool_comm_destroy_rccl */
int htool_comm_wrap_rccl(void *nccl_comm, int rank, int size, htool_comm *out);
void htool_comm_destroy_rccl(htool_comm *comm);

/* DefaultApproximationBuilder: builds rows(partition rank) x all columns */
int htool_distributed_create_default(const htool_generator *g, const htool_cluster *target_root, const htool_cluster *source_root,
                                     const htool_build_params *params, const htool_comm *comm, htool_distributed **out);
void htool_distributed_destroy(htool_distributed *d);
htool_hmatrix *htool_distributed_hmatrix(htool_distributed *d);                /* utility.hpp:29, borrowed */
htool_hmatrix *htool_distributed_block_diagonal_hmatrix(htool_distributed *d); /* utility.hpp:31, borrowed */
void htool_distributed_shape(const htool_distributed *d, int *rows, int *cols); /* distributed_operator.hpp:18 */
/* rows owned by rank p, as a range of cluster numbering (the depth-1 partition of the target tree) */
int htool_distributed_partition(const htool_distributed *d, int p, int *offset, int *size);
/* replicated user-numbered x in, replicated y out (distributed_operator.hpp:23-65) */
int htool_distributed_matvec(const htool_distributed *d, const void *x, void *y);
int htool_distributed_matmat(const htool_distributed *d, const void *X, int mu, void *Y);
/* GPU-resident form of the same product (distributed_operator.hpp:33): every rank passes ITS slice of x (the positions of
 * source partition `rank`, cluster numbering, device memory) and receives ITS rows of y (target partition `rank`, cluster
 * numbering).  The library gathers the slices with ONE ncclAllGather (zero-copy for equal slices, padded slices plus one
 * compaction kernel otherwise) and multiplies, all on `stream` (hipStream_t; NULL: the operator's own stream) without any
 * host synchronisation.  Needs a source tree partitioned like the target tree and a communicator with an RCCL handle
 * (htool_comm.allgather_device; without one the slices are staged through the host and allgatherv, see htool_comm).
 * The matmat form takes mu columns (column c at X_local + c * ldx elements) in one exchange. */
int htool_distributed_matvec_device(htool_distributed *d, const void *x_local_dev, void *y_local_dev, void *stream);
int htool_distributed_matmat_device(htool_distributed *d, const void *X_local_dev, int64_t ldx, void *Y_local_dev, int64_t ldy, int mu, void *stream);
/* The TRANSPOSED product with the same distribution (lib/htool's distributed operator has it; the reference's Python surface only
 * ever passes 'N', distributed_operator.hpp:23-65): y = op(A) x with op = transpose ('T') or conjugate transpose ('C'; 'T' for
 * real operators).  Every rank passes ITS slice of x -- the rows it owns: target partition `rank`, cluster numbering -- and
 * receives ITS slice of y (source partition `rank`): the local block applies op to its rows (a full-length vector), the ranks'
 * vectors are summed and dealt out by ONE reduce-scatter (htool_comm.reduce_scatter_device; padded slices unless the source
 * partition is even and mu = 1), all on `stream`. */
int htool_distributed_matmat_device_trans(htool_distributed *d, char trans, const void *X_local_dev, int64_t ldx, void *Y_local_dev, int64_t ldy, int mu, void *stream);
/* which exchange the device product of this operator uses: 0 none (one rank owns everything), 1 the communicator's
 * allgather_device straight into the contiguous vector (equal slices, one column), 2 the same on padded slices followed by
 * the compaction kernel, 3 / 4 = 1 / 2 staged through the host (no allgather_device); -1: no device exchange possible (the
 * source tree carries no partition of the communicator's size).  mu as in the matmat call. */
int htool_distributed_exchange_kind(const htool_distributed *d, int mu);
/* diagnostic entry for the unit test of the compaction kernel of the padded exchange: gathered_dev is [P][mu][pad]
 * elements (what the all-gather of padded slices delivers), x_full_dev receives column c at c * ldx with rank p's slice at
 * displs[p] .. displs[p] + counts[p] (counts / displs: host arrays of P ints). */
int htool_debug_compact_slices(const void *gathered_dev, void *x_full_dev, const int *counts, const int *displs, int P, int pad, int mu, int64_t ldx,
                               int is_complex, void *stream);

/* diagnostic entries for the unit tests of the two primitives the device-resident build stands on (csrc/device_scan.inc; the
 * block tree of hmatrix_tree_builder.hpp:36 is flattened with them).  htool_debug_scan_positions: element i asks for
 * counts[2 i] places in one output stream and counts[2 i + 1] in another; positions[2 i], positions[2 i + 1] receive the
 * exclusive prefix sums IN ELEMENT ORDER, totals2 the two sums.  htool_debug_sort_pairs: stable sort of n (key, value) pairs by
 * the low key_bits bits of the key, in place.  Device arrays. */
int htool_debug_scan_positions(const int *counts_dev, int64_t n, int64_t *positions_dev, int64_t *totals2);
int htool_debug_sort_pairs(uint32_t *keys_dev, uint32_t *values_dev, int64_t n, int key_bits);

/* ---- Krylov helper (solver/solver.hpp:22-65: the reference hands the operator to HPDDM; this package runs its own GMRES on
 * device-resident vectors, htool_python_amd/krylov.py) -------------------------------------------------------------------
 * The tail of an Arnoldi step with classical Gram-Schmidt applied twice, as ONE launch on `stream`.  Basis layout: vector l of
 * right-hand side c at V + c * ld_rhs + l * ld_basis (n entries each); w of right-hand side c at W + c * ldw; coefficient
 * type double or double[2].  With h1 (j + 1 coefficients of the first pass) and t2 = [h2 | w.w] (j + 2 values of the second),
 * both reduced over the ranks: hn^2 = w.w - |h2|^2; w <- (w - sum_l h2[l] v_l) * mask[c] / sqrt(hn^2) in one sweep (V NULL: no
 * subtraction -- the caller has taken the projection out already; scale = 0: no scaling; 0 when hn^2 <= 0; mask NULL = ones);
 * the row [h1 | h2 | w.w | hn^2] (2 j + 4 entries) written to coef + c * (2 j + 4) for the host to fetch.
 * (Round 3 also had the two Gram-Schmidt passes as kernels of this library; the BLAS library's GEMVs were faster -- 0.69 against
 * 0.72 ms per iteration at 62 500 rows, 5.9 against 6.4 ms at 500 000 -- and stayed.) */
int htool_krylov_finish_step(void *W_dev, int64_t ldw, int n, int mu, int is_complex, const void *h1_dev, const void *t2_dev, int j, const double *mask_dev,
                             void *coef_dev, int scale, const void *V_dev, int64_t ld_basis, int64_t ld_rhs, void *stream);

/* ---- hierarchical LU: the plan (src/htool/hmatrix/hmatrix.hpp:58-94: htool::lu_factorization / lu_solve / cholesky_*) ----
 * htool_hmatrix_lu_factorization / _cholesky_factorization factorise hierarchically on the device (csrc/hlu_device.hip): the
 * block-recursive algorithm is run once on the host on the block structure alone and turned into levels of leaf tasks
 * (csrc/hlu_symbolic.cpp, csrc/hlu.hpp).  The entries below expose that PLAN for tests: rects5 = (t_off, m, s_off, n, rank) per leaf
 * of a square operator on the cluster tree of `root`, rank < 0 for a dense leaf; the plan is host data only (no device needed).
 * symmetric != 0: the operator is symmetric positive definite and rects5 holds its LOWER triangle only (diagonal leaves included): the plan
 * is the hierarchical Cholesky factorisation A = L L^T.  super_rows / solve_slots (-1: the defaults, 1024 / on): how the solve programs are shortened -- explicit
 * inverse factors for diagonal blocks of at most super_rows rows, private slots + REDUCE tasks for the leaves of a block step (csrc/hlu.hpp).  htool_hlu_plan_info: see csrc/hlu_capi.cpp for the 23 values; htool_hlu_plan_program: the sorted task records (96 bytes
 * each, struct hm::hlu::Task), launch buckets, target runs and (solves) the contribution lists of the REDUCE tasks of one window of the factorisation (which >= 0) or of the
 * solves (-1: 'N', -2: 'T') or of the program that forms the explicit inverse factors of the small diagonal blocks after the factorisation (-3); htool_hlu_plan_tables: leaf and diagonal-leaf records. */
/* what htool_hmatrix_lu_factorization / _cholesky_factorization left behind: out17[0] = 0 nothing, 1 dense on the host, 2 dense on
 * the device, 3 hierarchical; for 3, out17[1..16] = unknowns, leaves, tasks, launches, windows, bytes of the factors as they stay resident (tight: their rank in columns), bytes of the arena (64 columns of room per leaf) and scratch while factorising,
 * truncations cut at a leaf's capacity, truncations, appended columns, columns out of dense-leaf products, tasks and launches of one
 * solve, sum of rank x (rows + columns) over the low-rank leaves, sum of (rows + columns), tolerance x 1e12; seconds4 = plan, leaves
 * into the factor arena, factorisation, total. */
int htool_hmatrix_factorization_info(const htool_hmatrix *h, int64_t *out17, double *seconds4);
typedef struct htool_hlu_plan htool_hlu_plan;
int htool_hlu_plan_create(const htool_cluster *root, int64_t n_leaves, const int32_t *rects5, double epsilon, int cap_min, int cap_max, double cap_factor,
                          int64_t window_scratch_elems, int64_t window_tasks, int symmetric, int super_rows, int solve_slots, htool_hlu_plan **out);
int htool_hlu_plan_info(const htool_hlu_plan *plan, int64_t *out, int n_out);
int htool_hlu_plan_program(const htool_hlu_plan *plan, int which, const void **tasks, int64_t *n_tasks, const void **buckets, int64_t *n_buckets,
                           const int64_t **seg, int64_t *n_seg, int64_t *scratch_elems, const int64_t **aux, int64_t *n_aux);
int htool_hlu_plan_tables(const htool_hlu_plan *plan, const void **leaves, const void **diags);
void htool_hlu_plan_free(htool_hlu_plan *plan);
/* diagnostic: windows first..last of the plan's factorisation (first >= 0), or one of its solves (first = -1 'N', -2 'T'), or the inverse factors of the small diagonal blocks (first = -3), executed by
 * the DEVICE kernels on host arrays laid out as the plan says (uploaded, run, downloaded) -- the counterpart of the CPU checker
 * oracle/hlu_exec.cpp, which tests feed the same arrays.  counters: 8 values; scratch (may be NULL): the window's scratch space, in and out. */
int htool_hlu_debug_execute(const htool_hlu_plan *plan, int first, int last, double *factor, double *diag, int32_t *rank, double *norm0, double *norm2,
                            int64_t *counters, double *rhs, int64_t ld_rhs, int nrhs, double *scratch);

#ifdef __cplusplus
}
#endif
#endif /* HTOOL_MI355X_H */
