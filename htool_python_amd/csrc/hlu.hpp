// hlu.hpp -- hierarchical LU of an H-matrix: plan (host, hlu_symbolic.cpp) and executor (device, hlu_device.hip).
//
// Replaces htool::lu_factorization / lu_solve / cholesky_factorization / cholesky_solve as bound at
// src/htool/hmatrix/hmatrix.hpp:58-94 (lib/htool itself is not vendored: the algorithm is the textbook
// block-recursive H-LU of Hackbusch / Bebendorf on the strong-admissibility block tree).
//
// (symmetric positive definite operators: the same with A = L L^T on the lower triangle only -- Params::symmetric, half the work:
//   CHOL(t): CHOL(t_i); A(t_j,t_i) <- A(t_j,t_i) L(t_i,t_i)^-T (j > i); A(t_j,t_k) <- A(t_j,t_k) - A(t_j,t_i) A(t_k,t_i)^T (j >= k > i) )
//   LU(t):  for the children t_i of t in order:  LU(t_i);
//                                                A(t_i,t_j) <- L(t_i,t_i)^-1 A(t_i,t_j),  A(t_j,t_i) <- A(t_j,t_i) U(t_i,t_i)^-1   (j > i)
//                                                A(t_j,t_k) <- A(t_j,t_k) - A(t_j,t_i) A(t_i,t_k)                                (j, k > i)
//
// The block structure is kept (same leaves as the operator); a low-rank leaf keeps the form U V^T with a column
// capacity `cap`, updates are APPENDED as further columns and the leaf is re-truncated (Gram matrices + Jacobi, on
// chip) when the room is used up or before the leaf is read.  The recursion is run ONCE on the host on the block
// structure alone (no numbers): it emits leaf-level tasks of five kinds, each with a LEVEL = 1 + the level of the
// last task that wrote what it reads (per cluster-leaf cell of every thin buffer).  The device runs a level as one
// batched launch per kind (one workgroup per task, or per run of updates of one target leaf: fixed order, no
// atomics, so the factorisation is bitwise reproducible).  lu_solve is a second, much shorter task list (two
// triangular sweeps over the leaves) kept on the device and replayed for every right-hand side.
#pragma once
#include <cstdint>
#include <vector>

#include "cluster.hpp"

namespace hm {
namespace hlu {

// ---- plain records shared by the host plan, the device executor and the CPU checker under oracle/ ----
enum TaskType : int32_t { T_FILL = 0, T_APPLY_DENSE = 1, T_APPLY_LR = 2, T_ADDLR = 3, T_FINAL = 4, T_DDPROD = 5, T_GETRF = 6, T_REDUCE = 7, T_NTYPES = 8 };
// T_REDUCE (solve programs): Y[rows of one cluster leaf] -= the sum, in the order of the list, of kconst private contributions (entries a ... of the program's
// aux list: a space-tagged reference and a leading dimension each).  A triangular sweep subtracts, at every node of the cluster tree, the product of an
// off-diagonal block with the part of the solution that is known: the leaves of that block overlap in their rows, and subtracting them one after the
// other is a chain of dependent launches (a dozen per node at the upper levels of the tree).  Instead every leaf writes its product into a slot of its own -- all
// of them side by side in one launch -- and one REDUCE task per cluster leaf sums them in a fixed order: two launches per node, bitwise reproducible.
constexpr int SOLVE_SLOT_COLUMNS = 8; // columns of right-hand sides a slot holds: a solve with more runs in chunks
enum TaskFlags : int32_t {
    F_TRANS = 1,     // APPLY_DENSE: M^T;  APPLY_LR: the leaf transposed (roles of U and V exchanged -- already folded into a / b)
    F_INPLACE = 2,   // APPLY_DENSE: Y is X (square M), Y = M X
    F_ACCUM = 4,     // APPLY_*: Y += alpha M X (otherwise Y = alpha M X)
    F_SUB = 8,       // alpha = -1
    F_XT = 16,       // x is a transposed view: element (i, c) at x + i * x_ld + c  (otherwise x + i + c * x_ld)
    F_YT = 32,       // the same for y
    F_IDENT = 128,   // FILL: the identity (the entry (i, i) of column i is one) instead of zeros
    F_SYM = 64,      // GETRF: the leaf is symmetric positive definite -- no pivoting, and the inverse factors are those of its CHOLESKY factor (L_c^-1 and its transpose)
};
// where an element offset points (top bits of a 64-bit reference)
enum Space : int { SP_FACTOR = 0, SP_DIAG = 1, SP_SCRATCH = 2, SP_RHS = 3 };
constexpr int SPACE_SHIFT = 60;
inline int64_t make_ref(int space, int64_t off) { return ((int64_t)space << SPACE_SHIFT) | off; }

struct Task { // 96 bytes
    int32_t type, flags, level;
    int32_t leaf;   // APPLY_LR: the leaf applied; ADDLR / FINAL: the target; GETRF: the diagonal leaf; DDPROD: the target (tolerance)
    int32_t kref;   // number of columns: >= 0 rank slot, -1 kconst, -2 the run-time number of right-hand sides
    int32_t kconst;
    int32_t m, n;   // APPLY: rows of Y, rows of X;  ADDLR: rows / columns of the updated sub-block;  DDPROD: rows / columns of the product
    int32_t r0, c0; // ADDLR: origin of the sub-block inside the target leaf;  DDPROD: r0 = inner dimension (F_TRANS: b is n x q, the product is a b^T)
    int32_t a_ld, b_ld, x_ld, y_ld; // (x_ld / y_ld of SP_RHS references: the run-time leading dimension)
    int64_t a, b;   // APPLY_DENSE: a = M;  APPLY_LR: a = rows of the output-side factor, b = rows of the input-side factor;  DDPROD: a (m x q), b (q x n)
    int64_t x, y;   // APPLY: input / output thin blocks;  ADDLR: X (m x k) and Z (n x k), update = -+ X Z^T;  DDPROD: outputs X' (ld m), Z' (ld n)
    int64_t w;      // DDPROD: m x n work block
};
static_assert(sizeof(Task) == 96, "task record layout");

struct Leaf { // 48 bytes
    int32_t t_off, m, s_off, n;
    int32_t kind; // 0 dense, 1 low rank
    int32_t cap;  // low rank: columns of room in U and V
    int64_t u, v; // space-tagged element offsets (the factor arena for the operator's leaves): dense D (m x n, ld m) at u;  low rank U (m x cap, ld m) at u, V (n x cap, ld n) at v
    int32_t diag; // dense (t, t): its record in the diagonal table, else -1
    int32_t rank0;
};
static_assert(sizeof(Leaf) == 48, "leaf record layout");

struct Diag { int32_t leaf, m; int64_t linv, uinv; }; // element offsets in the diagonal arena: (P L)^-1 and U^-1, m x m each

// A diagonal block of at most Params::super_rows rows whose inverse factors are formed explicitly after the factorisation (m x m each, in the
// diagonal arena): a triangular sweep is a SEQUENCE of dependent steps, one per cluster leaf and per tree node (~1600 per sweep at 62 500
// unknowns, ~30 us each); with the sweep inside such a block replaced by one dense product the sequence is 8-16 times shorter.
struct Super { int32_t node, m; int64_t linv, uinv; };

struct Bucket { int32_t type, level; int64_t begin, end; int64_t seg_begin, seg_end; };

struct Program {
    std::vector<Task> tasks;       // sorted by (level, type, [target leaf,] emission order)
    std::vector<Bucket> buckets;   // one batched launch each
    std::vector<int64_t> aux;      // REDUCE tasks: (reference, leading dimension) of every contribution
    std::vector<int64_t> seg;      // ADDLR / FINAL buckets: first task of every run with one target (bucket.seg_begin .. seg_end, + one end marker per bucket)
    int64_t scratch_elems = 0;
    int n_levels = 0;
};

struct Params {
    double eps = 1e-3;   // relative truncation tolerance of the low-rank arithmetic
    int cap_min = 64, cap_max = 64, cap_extra = 8; // capacity of a leaf: clamp(ceil(cap_factor rank) + cap_extra, cap_min, cap_max); by default every leaf gets the 64 columns the
                                                   // truncation kernel handles on chip: the room left above its rank is what it can take in before it is truncated again
    double cap_factor = 2.5;                       // (ranks grow with the tolerance: the caller scales it by log(eps) / log(eps of the operator))
    int64_t window_scratch_elems = (int64_t)1 << 29; // a window of the task stream may hold this much scratch (4 GB of doubles)
    int64_t window_tasks = (int64_t)1 << 23;
    int super_rows = 1024;  // solves: diagonal blocks of at most this many rows get explicit inverse factors (see Super; 0: none)
    bool solve_slots = true; // solves: private slots + REDUCE tasks (see T_REDUCE); false: the leaves of a block subtract one after the other
    bool symmetric = false; // the operator is symmetric positive definite and only its LOWER triangle (diagonal leaves included) is given: H-Cholesky, A = L L^T
    int split_min = 8, split_part = 4, split_max_parts = 24; // a run of more than split_min updates of one low-rank leaf in one launch is dealt out to
                                                              // up to split_max_parts workgroups (>= split_part updates each), each with a stage block of its own
};

struct Plan {
    const ClusterTree *tree = nullptr;
    Params params;
    int root = 0;  // cluster node the operator is built on (the root, or a partition: the diagonal block of a rank)
    int n = 0, pos0 = 0; // its size and first position: row i of a right-hand side is cluster position pos0 + i
    std::vector<Leaf> leaves;      // the leaves of the operator first (n_real_leaves), then the STAGE blocks of split update runs: low-rank
                                   // blocks in the scratch space (u / v are space-tagged references) that a part of a run accumulates into
    int64_t n_real_leaves = 0;
    std::vector<Diag> diags;
    int64_t factor_elems = 0, diag_elems = 0, scratch_elems = 0; // arena sizes (elements)
    int64_t n_slots = 0;                                          // rank slots: leaves first, then the outputs of DDPROD tasks
    std::vector<Program> factor;                                  // the factorisation, window after window
    Program solve_n, solve_t;                                     // x <- A^-1 x and x <- A^-T x on an SP_RHS block
    std::vector<Super> supers;                                    // diagonal blocks with explicit inverse factors ...
    Program invert;                                               // ... and the program that forms them (run once, after the factorisation)
    int64_t counts[T_NTYPES] = {0, 0, 0, 0, 0, 0, 0, 0};
    double plan_seconds = 0;
};

// input: the leaves of a square H-matrix on one cluster tree (both triangles), rank < 0: dense
struct LeafIn { int t_node, s_node, rank; };
Plan *make_plan(const ClusterTree &T, const std::vector<LeafIn> &leaves, const Params &P, int root = 0);

} // namespace hlu
} // namespace hm
