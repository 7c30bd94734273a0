// dense_device.hip -- dense expansion of an H-matrix ON THE DEVICE and the dense factorisations behind lu_factorization /
// cholesky_factorization for operators beyond the host fallback.
//
// The reference factorises H-matrices hierarchically (src/htool/hmatrix/hmatrix.hpp:58-94 -> htool::lu_factorization, lu_solve,
// cholesky_*); its one-level DDM preconditioner is exactly that, applied to the rank's diagonal block
// (example/use_ddm_solver.py:48-63: DDMSolverBuilder(operator, block_diagonal_hmatrix).solver.facto_one_level()).  True
// hierarchical LU is not part of this engine ("dense fallback initially, true H-LU later").  What is
// here serves the same calls up to the size where a dense copy fits in HBM (288 GB: 62 500 unknowns -- the per-GPU diagonal
// block of the 500 000-point GMRES configuration on 8 GPUs -- are 31 GB):
//   * dense(H) = H I in sweeps of 16 unit vectors on the fp64 matrix cores, everything device-resident, cluster numbering on
//     both sides (about a millisecond per sweep of a 62 500-row block);
//   * LU / Cholesky by the dense solver library (rocSOLVER getrf / getrs / potrf / potrs -- plain library calls, as the task's
//     rules allow for plain dense linear algebra), resolved at run time with dlopen so that the library has no link-time
//     dependency on it and binds to the copy already in the process (PyTorch bundles one under the same soname).
#include <dlfcn.h>

#include <memory>
#include <mutex>

#include "capi_internal.hpp"
#include "device_internal.hpp"

using namespace hm;

namespace {

typedef void *rb_handle;
typedef int rb_status;
struct rb_z { double x, y; };
enum { RB_OP_N = 111, RB_OP_T = 112, RB_OP_C = 113, RB_UPPER = 121, RB_LOWER = 122 };

struct SolverLib {
    void *blas = nullptr, *solver = nullptr;
    rb_status (*create_handle)(rb_handle *) = nullptr;
    rb_status (*destroy_handle)(rb_handle) = nullptr;
    rb_status (*set_stream)(rb_handle, hipStream_t) = nullptr;
    // the 64-bit interface: a 62 500 x 62 500 matrix has more entries than a 32-bit integer counts
    rb_status (*dgetrf)(rb_handle, int64_t, int64_t, double *, int64_t, int64_t *, int64_t *) = nullptr;
    rb_status (*zgetrf)(rb_handle, int64_t, int64_t, rb_z *, int64_t, int64_t *, int64_t *) = nullptr;
    rb_status (*dgetrs)(rb_handle, int, int64_t, int64_t, double *, int64_t, const int64_t *, double *, int64_t) = nullptr;
    rb_status (*zgetrs)(rb_handle, int, int64_t, int64_t, rb_z *, int64_t, const int64_t *, rb_z *, int64_t) = nullptr;
    rb_status (*dpotrf)(rb_handle, int, int64_t, double *, int64_t, int64_t *) = nullptr;
    rb_status (*dpotrs)(rb_handle, int, int64_t, int64_t, double *, int64_t, double *, int64_t) = nullptr;
    std::string error;
    std::mutex mu;
    bool load() {
        std::lock_guard<std::mutex> lock(mu); // (factorisations of several handles may start on several host threads)
        if (solver) return true;
        error.clear();
        // (dlerror() hands its message out ONCE and clears it: taken right after each failing dlopen, into a string)
        std::string why;
        auto open_first = [&](std::initializer_list<const char *> names) -> void * {
            for (const char *nm : names) {
                if (void *h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL)) return h;
                const char *e = dlerror();
                why += std::string(why.empty() ? "" : "; ") + (e ? e : "dlopen failed without a message");
            }
            return nullptr;
        };
        blas = open_first({"librocblas.so.5", "librocblas.so"});
        void *s = blas ? open_first({"librocsolver.so.0", "librocsolver.so"}) : nullptr;
        if (!blas || !s) { error = "the dense solver library could not be loaded (librocblas / librocsolver): " + why; return false; }
        auto sym = [&](void *lib, const char *name) { void *p = dlsym(lib, name); if (!p) error = std::string("missing symbol ") + name; return p; };
        *(void **)&create_handle = sym(blas, "rocblas_create_handle");
        *(void **)&destroy_handle = sym(blas, "rocblas_destroy_handle");
        *(void **)&set_stream = sym(blas, "rocblas_set_stream");
        *(void **)&dgetrf = sym(s, "rocsolver_dgetrf_64");
        *(void **)&zgetrf = sym(s, "rocsolver_zgetrf_64");
        *(void **)&dgetrs = sym(s, "rocsolver_dgetrs_64");
        *(void **)&zgetrs = sym(s, "rocsolver_zgetrs_64");
        *(void **)&dpotrf = sym(s, "rocsolver_dpotrf_64");
        *(void **)&dpotrs = sym(s, "rocsolver_dpotrs_64");
        if (!error.empty()) return false;
        solver = s;
        return true;
    }
};
SolverLib g_solver;

template <typename T>
__global__ void set_unit_entries_kernel(T *X, long long ldx, int j0, int nb, double value) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nb) {
        double *p = reinterpret_cast<double *>(X + (long long)c * ldx + j0 + c);
        p[0] = value; // (imaginary part stays zero)
    }
}
template <typename T>
__global__ void add_to_diagonal_kernel(T *A, long long ld, int n, double shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reinterpret_cast<double *>(A + (long long)i * ld + i)[0] += shift;
}
template <typename T>
__global__ void permute_rows_kernel(const T *src, T *dst, const int *perm, int n, int mu, int gather) {
    // gather: dst[c][i] = src[c][perm[i]] (user -> cluster); otherwise dst[c][perm[i]] = src[c][i] (cluster -> user)
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)n * mu) return;
    const int c = (int)(e / n), i = (int)(e - (long long)c * n);
    if (gather) dst[e] = src[(long long)c * n + perm[i]];
    else dst[(long long)c * n + perm[i]] = src[e];
}

} // namespace

// dense(H) into out_dev (column-major, leading dimension ld >= rows), cluster numbering of the rows and columns THIS operator
// covers (its row slice / source slice for partition-built operators): leaf by leaf (device_expand.inc); HTOOL_DENSE_EXPANSION=products
// keeps the first form (products with unit vectors, 16 columns per sweep of all panels) for comparison
void device_to_dense_device(const HMatrix &H, void *out_dev, long long ld, void *stream) {
    DeviceHMatrix *D = H.dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    static const bool by_products = getenv("HTOOL_DENSE_EXPANSION") && std::string(getenv("HTOOL_DENSE_EXPANSION")) == "products";
    if (!by_products) {
        device_expand_to_dense(const_cast<HMatrix &>(H), out_dev, ld, stream); // (uses the arena offsets of the block records as scratch, under the handle's lock)
        return;
    }
    HIP_OK(hipSetDevice(D->device));
    const size_t es = H.is_complex ? 16 : 8;
    const int ns = D->n_source, nr = D->row_size;
    HM_CHECK(ld >= nr, "to_dense_device: leading dimension smaller than the number of rows");
    if (ns == 0 || nr == 0) return;
    hipStream_t st = stream ? (hipStream_t)stream : D->stream;
    const int SW = 16; // one sweep of the matrix cores
    void *X = nullptr;
    HIP_OK(hipMalloc(&X, (size_t)ns * SW * es));
    try {
        HIP_OK(hipMemsetAsync(X, 0, (size_t)ns * SW * es, st));
        for (int j0 = 0; j0 < ns; j0 += SW) {
            const int nb = std::min(SW, ns - j0);
            if (H.is_complex) hipLaunchKernelGGL(set_unit_entries_kernel<double2>, dim3(1), dim3(64), 0, st, (double2 *)X, (long long)ns, j0, nb, 1.0);
            else hipLaunchKernelGGL(set_unit_entries_kernel<double>, dim3(1), dim3(64), 0, st, (double *)X, (long long)ns, j0, nb, 1.0);
            device_matmat_device(H, X, ns, (char *)out_dev + (size_t)j0 * (size_t)ld * es, ld, nb, 1, (void *)st);
            if (H.is_complex) hipLaunchKernelGGL(set_unit_entries_kernel<double2>, dim3(1), dim3(64), 0, st, (double2 *)X, (long long)ns, j0, nb, 0.0);
            else hipLaunchKernelGGL(set_unit_entries_kernel<double>, dim3(1), dim3(64), 0, st, (double *)X, (long long)ns, j0, nb, 0.0);
        }
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(st));
    } catch (...) {
        (void)hipFree(X);
        throw;
    }
    (void)hipFree(X);
}

// the same into a HOST matrix (column-major rows x columns of the operator, cluster numbering); false: the dense copy does not fit the
// device next to the operator (the caller falls back to products with unit vectors, 64 columns at a time)
bool device_to_dense_host(const HMatrix &H, void *out) {
    DeviceHMatrix *D = H.dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    HIP_OK(hipSetDevice(D->device));
    const size_t es = H.is_complex ? 16 : 8;
    const size_t bytes = (size_t)H.row_size * (size_t)H.col_size * es;
    if (bytes == 0) return true;
    size_t free_b = 0, total_b = 0;
    HIP_OK(hipMemGetInfo(&free_b, &total_b));
    if (bytes + ((size_t)2 << 30) > free_b) return false;
    void *d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    try {
        device_to_dense_device(H, d, H.row_size, nullptr);
        HIP_OK(hipMemcpy(out, d, bytes, hipMemcpyDeviceToHost));
    } catch (...) {
        (void)hipFree(d);
        throw;
    }
    (void)hipFree(d);
    return true;
}

struct DeviceDenseFactor {
    int kind = 0, n = 0, device = 0; // kind: what the matrix holds (1 LU, 2 Cholesky)
    int asked = 0;                   // what the caller asked for (an LU request for a symmetric operator may hold a Cholesky factor)
    bool is_complex = false;
    char uplo = 'L';
    void *a = nullptr;
    int64_t *ipiv = nullptr, *info = nullptr;
    rb_handle handle = nullptr;
    ~DeviceDenseFactor() {
        (void)hipSetDevice(device);
        if (handle && g_solver.destroy_handle) (void)g_solver.destroy_handle(handle);
            for (void *p : {a, (void *)ipiv, (void *)info}) if (p) (void)hipFree(p);
    }
};
void device_dense_factor_free(DeviceDenseFactor *f) { delete f; }
int device_dense_factor_kind(const DeviceDenseFactor *f) { return f ? f->asked : 0; }

// kind 1 = LU with partial pivoting, 2 = Cholesky (real, uplo triangle of the dense copy); shift is added to the diagonal first
DeviceDenseFactor *device_dense_factor(const HMatrix &H, int kind, char uplo, double shift) {
    DeviceHMatrix *D = H.dev;
    HM_CHECK(D != nullptr, "H-matrix has no device data");
    HM_CHECK(D->row_size == D->n_source, "factorization needs a square H-matrix");
    HM_CHECK(kind == 1 || !H.is_complex, "cholesky_factorization: complex operators are not supported");
    HM_CHECK(g_solver.load(), "factorization on the device: " + g_solver.error);
    const int n = D->row_size;
    const size_t es = H.is_complex ? 16 : 8;
    std::unique_ptr<DeviceDenseFactor> f(new DeviceDenseFactor);
    f->kind = kind; f->asked = kind; f->n = n; f->is_complex = H.is_complex; f->uplo = uplo; f->device = D->device;
    HIP_OK(hipSetDevice(D->device));
    size_t free_b = 0, total_b = 0;
    HIP_OK(hipMemGetInfo(&free_b, &total_b));
    const double need = (double)n * n * es * 1.05 + 256e6; // the dense copy + the solver's workspace
    if (need > (double)free_b) (void)device_release_workspace();
    HIP_OK(hipMemGetInfo(&free_b, &total_b));
    HM_CHECK(need <= (double)free_b, strprintf("factorization: the dense copy of the %d x %d operator needs %.1f GB, %.1f GB are free -- hierarchical LU is not part of this engine", n, n,
                                               need / 1e9, free_b / 1e9));
    HIP_OK(hipMalloc(&f->a, std::max<size_t>((size_t)n * n * es, 1)));
    HIP_OK(hipMalloc((void **)&f->ipiv, sizeof(int64_t) * std::max(n, 1)));
    HIP_OK(hipMalloc((void **)&f->info, sizeof(int64_t)));
    HM_CHECK(g_solver.create_handle(&f->handle) == 0, "rocblas_create_handle failed");
    HM_CHECK(g_solver.set_stream(f->handle, D->stream) == 0, "rocblas_set_stream failed");
    double t_expand = 0;
    auto expand = [&]() {
        const double t0 = wall_seconds();
        device_to_dense_device(H, f->a, n, D->stream);
        if (shift != 0.0 && n > 0) {
            if (H.is_complex) hipLaunchKernelGGL(add_to_diagonal_kernel<double2>, dim3((n + 255) / 256), dim3(256), 0, D->stream, (double2 *)f->a, (long long)n, n, shift);
            else hipLaunchKernelGGL(add_to_diagonal_kernel<double>, dim3((n + 255) / 256), dim3(256), 0, D->stream, (double *)f->a, (long long)n, n, shift);
        }
        t_expand += wall_seconds() - t0;
    };
    auto run = [&](int what) { // the factorisation in place; returns the library's info (0: done)
        rb_status rs;
        if (what == 1) rs = H.is_complex ? g_solver.zgetrf(f->handle, n, n, (rb_z *)f->a, n, f->ipiv, f->info) : g_solver.dgetrf(f->handle, n, n, (double *)f->a, n, f->ipiv, f->info);
        else rs = g_solver.dpotrf(f->handle, f->uplo == 'U' ? RB_UPPER : RB_LOWER, n, (double *)f->a, n, f->info);
        HM_CHECK(rs == 0, strprintf("the dense solver library reported status %d", rs));
        int64_t info = 0;
        HIP_OK(hipMemcpyAsync(&info, f->info, sizeof(int64_t), hipMemcpyDeviceToHost, D->stream));
        HIP_OK(hipStreamSynchronize(D->stream));
        return info;
    };
    const double t_begin = wall_seconds();
    expand();
    log_message(LOG_DEBUG, strprintf("dense copy of the %d x %d operator on the device: %.3f s", n, n, t_expand));
    int64_t info = -1;
    // an LU of a real SYMMETRIC operator ('S': the block of a symmetric operator a DDM preconditioner inverts, use_ddm_solver.py:42):
    // Cholesky costs half the arithmetic; if the matrix turns out not to be positive definite the copy is written again (cheap:
    // leaf by leaf) and factorised with pivoting after all.  HTOOL_DENSE_LU_CHOLESKY=0: always with pivoting.
    static const bool chol_off = getenv("HTOOL_DENSE_LU_CHOLESKY") && std::string(getenv("HTOOL_DENSE_LU_CHOLESKY")) == "0";
    if (kind == 1 && !H.is_complex && H.params.symmetry == 'S' && !chol_off && n > 0) {
        f->uplo = 'L';
        info = run(2);
        if (info == 0) f->kind = 2;
        else {
            log_message(LOG_DEBUG, strprintf("lu_factorization of a symmetric operator: not positive definite (leading minor %lld), factorising with pivoting", (long long)info));
            expand();
        }
    }
    if (info != 0) info = run(f->kind);
    HM_CHECK(info == 0, kind == 1 ? "lu_factorization: singular matrix" : "cholesky_factorization: matrix is not positive definite");
    log_message(LOG_INFO, strprintf("dense %s of the %d x %d operator on the device: expansion %.3f s, factorisation %.3f s (dense fallback: hierarchical LU is not part of this engine)",
                                    f->kind == 1 ? "LU" : (kind == 1 ? "LU by Cholesky (symmetric positive definite)" : "Cholesky"), n, n, t_expand, wall_seconds() - t_begin - t_expand));
    return f.release();
}

// B (n x mu, column c at B_dev + c * ldb, CLUSTER numbering of the operator's rows) <- solution, enqueued on `stream`
void device_dense_solve(const DeviceDenseFactor *f, char trans, void *B_dev, long long ldb, int mu, void *stream) {
    HM_CHECK(f != nullptr, "factor solve: no factorisation");
    HM_CHECK(trans == 'N' || trans == 'T' || trans == 'C', "factor solve: trans must be 'N', 'T' or 'C'");
    HM_CHECK(ldb >= f->n && mu >= 0, "factor solve: bad leading dimension");
    if (f->n == 0 || mu == 0) return;
    HIP_OK(hipSetDevice(f->device));
    HM_CHECK(g_solver.set_stream(f->handle, (hipStream_t)stream) == 0, "rocblas_set_stream failed");
    rb_status rs;
    if (f->kind == 1) {
        const int op = trans == 'N' ? RB_OP_N : (trans == 'C' && f->is_complex ? RB_OP_C : RB_OP_T);
        rs = f->is_complex ? g_solver.zgetrs(f->handle, op, f->n, mu, (rb_z *)f->a, f->n, f->ipiv, (rb_z *)B_dev, (int64_t)ldb)
                           : g_solver.dgetrs(f->handle, op, f->n, mu, (double *)f->a, f->n, f->ipiv, (double *)B_dev, (int64_t)ldb);
    } else rs = g_solver.dpotrs(f->handle, f->uplo == 'U' ? RB_UPPER : RB_LOWER, f->n, mu, (double *)f->a, f->n, (double *)B_dev, (int64_t)ldb);
    HM_CHECK(rs == 0, strprintf("the dense solver library reported status %d", rs));
}

// host right-hand sides in USER numbering (the reference's lu_solve / cholesky_solve): permuted to the operator's cluster
// numbering on the device, solved, permuted back
void device_dense_solve_host(const HMatrix &H, const DeviceDenseFactor *f, char trans, void *B, int mu) {
    const int n = f->n;
    if (n == 0 || mu == 0) return;
    const size_t es = f->is_complex ? 16 : 8;
    DeviceHMatrix *D = H.dev;
    HIP_OK(hipSetDevice(f->device));
    const bool whole = H.t_root == 0 && !H.local_numbering; // (partition-built blocks take and return their slice in cluster order)
    void *d_in = nullptr, *d_cl = nullptr;
    HIP_OK(hipMalloc(&d_in, (size_t)n * mu * es));
    if (hipMalloc(&d_cl, (size_t)n * mu * es) != hipSuccess) { (void)hipFree(d_in); throw Error("factor solve: out of device memory"); }
    try {
        hipStream_t st = D->stream;
        HIP_OK(hipMemcpyAsync(d_in, B, (size_t)n * mu * es, hipMemcpyHostToDevice, st));
        const unsigned nblk = (unsigned)(((long long)n * mu + 255) / 256);
        // rows: 'N' solves A x = b with b indexed by TARGET points; 'T' / 'C' by source points -- the same cloud here (square, one tree)
        const int *perm_in = trans == 'N' ? D->perm_t : D->perm_s, *perm_out = trans == 'N' ? D->perm_s : D->perm_t;
        if (whole) {
            if (f->is_complex) hipLaunchKernelGGL(permute_rows_kernel<double2>, dim3(nblk), dim3(256), 0, st, (const double2 *)d_in, (double2 *)d_cl, perm_in, n, mu, 1);
            else hipLaunchKernelGGL(permute_rows_kernel<double>, dim3(nblk), dim3(256), 0, st, (const double *)d_in, (double *)d_cl, perm_in, n, mu, 1);
        } else HIP_OK(hipMemcpyAsync(d_cl, d_in, (size_t)n * mu * es, hipMemcpyDeviceToDevice, st));
        device_dense_solve(f, trans, d_cl, n, mu, (void *)st);
        if (whole) {
            if (f->is_complex) hipLaunchKernelGGL(permute_rows_kernel<double2>, dim3(nblk), dim3(256), 0, st, (const double2 *)d_cl, (double2 *)d_in, perm_out, n, mu, 0);
            else hipLaunchKernelGGL(permute_rows_kernel<double>, dim3(nblk), dim3(256), 0, st, (const double *)d_cl, (double *)d_in, perm_out, n, mu, 0);
        } else HIP_OK(hipMemcpyAsync(d_in, d_cl, (size_t)n * mu * es, hipMemcpyDeviceToDevice, st));
        HIP_OK(hipMemcpyAsync(B, d_in, (size_t)n * mu * es, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
    } catch (...) {
        (void)hipFree(d_in); (void)hipFree(d_cl);
        throw;
    }
    (void)hipFree(d_in); (void)hipFree(d_cl);
}

// library warm-up (device.hip: device_warm_up): the first launch of a kernel of this translation unit loads its code object
namespace hm {
__global__ void warm_kernel_dense() {}
void warm_up_dense() { hipLaunchKernelGGL(warm_kernel_dense, dim3(1), dim3(64), 0, 0); }
} // namespace hm
