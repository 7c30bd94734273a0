// krylov_device.hip -- the finishing kernel of an Arnoldi step of the package's GMRES (htool_python_amd/krylov.py), which stands
// where the reference hands its operator to HPDDM (src/htool/solver/solver.hpp:22-65).
//
// A step is: w = A v_j (the H-matrix product), two classical Gram-Schmidt passes (GEMVs of the BLAS library) with the square
// w.w riding on the second pass's reduce, and then a handful of scalar operations per right-hand side -- the norm
// hn^2 = w.w - |h2|^2, its reciprocal square root, the scaling of w, the packing of the coefficients the host wants to see.
// Done with tensor-library calls that is ten launches of one-element kernels (about 50 us at 62 500 rows, a tenth of the
// product); here it is ONE launch: every workgroup recomputes the (tiny) scalar part for its right-hand side and scales its
// piece of w; the first workgroup of a right-hand side also writes the coefficient row [h1 | h2 | w.w | hn^2].
#include "capi_internal.hpp"
#include "device_internal.hpp"

using namespace hm;

namespace {

constexpr int KR_CH = 512;   // entries of a vector per workgroup (two per thread)

template <typename T> struct KOps;
template <> struct KOps<double> {
    static __device__ __forceinline__ double zero() { return 0.0; }
    static __device__ __forceinline__ double cmul(double a, double b) { return a * b; }                       // conj(a) b
    static __device__ __forceinline__ double mul(double a, double b) { return a * b; }
    static __device__ __forceinline__ double add(double a, double b) { return a + b; }
    static __device__ __forceinline__ double sub(double a, double b) { return a - b; }
    static __device__ __forceinline__ double shfl_xor(double a, int m) { return __shfl_xor(a, m); }
    static __device__ __forceinline__ double scale(double a, double s) { return a * s; }
};
template <> struct KOps<double2> {
    static __device__ __forceinline__ double2 zero() { return make_double2(0.0, 0.0); }
    static __device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x); }
    static __device__ __forceinline__ double2 mul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
    static __device__ __forceinline__ double2 add(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
    static __device__ __forceinline__ double2 sub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
    static __device__ __forceinline__ double2 shfl_xor(double2 a, int m) { return make_double2(__shfl_xor(a.x, m), __shfl_xor(a.y, m)); }
    static __device__ __forceinline__ double2 scale(double2 a, double s) { return make_double2(a.x * s, a.y * s); }
};

// The scalar tail of the step.  T = double or double2.  h1: (mu, j + 1), t2: (mu, j + 2) = [h2 | w.w], coef: (mu, 2 j + 4), all
// contiguous.  With V given, w -= sum_l h2[l] v_l is applied first (the second pass's projection), in the same sweep as the scaling.
template <typename T>
__global__ __launch_bounds__(256) void krylov_finish_step_kernel(T *W, long long ldw, int n, const T *h1, const T *t2, int j, const double *mask, T *coef, int scale,
                                                                 const T *__restrict__ V, long long ld_basis, long long ld_rhs) {
    typedef KOps<T> K;
    constexpr bool cplx = sizeof(T) == 16;
    const int c = blockIdx.y;
    const T *t = t2 + (long long)c * (j + 2);
    const double *td = reinterpret_cast<const double *>(t);
    // |h2|^2 in index order (every thread the same sum: uniform loads, j + 1 terms)
    double s = 0;
    for (int l = 0; l <= j; l++) {
        if (cplx) s += td[2 * l] * td[2 * l] + td[2 * l + 1] * td[2 * l + 1];
        else s += td[l] * td[l];
    }
    const double ww = cplx ? td[2 * (j + 1)] : td[j + 1];
    const double hn2 = ww - s;
    if (blockIdx.x == 0) {
        T *o = coef + (long long)c * (2 * j + 4);
        const T *a = h1 + (long long)c * (j + 1);
        for (int l = threadIdx.x; l <= j; l += 256) o[l] = a[l];
        for (int l = threadIdx.x; l <= j + 1; l += 256) o[j + 1 + l] = t[l];
        if (threadIdx.x == 0) {
            double *od = reinterpret_cast<double *>(o + 2 * j + 3);
            od[0] = hn2;
            if (cplx) od[1] = 0.0;
        }
    }
    if (!scale && !V) return;
    double inv = 1.0;
    if (scale) {
        inv = hn2 > 0 ? 1.0 / sqrt(hn2) : 0.0;
        if (mask) inv *= mask[c];
    }
    T *w = W + (long long)c * ldw;
    const T *Vc = V ? V + (long long)c * ld_rhs : nullptr;
    for (int x = blockIdx.x * KR_CH + threadIdx.x; x < min(n, (int)(blockIdx.x + 1) * KR_CH); x += 256) {
        T v = w[x];
        if (Vc) {
#pragma unroll 4
            for (int l = 0; l <= j; l++) v = K::sub(v, K::mul(t[l], Vc[(long long)l * ld_basis + x]));
        }
        w[x] = K::scale(v, inv);
    }
}

} // namespace

extern "C" int htool_krylov_finish_step(void *W_dev, int64_t ldw, int n, int mu, int is_complex, const void *h1_dev, const void *t2_dev, int j, const double *mask_dev,
                                        void *coef_dev, int scale, const void *V_dev, int64_t ld_basis, int64_t ld_rhs, void *stream) {
    API_BEGIN
    HM_CHECK(W_dev && h1_dev && t2_dev && coef_dev && n >= 0 && mu >= 1 && j >= 0 && ldw >= n, "htool_krylov_finish_step: bad argument");
    const dim3 grid((unsigned)std::max(1, (n + KR_CH - 1) / KR_CH), (unsigned)mu), block(256);
    if (is_complex) hipLaunchKernelGGL(krylov_finish_step_kernel<double2>, grid, block, 0, (hipStream_t)stream, (double2 *)W_dev, (long long)ldw, n, (const double2 *)h1_dev, (const double2 *)t2_dev, j, mask_dev,
                                       (double2 *)coef_dev, scale, (const double2 *)V_dev, (long long)ld_basis, (long long)ld_rhs);
    else hipLaunchKernelGGL(krylov_finish_step_kernel<double>, grid, block, 0, (hipStream_t)stream, (double *)W_dev, (long long)ldw, n, (const double *)h1_dev, (const double *)t2_dev, j, mask_dev, (double *)coef_dev,
                            scale, (const double *)V_dev, (long long)ld_basis, (long long)ld_rhs);
    HIP_OK(hipGetLastError());
    API_END
}

// library warm-up (device.hip: device_warm_up): the first launch of a kernel of this translation unit loads its code object
namespace hm {
__global__ void warm_kernel_krylov() {}
void warm_up_krylov() { hipLaunchKernelGGL(warm_kernel_krylov, dim3(1), dim3(64), 0, 0); }
} // namespace hm
