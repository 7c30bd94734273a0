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
constexpr int KR_KMAX = 264; // most basis vectors (+ w itself) a pass handles: the LDS slots of the per-wave partial sums

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

// One classical Gram-Schmidt pass of mu right-hand sides against their first nvec basis vectors, ONE launch:
//   (h_in given)  w -= sum_l h_in[l] v_l          -- the projection found by the previous pass is taken out first
//   out[l] = <v_l, w> for l < nvec, and out[nvec] = <w, w> when with_ww
// Workgroup (g, c) owns entries [g KR_CH, (g + 1) KR_CH) of right-hand side c: it updates its piece of w, forms its partial
// sums, and the LAST workgroup of a right-hand side to finish (a ticket, the only atomic) adds the partials up in workgroup order:
// a fixed order, so the coefficients do not depend on the scheduling.
template <typename T>
__global__ __launch_bounds__(256) void krylov_project_kernel(const T *__restrict__ V, long long ld_basis, long long ld_rhs, int n, int nvec, T *W, long long ldw,
                                                             const T *__restrict__ h_in, int with_ww, T *partial, int *counter, T *out) {
    typedef KOps<T> K;
    __shared__ T s_p[4][KR_KMAX];
    __shared__ int s_ticket;
    const int c = blockIdx.y, g = blockIdx.x, G = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T *Vc = V + (long long)c * ld_rhs;
    T *w = W + (long long)c * ldw;
    const int x0 = g * KR_CH + tid, x1 = x0 + 256;
    const bool ok0 = x0 < n, ok1 = x1 < n;
    T w0 = ok0 ? w[x0] : K::zero(), w1 = ok1 ? w[x1] : K::zero();
    if (h_in) {
        const T *h = h_in + (long long)c * nvec;
#pragma unroll 4
        for (int l = 0; l < nvec; l++) {
            const T hl = h[l];
            const T *v = Vc + (long long)l * ld_basis;
            if (ok0) w0 = K::sub(w0, K::mul(hl, v[x0]));
            if (ok1) w1 = K::sub(w1, K::mul(hl, v[x1]));
        }
        if (ok0) w[x0] = w0;
        if (ok1) w[x1] = w1;
    }
    const int Kc = nvec + (with_ww ? 1 : 0);
    for (int l = 0; l < Kc; l++) {
        T p = K::zero();
        if (l < nvec) {
            const T *v = Vc + (long long)l * ld_basis;
            if (ok0) p = K::add(p, K::cmul(v[x0], w0));
            if (ok1) p = K::add(p, K::cmul(v[x1], w1));
        } else p = K::add(K::cmul(w0, w0), K::cmul(w1, w1));
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) p = K::add(p, K::shfl_xor(p, d));
        if (lane == 0) s_p[wave][l] = p;
    }
    __syncthreads();
    T *mine = partial + ((long long)c * G + g) * Kc;
    for (int l = tid; l < Kc; l += 256) mine[l] = K::add(K::add(K::add(s_p[0][l], s_p[1][l]), s_p[2][l]), s_p[3][l]);
    __threadfence();
    __syncthreads();
    if (tid == 0) s_ticket = atomicAdd(&counter[c], 1);
    __syncthreads();
    if (s_ticket != G - 1) return;
    __threadfence(); // the last workgroup of this right-hand side: everybody's partials are visible
    for (int l = tid; l < Kc; l += 256) {
        T s = K::zero();
        for (int q = 0; q < G; q++) s = K::add(s, partial[((long long)c * G + q) * Kc + l]);
        out[(long long)c * Kc + l] = s;
    }
    if (tid == 0) counter[c] = 0; // ready for the next pass
}

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

extern "C" int htool_krylov_project(const void *V_dev, int64_t ld_basis, int64_t ld_rhs, int n, int nvec, int mu, int is_complex, void *W_dev, int64_t ldw,
                                    const void *h_in_dev, int with_ww, void *partial_dev, int *counter_dev, void *out_dev, void *stream) {
    API_BEGIN
    HM_CHECK(V_dev && W_dev && partial_dev && counter_dev && out_dev && n >= 1 && mu >= 1 && nvec >= 0 && ldw >= n && ld_basis >= n, "htool_krylov_project: bad argument");
    HM_CHECK(nvec + (with_ww ? 1 : 0) <= KR_KMAX && nvec + (with_ww ? 1 : 0) >= 1, "htool_krylov_project: too many basis vectors for one pass");
    const dim3 grid((unsigned)((n + KR_CH - 1) / KR_CH), (unsigned)mu), block(256);
    if (is_complex) hipLaunchKernelGGL(krylov_project_kernel<double2>, grid, block, 0, (hipStream_t)stream, (const double2 *)V_dev, (long long)ld_basis, (long long)ld_rhs, n, nvec, (double2 *)W_dev,
                                       (long long)ldw, (const double2 *)h_in_dev, with_ww, (double2 *)partial_dev, counter_dev, (double2 *)out_dev);
    else hipLaunchKernelGGL(krylov_project_kernel<double>, grid, block, 0, (hipStream_t)stream, (const double *)V_dev, (long long)ld_basis, (long long)ld_rhs, n, nvec, (double *)W_dev, (long long)ldw,
                            (const double *)h_in_dev, with_ww, (double *)partial_dev, counter_dev, (double *)out_dev);
    HIP_OK(hipGetLastError());
    API_END
}
extern "C" int htool_krylov_max_basis(void) { return KR_KMAX - 1; }
extern "C" int64_t htool_krylov_partial_elements(int n, int nvec_max, int mu) { return (int64_t)mu * ((n + KR_CH - 1) / KR_CH) * (nvec_max + 1); }

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
