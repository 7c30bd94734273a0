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

// T = double or double2.  h1: (mu, j + 1), t2: (mu, j + 2) = [h2 | w.w], coef: (mu, 2 j + 4), all contiguous.
template <typename T>
__global__ __launch_bounds__(256) void krylov_finish_step_kernel(T *W, long long ldw, int n, const T *h1, const T *t2, int j, const double *mask, T *coef, int scale) {
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
    if (!scale) return;
    double inv = hn2 > 0 ? 1.0 / sqrt(hn2) : 0.0;
    if (mask) inv *= mask[c];
    double *w = reinterpret_cast<double *>(W + (long long)c * ldw);
    const long long len = cplx ? 2LL * n : n;
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < std::min<long long>(len, ((long long)blockIdx.x + 1) * 1024); i += 256) w[i] *= inv;
}

} // namespace

extern "C" int htool_krylov_finish_step(void *W_dev, int64_t ldw, int n, int mu, int is_complex, const void *h1_dev, const void *t2_dev, int j, const double *mask_dev,
                                        void *coef_dev, int scale, void *stream) {
    API_BEGIN
    HM_CHECK(W_dev && h1_dev && t2_dev && coef_dev && n >= 0 && mu >= 1 && j >= 0 && ldw >= n, "htool_krylov_finish_step: bad argument");
    const long long len = is_complex ? 2LL * n : n;
    const dim3 grid((unsigned)std::max<long long>(1, (len + 1023) / 1024), (unsigned)mu), block(256);
    if (is_complex) hipLaunchKernelGGL(krylov_finish_step_kernel<double2>, grid, block, 0, (hipStream_t)stream, (double2 *)W_dev, (long long)ldw, n, (const double2 *)h1_dev, (const double2 *)t2_dev, j, mask_dev, (double2 *)coef_dev, scale);
    else hipLaunchKernelGGL(krylov_finish_step_kernel<double>, grid, block, 0, (hipStream_t)stream, (double *)W_dev, (long long)ldw, n, (const double *)h1_dev, (const double *)t2_dev, j, mask_dev, (double *)coef_dev, scale);
    HIP_OK(hipGetLastError());
    API_END
}
