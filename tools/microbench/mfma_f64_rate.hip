// mfma_f64_rate.hip -- issue rate of v_mfma_f64_16x16x4_f64 on gfx950 with 1, 2, 4, 8 accumulators per wave, visited round-robin one
// MFMA at a time or in runs of 2 / 4 / 8 MFMAs on the same accumulator, and 1, 2 waves per SIMD: the matrix-core bound that the
// 16-wide sweeps of product_mfma.inc are priced against.  __launch_bounds__(256, 2): a budget of 256 registers makes
// the compiler keep the accumulators in VGPRs (with 512 it selects the AGPR form and copies every accumulator in and out around each MFMA
// of this loop -- 16 v_accvgpr moves per MFMA, which is what a first version of this file measured for 4 and 8 accumulators).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_f64_rate.hip -o tools/microbench/_bin/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int CHAINS, int RUN = 1>
__global__ __launch_bounds__(256, 2) void rate_kernel(double *out, int iters, double a0, double b0) {
    v4f64 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = v4f64{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < (8 / CHAINS / RUN > 0 ? 8 / CHAINS / RUN : 1); r++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++)
#pragma unroll
                for (int q = 0; q < RUN; q++) {
                    acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0); // (keep the order written here)
                }
    }
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS, int RUN = 1>
static void run(int wg_per_cu, double *out) {
    const int iters = 20000, blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<CHAINS, RUN>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0, 1e-300);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rate_kernel<CHAINS, RUN>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-300);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_iter = (8 / CHAINS / RUN > 0 ? 8 / CHAINS / RUN : 1) * CHAINS * RUN;
    const double mfmas_per_simd = (double)iters * per_iter * wg_per_cu; // one wave of each workgroup per SIMD
    const double ns_each = ms * 1e6 / mfmas_per_simd;
    const double tflops = (double)blocks * 4 * iters * per_iter * 2048.0 / (ms * 1e-3) / 1e12;
    printf("accumulators %d, runs of %d, waves/SIMD %d: %.3f ms, %.2f ns per MFMA per SIMD, %.1f TFLOP/s\n", CHAINS, RUN, wg_per_cu, ms, ns_each, tflops);
}
int main() {
    double *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
    for (int w = 1; w <= 2; w++) { run<1>(w, out); run<2>(w, out); run<4>(w, out); run<8>(w, out); run<8, 2>(w, out); run<8, 4>(w, out); run<8, 8>(w, out); run<2, 4>(w, out); run<1>(w, out); }
    hipFree(out);
    return 0;
}
