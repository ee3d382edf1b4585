// Ceiling check: back-to-back v_mfma_f64_16x16x4_f64 from registers only (no LDS, no HBM).
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_peak.hip -o gpurun_out/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(512, 2) k(double* out, int iters, double x, double y) {
    v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = x + threadIdx.x * 1e-9, b = y - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int threads, int blocks_per_cu, const char* tag) {
    int blocks = 256 * blocks_per_cu, iters = 20000;
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = 2048.0 * NACC * iters * (double)blocks * (threads / 64);
        if (rep == 2) printf("%-28s threads %4d blocks/CU %d nacc %2d: %.2f ms  %.2f TFLOP/s (%.1f%% of 78.6)\n", tag, threads, blocks_per_cu, NACC, ms, flops / ms / 1e9, flops / ms / 1e9 / 78.6 * 100);
    }
    hipFree(out);
}
int main() {
    run<16>(256, 1, "1 wave/SIMD");
    run<16>(512, 1, "2 waves/SIMD");
    run<4>(512, 1, "2 waves/SIMD");
    run<2>(256, 1, "1 wave/SIMD dep-limited");
    run<1>(256, 1, "1 wave/SIMD fully dependent");
    run<16>(256, 2, "2 blocks x 1 wave/SIMD");
    return 0;
}
