// Ceiling check: back-to-back v_mfma_f64_16x16x4_f64 from registers only (no LDS, no HBM).
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_peak.hip -o gpurun_out/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
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
// --probe SECONDS: keep the 2-waves/SIMD loop running for about that long and print the rate of the
// second half as one JSON line (bench.py records it as the box's MFMA ceiling at its current clock)
static int probe(double seconds) {
    const int threads = 512, blocks = 256, iters = 20000;
    double* out;
    if (hipMalloc(&out, sizeof(double) * blocks * threads) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops = 2048.0 * 16 * iters * (double)blocks * (threads / 64);
    double total_ms = 0, late_ms = 0;
    int late_n = 0;
    while (total_ms < seconds * 1e3) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) return 1;
        float ms; hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms;
        if (total_ms > seconds * 5e2) { late_ms += ms; ++late_n; }
    }
    hipFree(out);
    if (!late_n) return 1;
    printf("{\"tflops\": %.3f, \"seconds\": %.3f}\n", flops * late_n / late_ms / 1e9, total_ms / 1e3);
    return 0;
}
int main(int argc, char** argv) {
    if (argc == 3 && std::string(argv[1]) == "--probe") return probe(atof(argv[2]));
    run<16>(256, 1, "1 wave/SIMD");
    run<16>(512, 1, "2 waves/SIMD");
    run<4>(512, 1, "2 waves/SIMD");
    run<2>(256, 1, "1 wave/SIMD dep-limited");
    run<1>(256, 1, "1 wave/SIMD fully dependent");
    run<16>(256, 2, "2 blocks x 1 wave/SIMD");
    return 0;
}
