// Issue rate of single VALU instructions on gfx950, relative to v_fma_f64 (which is full rate: 4 cycles
// per wave64).  One workgroup of 256 threads per CU slot, 8 independent register chains per thread.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define KERNEL(NAME, TYPE, CONSTRAINT, ASM)                                             \
    __global__ void __launch_bounds__(256) NAME(TYPE* out, int iters) {                 \
        TYPE r[8];                                                                      \
        for (int k = 0; k < 8; ++k) r[k] = (TYPE)(1.5 + threadIdx.x * 1e-3 + k);        \
        for (int it = 0; it < iters; ++it) {                                            \
            _Pragma("unroll") for (int rep = 0; rep < 4; ++rep)                         \
            _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile(ASM : "+" CONSTRAINT(r[k])); \
        }                                                                               \
        TYPE s = 0;                                                                     \
        for (int k = 0; k < 8; ++k) s += r[k];                                          \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                        \
    }

KERNEL(k_fma64, double, "v", "v_fma_f64 %0, %0, %0, %0")
KERNEL(k_mul64, double, "v", "v_mul_f64 %0, %0, %0")
KERNEL(k_rsq64, double, "v", "v_rsq_f64 %0, %0")
KERNEL(k_rcp64, double, "v", "v_rcp_f64 %0, %0")
KERNEL(k_sqrt64, double, "v", "v_sqrt_f64 %0, %0")
KERNEL(k_rsq32, float, "v", "v_rsq_f32 %0, %0")
KERNEL(k_sqrt32, float, "v", "v_sqrt_f32 %0, %0")
KERNEL(k_fma32, float, "v", "v_fma_f32 %0, %0, %0, %0")
KERNEL(k_exp32, float, "v", "v_exp_f32 %0, %0")

__global__ void __launch_bounds__(256) k_cvt(double* out, int iters) {
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = 1.5 + threadIdx.x * 1e-3 + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float f;
                asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(r[k]));
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r[k]) : "v"(f));
            }
    }
    double s = 0;
    for (int k = 0; k < 8; ++k) s += r[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename T, typename F>
static int run(const char* name, F kernel, double base_ms, double* base_out) {
    T* out;
    const int blocks = 256 * 8, iters = 20000;
    CHK(hipMalloc(&out, blocks * 256 * sizeof(T)));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 100);
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms;
    CHK(hipEventElapsedTime(&ms, a, b));
    // wave instructions per SIMD: blocks * 4 waves * iters * 32 / 1024 SIMDs
    const double per_simd = (double)blocks * 4 * iters * 32 / 1024;
    if (base_out) *base_out = ms;
    printf("%-10s %8.2f ms  %6.2f ns per wave-instruction per SIMD", name, ms, ms * 1e6 / per_simd);
    if (base_ms > 0) printf("  = %.2f x v_fma_f64 (x4 cycles = %.1f cycles)", ms / base_ms, 4.0 * ms / base_ms);
    printf("\n");
    CHK(hipFree(out));
    return 0;
}

int main() {
    double base = 0;
    if (run<double>("fma_f64", k_fma64, 0, &base)) return 1;
    run<double>("mul_f64", k_mul64, base, nullptr);
    run<double>("rsq_f64", k_rsq64, base, nullptr);
    run<double>("rcp_f64", k_rcp64, base, nullptr);
    run<double>("sqrt_f64", k_sqrt64, base, nullptr);
    run<float>("fma_f32", k_fma32, base, nullptr);
    run<float>("rsq_f32", k_rsq32, base, nullptr);
    run<float>("sqrt_f32", k_sqrt32, base, nullptr);
    run<float>("exp_f32", k_exp32, base, nullptr);
    run<double>("cvt pair", k_cvt, base, nullptr);   // 16 cvt_f32_f64 + 16 cvt_f64_f32 per iteration: same count of 32
    return 0;
}
