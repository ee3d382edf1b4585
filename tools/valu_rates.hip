// Issue cost of FP64 vector instructions on gfx950, one wave per SIMD: tools/valu_rates.hip -> cycles per instruction.
// hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o tools/valu_rates.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
template <int K>
__global__ void __launch_bounds__(64) k(double* out, int iters) {
    double a0 = threadIdx.x + 1.5, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    for (int i = 0; i < iters; ++i) {
        if constexpr (K == 0) { REP8(asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (K == 1) { REP8(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (K == 2) { REP8(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if constexpr (K == 3) { REP8(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if constexpr (K == 4) { REP8(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));) }
        if constexpr (K == 5) { REP8(asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (K == 6) { REP8(asm volatile("v_mul_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_mul_f64 %2, %2, %2\n v_mul_f64 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (K == 7) { REP8(asm volatile("v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (K == 8) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if constexpr (K == 9) { REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3;
}
template <int K> void run(const char* name, double* out, int waves_per_simd) {
    const int iters = 4096, blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<K><<<blocks, 64>>>(out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<K><<<blocks, 64>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e-3 * 2.4e9 / ((double)iters * 32 * waves_per_simd);
    printf("%-14s waves/SIMD %d: %.2f cycles per instruction and wave (at 2.4 GHz)\n", name, waves_per_simd, per);
}
int main() {
    double* out; hipMalloc(&out, 256 * 4 * 4 * 64 * 8);
    for (int w : {1, 2}) {
        run<0>("v_fma_f64", out, w); run<5>("v_add_f64", out, w); run<6>("v_mul_f64", out, w); run<1>("v_rsq_f64", out, w);
        run<7>("v_sqrt_f64", out, w); run<9>("v_rcp_f64", out, w); run<2>("v_rsq_f32", out, w); run<3>("v_cvt_f32_f64", out, w);
        run<4>("v_cvt_f64_f32", out, w); run<8>("v_fma_f32", out, w);
    }
    return 0;
}
