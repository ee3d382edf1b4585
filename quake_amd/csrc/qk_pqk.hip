// Projected quantum kernel: per-qubit reduced density matrices, then
// K[i][j] = exp(-gamma * sum_q ||rho_i^q - rho_j^q||_F)   (reference genetic_v3.py:793-805).
//
// rdm1_kernel: one workgroup per sample streams its 2^n amplitudes once from HBM
// (later sweeps hit L2) and reduces, for every qubit q,
//     rho00 = sum_{b_q=0} |psi_b|^2, rho11 = sum_{b_q=1} |psi_b|^2,
//     rho01 = sum_{b_q=0} psi_b * conj(psi_{b | 1<<q})
// pqk_kernel: 32 x 32 output tiles, RDM rows staged in LDS, upper tiles mirrored.
#include <hip/hip_runtime.h>

#include "qk_common.h"

namespace qk {

constexpr int RDM_THREADS = 256;

__global__ void __launch_bounds__(RDM_THREADS) rdm1_kernel(const double2* __restrict__ psi, int64_t N, int n,
                                                           double* __restrict__ rdm) {
    __shared__ double red[4][RDM_THREADS / 64];
    const int64_t s = blockIdx.x;
    const double2* p = psi + (s << n);
    const uint32_t half = 1u << (n - 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = 0; q < n; ++q) {
        const uint32_t low = (1u << q) - 1u;
        double r00 = 0, r11 = 0, xr = 0, xi = 0;
        for (uint32_t i = tid; i < half; i += RDM_THREADS) {
            const uint32_t b0 = ((i & ~low) << 1) | (i & low);
            const double2 a0 = p[b0], a1 = p[b0 | (1u << q)];
            r00 = fma(a0.x, a0.x, fma(a0.y, a0.y, r00));
            r11 = fma(a1.x, a1.x, fma(a1.y, a1.y, r11));
            // a0 * conj(a1)
            xr = fma(a0.x, a1.x, fma(a0.y, a1.y, xr));
            xi = fma(a0.y, a1.x, fma(-a0.x, a1.y, xi));
        }
        double v[4] = {r00, r11, xr, xi};
        for (int k = 0; k < 4; ++k) {
            double x = v[k];
            for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
            if (lane == 0) red[k][wave] = x;
        }
        __syncthreads();
        if (tid < 4) {
            double x = 0;
            for (int w = 0; w < RDM_THREADS / 64; ++w) x += red[tid][w];
            rdm[(s * n + q) * 4 + tid] = x;
        }
        __syncthreads();
    }
}

constexpr int PT = 32;  // pqk tile side

__global__ void __launch_bounds__(256) pqk_kernel(const double* __restrict__ rx, int64_t N,
                                                   const double* __restrict__ ry, int64_t M, int n, double gamma,
                                                   double* __restrict__ K, int64_t ldk, int sym, int tiles_n) {
    extern __shared__ double sh[];  // [PT][n*4] rows of x, then of y
    const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n;
    if (sym && tj < ti) return;
    const int w = n * 4;
    double* sx = sh;
    double* sy = sh + PT * w;
    const int tid = threadIdx.x;
    for (int e = tid; e < PT * w; e += 256) {
        const int r = e / w, c = e % w;
        const int64_t gi = (int64_t)ti * PT + r, gj = (int64_t)tj * PT + r;
        sx[e] = gi < N ? rx[gi * w + c] : 0.0;
        sy[e] = gj < M ? ry[gj * w + c] : 0.0;
    }
    __syncthreads();
    const int cx = tid & 31, ry0 = tid >> 5;  // 8 rows of threads, 4 rows each
    for (int rr = 0; rr < 4; ++rr) {
        const int r = ry0 + 8 * rr;
        const int64_t i = (int64_t)ti * PT + r, j = (int64_t)tj * PT + cx;
        if (i >= N || j >= M) continue;
        double acc = 0.0;
        for (int q = 0; q < n; ++q) {
            const double* a = sx + r * w + 4 * q;
            const double* b = sy + cx * w + 4 * q;
            const double d0 = a[0] - b[0], d1 = a[1] - b[1], dr = a[2] - b[2], di = a[3] - b[3];
            // Frobenius norm of the 2x2 difference: |d00|^2 + |d11|^2 + 2 |d01|^2
            acc += sqrt(fma(d0, d0, fma(d1, d1, 2.0 * fma(dr, dr, di * di))));
        }
        double v = exp(-gamma * acc);
        if (sym) {
            if (i == j) v = 1.0;            // reference adds the identity (genetic_v3.py:805)
            else if (ti == tj && i > j) continue;
            K[i * ldk + j] = v;
            if (i != j) K[j * ldk + i] = v;
        } else {
            K[i * ldk + j] = v;
        }
    }
}

}  // namespace qk

extern "C" int qk_rdm1_dev(const void* psi_dev, int64_t N, int n_qubits, double* rdm_dev, void* stream) {
    using namespace qk;
    if (!psi_dev || !rdm_dev) return fail(QK_ERR_INVALID, "qk_rdm1_dev: null pointer");
    if (n_qubits < 1 || n_qubits > 30) return fail(QK_ERR_INVALID, "qk_rdm1_dev: n_qubits out of range");
    if (N < 0 || N > 0x7fffffffLL) return fail(QK_ERR_INVALID, "qk_rdm1_dev: N out of range");
    if (N == 0) return QK_OK;
    hipLaunchKernelGGL(rdm1_kernel, dim3((unsigned)N), dim3(RDM_THREADS), 0, (hipStream_t)stream,
                       (const double2*)psi_dev, N, n_qubits, rdm_dev);
    QK_HIP(hipGetLastError());
    return QK_OK;
}

extern "C" int qk_pqk_dev(const double* rdm_x, int64_t N, const double* rdm_y, int64_t M, int n_qubits,
                          double gamma, double* K_dev, int64_t ldk, void* stream) {
    using namespace qk;
    if (!rdm_x || !K_dev) return fail(QK_ERR_INVALID, "qk_pqk_dev: null pointer");
    if (n_qubits < 1 || n_qubits > 30) return fail(QK_ERR_INVALID, "qk_pqk_dev: n_qubits out of range");
    const int sym = rdm_y ? 0 : 1;
    if (sym) { rdm_y = rdm_x; M = N; }
    if (N < 0 || M < 0 || ldk < M) return fail(QK_ERR_INVALID, "qk_pqk_dev: bad shape");
    if (N == 0 || M == 0) return QK_OK;
    const int64_t tm = (N + PT - 1) / PT, tn = (M + PT - 1) / PT;
    if (tm * tn > 0x7fffffffLL) return fail(QK_ERR_UNSUPPORTED, "qk_pqk_dev: too many tiles");
    const size_t lds = (size_t)2 * PT * n_qubits * 4 * sizeof(double);
    hipLaunchKernelGGL(pqk_kernel, dim3((unsigned)(tm * tn)), dim3(256), lds, (hipStream_t)stream, rdm_x, N, rdm_y, M,
                       n_qubits, gamma, K_dev, ldk, sym, (int)tn);
    QK_HIP(hipGetLastError());
    return QK_OK;
}
