// Projected quantum kernel: per-qubit reduced density matrices, then
// K[i][j] = exp(-gamma * sum_q ||rho_i^q - rho_j^q||_F)   (reference genetic_v3.py:793-805).
//
// rdm1_lds_kernel (n <= 13) stages a sample in LDS and reduces every qubit in one sweep;
// rdm1_kernel (n > 13): one workgroup per sample streams its 2^n amplitudes once from HBM
// (later sweeps hit L2).  Both reduce, for every qubit q,
//     rho00 = sum_{b_q=0} |psi_b|^2, rho11 = sum_{b_q=1} |psi_b|^2,
//     rho01 = sum_{b_q=0} psi_b * conj(psi_{b | 1<<q})
// pqk_kernel: 64 x 64 output tiles (4 x 4 per thread), RDM rows staged in LDS, upper tiles mirrored.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "qk_common.h"
#include "qk_rdm.h"

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

// n <= 13: the sample's statevector is staged in LDS once (one coalesced HBM read, the algorithmic
// minimum), then rdm1_from_lds (qk_rdm.h) reduces every qubit in one sweep.
__global__ void __launch_bounds__(RDM_THREADS) rdm1_lds_kernel(const double2* __restrict__ psi, int64_t N, int n,
                                                               double* __restrict__ rdm) {
    extern __shared__ double2 st[];                 // 2^n amplitudes, then RDM_SLOTS * waves partial sums
    const uint32_t D = 1u << n;
    double* red = (double*)(st + D);
    const int tid = threadIdx.x;
    for (int64_t s = blockIdx.x; s < N; s += gridDim.x) {
        const double2* p = psi + (s << n);
        for (uint32_t b = tid; b < D; b += RDM_THREADS) st[b] = p[b];
        __syncthreads();
        rdm1_from_lds(st, n, red, rdm + s * n * 4, tid, RDM_THREADS);
    }
}

constexpr int PT = 64;  // pqk tile side: 256 threads x (4 x 4) entries

// sqrt for x in [0, 16]: the pairwise kernel is bound by one square root per (pair, qubit), and the
// library routine spends half its instructions on range scaling this input never needs.  Hardware
// reciprocal-square-root seed (~26 bits), one Goldschmidt step and a residual correction: the kernel
// matrix stays within 1 ulp of a long-double evaluation for distances from 1 down to 1e-12.
__device__ __forceinline__ double sqrt_small(double x) {
    const double y = __builtin_amdgcn_rsq(x);        // +inf for x == 0, handled by the select below
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);                  // residual correction: g += (x - g^2) * h
    g = fma(d, h, g);
    return x < 1e-280 ? 0.0 : g;
}

// LDS image per side: [qubit][component][row] (rows contiguous): the 16 lanes that differ in their
// column read 16 consecutive doubles (conflict free), the lanes that share a row read one address
// (broadcast).  Each thread keeps a 4 x 4 block of entries in registers, so one (row, qubit) RDM is
// read once per 4 entries instead of once per entry.
__global__ void __launch_bounds__(256) pqk_kernel(const double* __restrict__ rx, int64_t N,
                                                   const double* __restrict__ ry, int64_t M, int n, double gamma,
                                                   double* __restrict__ K, int64_t ldk, int sym, int tiles_n) {
    extern __shared__ double sh[];  // x side then y side, each [n][4][PT]
    int ti, tj;
    if (sym) {  // upper-triangular tile list: blockIdx -> (ti, tj >= ti)
        int rem = blockIdx.x, row = 0, len = tiles_n;
        while (rem >= len) { rem -= len; --len; ++row; }
        ti = row;
        tj = row + rem;
    } else {
        ti = blockIdx.x / tiles_n;
        tj = blockIdx.x % tiles_n;
    }
    const int w = n * 4;
    double* sx = sh;
    double* sy = sh + PT * w;
    const int tid = threadIdx.x;
    for (int e = tid; e < PT * w; e += 256) {
        const int r = e / w, c = e % w;   // global reads: consecutive lanes walk one RDM row
        const int64_t gi = (int64_t)ti * PT + r, gj = (int64_t)tj * PT + r;
        sx[c * PT + r] = gi < N ? rx[gi * w + c] : 0.0;
        sy[c * PT + r] = gj < M ? ry[gj * w + c] : 0.0;
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
    for (int q = 0; q < n; ++q) {
        double a[4][4], b[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r][k] = sx[(q * 4 + k) * PT + ty + 16 * r];
                b[r][k] = sy[(q * 4 + k) * PT + tx + 16 * r];
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double d0 = a[r][0] - b[c][0], d1 = a[r][1] - b[c][1];
                const double dr = a[r][2] - b[c][2], di = a[r][3] - b[c][3];
                // Frobenius norm of the 2x2 difference: |d00|^2 + |d11|^2 + 2 |d01|^2
                acc[r][c] += sqrt_small(fma(d0, d0, fma(d1, d1, 2.0 * fma(dr, dr, di * di))));
            }
    }
    const bool diag_tile = sym && ti == tj;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t i = (int64_t)ti * PT + ty + 16 * r, j = (int64_t)tj * PT + tx + 16 * c;
            if (i >= N || j >= M) continue;
            double v = exp(-gamma * acc[r][c]);
            if (sym) {
                if (i == j) v = 1.0;            // reference adds the identity (genetic_v3.py:805)
                else if (diag_tile && i > j) continue;
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
    if (n_qubits <= RDM_MAXQ) {
        const size_t lds = ((size_t)16 << n_qubits) + (size_t)RDM_SLOTS * (RDM_THREADS / 64) * sizeof(double);
        static DeviceOnce attr_set;
        if (attr_set.pending()) {
            QK_HIP(hipFuncSetAttribute((const void*)rdm1_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set.done();
        }
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
        const int64_t blocks = std::min<int64_t>(N, (int64_t)256 * per_cu);
        hipLaunchKernelGGL(rdm1_lds_kernel, dim3((unsigned)blocks), dim3(RDM_THREADS), lds, (hipStream_t)stream,
                           (const double2*)psi_dev, N, n_qubits, rdm_dev);
    } else {
        hipLaunchKernelGGL(rdm1_kernel, dim3((unsigned)N), dim3(RDM_THREADS), 0, (hipStream_t)stream,
                           (const double2*)psi_dev, N, n_qubits, rdm_dev);
    }
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
    const int64_t nblocks = sym ? tn * (tn + 1) / 2 : tm * tn;
    if (nblocks > 0x7fffffffLL) return fail(QK_ERR_UNSUPPORTED, "qk_pqk_dev: too many tiles");
    const size_t lds = (size_t)2 * PT * n_qubits * 4 * sizeof(double);
    if (lds > 64 * 1024) {
        static DeviceOnce attr_set;
        if (attr_set.pending()) {
            QK_HIP(hipFuncSetAttribute((const void*)pqk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set.done();
        }
    }
    hipLaunchKernelGGL(pqk_kernel, dim3((unsigned)nblocks), dim3(256), lds, (hipStream_t)stream, rdm_x, N, rdm_y, M,
                       n_qubits, gamma, K_dev, ldk, sym, (int)tn);
    QK_HIP(hipGetLastError());
    return QK_OK;
}
