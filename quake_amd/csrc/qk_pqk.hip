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

constexpr int PT = 64;   // pqk tile side: 256 threads x (4 x 4) entries
constexpr int PTP = 65;  // row stride of the LDS image: the staging writes walk components, 65 keeps them on distinct banks

// The pairwise kernel is bound by VALU issue (PMC: 88 % busy), most of it the square root per (pair,
// qubit); the library routine spends half its instructions on range scaling this input (x in [0, 16])
// never needs.  Here: hardware reciprocal-square-root seed y (~26 bits), g = x * y, and one residual
// correction, sqrt(x) = g + (x - g^2) * y / 2 (relative error ~2^-52).  The loop accumulates TWICE the
// root, 2 g + (x - g^2) y, as two fused multiply-adds; the factor 1/2 goes into the exponent.  x carries
// an addend of 1e-300 so that rsq never sees 0 (identical rows: sqrt = 1e-150 instead of 0, far below
// one ulp of the exponent's argument).
__device__ __forceinline__ double add_twice_sqrt(double acc, double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y;
    const double d = fma(-g, g, x);
    return fma(2.0, g, fma(d, y, acc));
}

// exp(x) for the epilogue, one fused multiply-add per polynomial step: the coefficients are read from
// a mutable global (so they are not folded into the instruction stream) with scalar loads and sit
// in SGPRs as the addend of v_fma_f64 -- the library routine's constants have to be moved into the
// destination register pair of a v_fmac_f64 before every step (3 VALU instructions per step, and
// the epilogue was a quarter of the kernel's VALU time at 12 qubits).
// k = rint(x / ln 2), r = x - k ln 2 in two pieces (|r| <= 0.347), Taylor polynomial of degree 13
// (remainder < 4e-18), result scaled by 2^k; underflows to 0 like the library routine.
__device__ double g_exp_coef[12] = {
    1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0,
    1.0 / 5040.0,       1.0 / 720.0,       1.0 / 120.0,      1.0 / 24.0,      1.0 / 6.0,      0.5};

__device__ __forceinline__ double exp_poly(double x, const double (&c)[12]) {
    x = fmax(x, -800.0);
    const double k = rint(x * 1.4426950408889634);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = c[0];
#pragma unroll
    for (int t = 1; t < 12; ++t) p = fma(p, r, c[t]);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// LDS image per side: [qubit][component][row] (rows contiguous): the 16 lanes that differ in their
// column read 16 consecutive doubles (conflict free), the lanes that share a row read one address
// (broadcast).  Each thread keeps a 4 x 4 block of entries in registers, so one (row, qubit) RDM is
// read once per 4 entries instead of once per entry.
__global__ void __launch_bounds__(256) pqk_kernel(const double* __restrict__ rx, int64_t N,
                                                   const double* __restrict__ ry, int64_t M, int n, double gamma,
                                                   double* __restrict__ K, int64_t ldk, int sym, int tiles_n) {
    extern __shared__ double sh[];  // x side then y side, each [n][3][PTP]
    int ti, tj;
    if (sym) {  // upper-triangular tile list: blockIdx -> (ti, tj >= ti); T(r) = r * tiles_n - r (r - 1) / 2 tiles precede row r
        const int b = blockIdx.x;
        const double t2 = 2.0 * tiles_n + 1.0;
        int row = (int)(0.5 * (t2 - sqrt(fmax(t2 * t2 - 8.0 * b, 0.0))));
        row = row < 0 ? 0 : (row >= tiles_n ? tiles_n - 1 : row);
        auto before = [&](int r) { return (int64_t)r * tiles_n - (int64_t)r * (r - 1) / 2; };
        while (row > 0 && before(row) > b) --row;                 // guards the rounding of the estimate
        while (row + 1 < tiles_n && before(row + 1) <= b) ++row;
        ti = row;
        tj = row + (int)(b - before(row));
    } else {
        ti = blockIdx.x / tiles_n;
        tj = blockIdx.x % tiles_n;
    }
    const int w = n * 4;          // doubles per row of the RDM table (rho00, rho11, Re rho01, Im rho01 per qubit)
    const int w3 = n * 3;         // ... of the LDS image: rho11 = 1 - rho00 carries nothing the norm does not already have
    double* sx = sh;
    double* sy = sh + PTP * w3;
    const int tid = threadIdx.x;
    {
        // The tile's 64 RDM rows are one contiguous run of 64 * w doubles: read it linearly (coalesced)
        // and keep (row, component) of the element by increments instead of a division per element.
        const int total = PT * w;
        const int64_t xbase = (int64_t)ti * total, ybase = (int64_t)tj * total;
        const int xl = (int)(N * w - xbase < total ? N * w - xbase : total);   // elements of existing rows
        const int yl = (int)(M * w - ybase < total ? M * w - ybase : total);
        int r = tid / w, c = tid - r * w;
        const int dr = 256 / w, dc = 256 - dr * w;
        for (int e = tid; e < total; e += 256) {
            // ||d rho||_F^2 = d00^2 + d11^2 + 2 |d01|^2 = 2 (d00^2 + Re d01^2 + Im d01^2) since d11 = -d00 (unit trace):
            // the three kept components are stored times sqrt(2) and the norm is a plain sum of three squares
            const int comp = c & 3;
            if (comp != 1) {
                const int slot = (c >> 2) * 3 + (comp ? comp - 1 : 0);
                sx[slot * PTP + r] = e < xl ? 1.4142135623730951 * rx[xbase + e] : 0.0;
                sy[slot * PTP + r] = e < yl ? 1.4142135623730951 * ry[ybase + e] : 0.0;
            }
            r += dr;
            c += dc;
            if (c >= w) { c -= w; ++r; }
        }
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
    for (int q = 0; q < n; ++q) {
        double a[4][3], b[4][3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r][k] = sx[(q * 3 + k) * PTP + ty + 16 * r];
                b[r][k] = sy[(q * 3 + k) * PTP + tx + 16 * r];
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double d0 = a[r][0] - b[c][0], dr = a[r][1] - b[c][1], di = a[r][2] - b[c][2];
                acc[r][c] = add_twice_sqrt(acc[r][c], fma(d0, d0, fma(dr, dr, fma(di, di, 1e-300))));
            }
    }
    // exponentials first, branch-free over the whole 4 x 4 block (so the polynomial's constants are
    // loaded once), then the predicated stores
    double ec[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) ec[t] = g_exp_coef[t];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = exp_poly(-0.5 * gamma * acc[r][c], ec);   // acc = 2 * sum_q ||rho_i^q - rho_j^q||_F
    const bool diag_tile = sym && ti == tj;
    const int i0 = ti * PT + ty, j0 = tj * PT + tx;    // N, M < 2^31 (checked by the host)
    const int Ni = (int)N, Mi = (int)M;
    double* KjT[4];   // mirror entries: row j of K, offset to column i0
#pragma unroll
    for (int c = 0; c < 4; ++c) KjT[c] = K + (int64_t)(j0 + 16 * c) * ldk + i0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + 16 * r;
        if (i >= Ni) continue;
        double* Ki = K + (int64_t)i * ldk + j0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = j0 + 16 * c;
            if (j >= Mi) continue;
            double v = acc[r][c];
            if (sym) {
                if (i == j) v = 1.0;            // reference adds the identity (genetic_v3.py:805)
                else if (diag_tile && i > j) continue;
                Ki[16 * c] = v;
                if (i != j) KjT[c][16 * r] = v;
            } else {
                Ki[16 * c] = v;
            }
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
    if (N > 0x7fffffffLL - PT || M > 0x7fffffffLL - PT) return fail(QK_ERR_UNSUPPORTED, "qk_pqk_dev: more than 2^31 rows");
    if (N == 0 || M == 0) return QK_OK;
    const int64_t tm = (N + PT - 1) / PT, tn = (M + PT - 1) / PT;
    const int64_t nblocks = sym ? tn * (tn + 1) / 2 : tm * tn;
    if (nblocks > 0x7fffffffLL) return fail(QK_ERR_UNSUPPORTED, "qk_pqk_dev: too many tiles");
    const size_t lds = (size_t)2 * PTP * n_qubits * 3 * sizeof(double);
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
