// Epilogues on a symmetric kernel matrix held in HBM: the GA's kernel statistics
// (one read of the upper triangle: 8 B per entry) and the noise models (one read-modify-write).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "qk_common.h"

namespace qk {
namespace {

constexpr int kStatThreads = 256;
constexpr int kStatMaxBlocks = 1024;   // QK_STATS_WS = kStatMaxBlocks * 8 doubles

// fixed-shape tree over the block: deterministic
__device__ __forceinline__ void block_reduce5(double v[5], double* sh /* 5 * kStatThreads */, int tid) {
#pragma unroll
    for (int k = 0; k < 5; ++k) sh[k * kStatThreads + tid] = v[k];
    __syncthreads();
    for (int s = kStatThreads / 2; s > 0; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int k = 0; k < 5; ++k) sh[k * kStatThreads + tid] += sh[k * kStatThreads + tid + s];
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = sh[k * kStatThreads];
}

// stage 1: block b owns rows b, b + gridDim, ... ; lanes walk the columns j > i (coalesced)
__global__ void __launch_bounds__(kStatThreads) stats_partial_kernel(const double* __restrict__ K, int64_t N, int64_t ldk,
                                                                     const int32_t* __restrict__ labels,
                                                                     double* __restrict__ partial) {
    __shared__ double sh[5 * kStatThreads];
    const int tid = threadIdx.x;
    double acc[5] = {0, 0, 0, 0, 0};
    for (int64_t i = blockIdx.x; i < N; i += gridDim.x) {
        const int li = labels ? labels[i] : -1;
        const double* row = K + i * ldk;
        for (int64_t j = i + 1 + tid; j < N; j += kStatThreads) {
            const double v = row[j];
            acc[0] += v;
            acc[1] = fma(v, v, acc[1]);
            if (labels) {
                const int lj = labels[j];
                if (li == 0 && lj == 0) acc[2] += v;
                else if (li == 1 && lj == 1) acc[3] += v;
                else if ((li == 0 && lj == 1) || (li == 1 && lj == 0)) acc[4] += v;
            }
        }
    }
    block_reduce5(acc, sh, tid);
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) partial[(int64_t)blockIdx.x * 8 + k] = acc[k];
    }
}

// stage 2: one block folds the partials in a fixed order and counts the labels
__global__ void __launch_bounds__(kStatThreads) stats_final_kernel(const double* __restrict__ partial, int n_partial,
                                                                   const int32_t* __restrict__ labels, int64_t N,
                                                                   double* __restrict__ stats) {
    __shared__ double sh[5 * kStatThreads];
    const int tid = threadIdx.x;
    double acc[5] = {0, 0, 0, 0, 0};
    for (int b = tid; b < n_partial; b += kStatThreads)
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k] += partial[(int64_t)b * 8 + k];
    block_reduce5(acc, sh, tid);
    __syncthreads();
    double cnt[5] = {0, 0, 0, 0, 0};
    if (labels)
        for (int64_t i = tid; i < N; i += kStatThreads) {
            cnt[0] += labels[i] == 0;
            cnt[1] += labels[i] == 1;
        }
    block_reduce5(cnt, sh, tid);
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) stats[k] = acc[k];
        stats[5] = cnt[0];
        stats[6] = cnt[1];
        stats[7] = 0.0;
    }
}

// ---- Philox4x32-10 (Salmon et al., SC'11), counter = (i, j, t, 0), key = seed -------------------
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// one noisy entry: Gaussian noise + clip, or a binomial estimate from `param` shots; counter = (i, j, t, 0)
__device__ __forceinline__ double noisy_value(double v, int64_t i, int64_t j, int mode, double param, uint32_t k0, uint32_t k1) {
    if (mode == QK_NOISE_GAUSSIAN) {
        const U4 r = philox((uint32_t)i, (uint32_t)j, 0u, 0u, k0, k1);
        // two 53-bit uniforms in (0, 1]
        const double u1 = ((double)(((uint64_t)r.x << 21) ^ (r.y >> 11)) + 1.0) * (1.0 / 9007199254740992.0);
        const double u2 = ((double)(((uint64_t)r.z << 21) ^ (r.w >> 11)) + 1.0) * (1.0 / 9007199254740992.0);
        const double g = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
        return fmin(fmax(v + param * g, 0.0), 1.0);
    }
    const int64_t shots = (int64_t)param;
    const double p = fmin(fmax(v, 0.0), 1.0);
    const uint64_t thr = (uint64_t)(p * 4294967296.0);       // P(u32 < thr) = p to 2^-32
    int64_t hits = 0;
    for (int64_t t = 0; t * 4 < shots; ++t) {
        const U4 r = philox((uint32_t)i, (uint32_t)j, (uint32_t)t, (uint32_t)(t >> 32), k0, k1);
        const int64_t left = shots - t * 4;
        hits += (uint64_t)r.x < thr;
        if (left > 1) hits += (uint64_t)r.y < thr;
        if (left > 2) hits += (uint64_t)r.z < thr;
        if (left > 3) hits += (uint64_t)r.w < thr;
    }
    return (double)hits / (double)shots;
}

// rows [i0, i1) of a matrix with `cols` columns, 16 x 16 thread tiles.  Symmetric: the entries right of the diagonal
// are drawn and mirrored (the diagonal stays exact); otherwise every entry is drawn.  The Philox counter of an entry
// is its (row, column) in the WHOLE matrix, so a matrix sampled band by band equals the one sampled in one launch.
__global__ void __launch_bounds__(256) noise_kernel(double* __restrict__ K, int64_t i0, int64_t i1, int64_t cols, int64_t ldk,
                                                    int symmetric, int mode, double param, uint32_t k0, uint32_t k1) {
    const int64_t j = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
    const int64_t i = i0 + (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
    if (i >= i1 || j >= cols || (symmetric && i >= j)) return;
    const double out = noisy_value(K[i * ldk + j], i, j, mode, param, k0, k1);
    K[i * ldk + j] = out;
    if (symmetric) K[j * ldk + i] = out;
}

}  // namespace
int launch_noise(double* K, int64_t i0, int64_t i1, int64_t cols, int64_t ldk, bool symmetric, int mode, double param,
                 uint64_t seed, hipStream_t stream) {
    if (mode == QK_NOISE_NONE || i1 <= i0 || cols < 1) return QK_OK;
    if (mode == QK_NOISE_GAUSSIAN) {
        if (!(param >= 0.0)) return fail(QK_ERR_INVALID, "noise: negative standard deviation");
    } else if (mode == QK_NOISE_SHOTS) {
        if (!(param >= 1.0) || param > 1e9) return fail(QK_ERR_INVALID, "noise: shots must be in 1..1e9");
    } else {
        return fail(QK_ERR_INVALID, "noise: unknown noise mode");
    }
    if (i1 > (int64_t)1 << 32 || cols > (int64_t)1 << 32) return fail(QK_ERR_UNSUPPORTED, "noise: matrix too large for the counter layout");
    const int64_t tx = (cols + 15) / 16, ty = (i1 - i0 + 15) / 16;
    if (ty > 65535) return fail(QK_ERR_UNSUPPORTED, "noise: more than 1048560 rows in one launch");     // grid.y limit
    hipLaunchKernelGGL(noise_kernel, dim3((unsigned)tx, (unsigned)ty), dim3(256), 0, stream, K, i0, i1, cols, ldk,
                       symmetric ? 1 : 0, mode, param, (uint32_t)seed, (uint32_t)(seed >> 32));
    QK_HIP(hipGetLastError());
    return QK_OK;
}
namespace {

// One GA generation's epilogue in ONE launch: block g owns chromosome g's (small) kernel matrix -- noise on the
// upper triangle (mirrored), then the statistics of the noisy matrix, reduced inside the block in a fixed order.
__global__ void __launch_bounds__(kStatThreads) batch_epilogue_kernel(const EpiJob* __restrict__ jobs, int mode, double param,
                                                                      int want_stats) {
    __shared__ double sh[5 * kStatThreads];
    const EpiJob J = jobs[blockIdx.x];
    const int tid = threadIdx.x;
    const int64_t N = J.N;
    const uint32_t k0 = (uint32_t)J.seed, k1 = (uint32_t)(J.seed >> 32);
    double acc[5] = {0, 0, 0, 0, 0};
    for (int64_t i = 0; i < N; ++i) {
        const int li = (want_stats && J.labels) ? J.labels[i] : -1;
        double* row = J.K + i * N;
        for (int64_t j = i + 1 + tid; j < N; j += kStatThreads) {
            double v = row[j];
            if (mode != QK_NOISE_NONE) {
                v = noisy_value(v, i, j, mode, param, k0, k1);
                row[j] = v;
                J.K[j * N + i] = v;
            }
            if (want_stats) {
                acc[0] += v;
                acc[1] = fma(v, v, acc[1]);
                const int lj = J.labels[j];
                if (li == 0 && lj == 0) acc[2] += v;
                else if (li == 1 && lj == 1) acc[3] += v;
                else if ((li == 0 && lj == 1) || (li == 1 && lj == 0)) acc[4] += v;
            }
        }
    }
    if (!want_stats) return;
    block_reduce5(acc, sh, tid);
    __syncthreads();
    double cnt[5] = {0, 0, 0, 0, 0};
    for (int64_t i = tid; i < N; i += kStatThreads) {
        cnt[0] += J.labels[i] == 0;
        cnt[1] += J.labels[i] == 1;
    }
    block_reduce5(cnt, sh, tid);
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) J.stats[k] = acc[k];
        J.stats[5] = cnt[0];
        J.stats[6] = cnt[1];
        J.stats[7] = 0.0;
    }
}

}  // namespace
}  // namespace qk

namespace qk {
int launch_batch_epilogue(const EpiJob* table_dev, int count, int noise_mode, double noise_param, bool want_stats,
                          hipStream_t stream) {
    if (count < 1) return QK_OK;
    hipLaunchKernelGGL(batch_epilogue_kernel, dim3((unsigned)count), dim3(kStatThreads), 0, stream, table_dev, noise_mode,
                       noise_param, want_stats ? 1 : 0);
    QK_HIP(hipGetLastError());
    return QK_OK;
}
}  // namespace qk

using namespace qk;

extern "C" int qk_kernel_stats_dev(const double* K, int64_t N, int64_t ldk, const int32_t* labels, double* stats,
                                   double* workspace, void* stream) {
    if (!K || !stats || !workspace) return fail(QK_ERR_INVALID, "qk_kernel_stats_dev: null pointer");
    if (N < 1 || ldk < N) return fail(QK_ERR_INVALID, "qk_kernel_stats_dev: bad shape");
    const int blocks = (int)std::min<int64_t>(N, kStatMaxBlocks);
    hipLaunchKernelGGL(stats_partial_kernel, dim3(blocks), dim3(kStatThreads), 0, (hipStream_t)stream, K, N, ldk, labels,
                       workspace);
    hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(kStatThreads), 0, (hipStream_t)stream, workspace, blocks, labels, N,
                       stats);
    QK_HIP(hipGetLastError());
    return QK_OK;
}

// ---- the entry point that allocates nothing -------------------------------------------------------
// The partial sums of stats_partial_kernel (the `workspace` of qk_kernel_stats_dev) live in a file-scope
// device array instead: QK_STATS_WS doubles = 64 KiB per GPU the library runs on, held until the process
// ends.  In host code the array's NAME is only a stand-in (passing it to a kernel would fault on the GPU):
// its address on a device comes from hipGetSymbolAddress, asked once per device with that device current,
// and only that pointer is ever handed to the kernels.
static __device__ __align__(256) double scratch[QK_STATS_WS];

namespace qk {
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
double* g_scratch_ptr[kMaxDevices] = {nullptr};

// the scratch array's address on the current device (looked up on the first call there).  Every GPU holds its own
// copy of the array, so the copy must be the one of the GPU that K lives on: a K on another device than the current
// one is refused here instead of handing the kernels an address that is not mapped where they run.
int scratch_on_current_device(const void* K, double** out) {
    int dev = -1;
    QK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(QK_ERR_UNSUPPORTED, "qk_kernel_stats_noalloc_dev: device index out of range");
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, K) != hipSuccess) {
        (void)hipGetLastError();      // not a pointer HIP knows: the launch below reports what is wrong with it
    } else if (at.type == hipMemoryTypeDevice && at.device != dev) {
        return fail(QK_ERR_INVALID, "qk_kernel_stats_noalloc_dev: K lives on device " + std::to_string(at.device) +
                                        " but device " + std::to_string(dev) + " is current (hipSetDevice first)");
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        QK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(QK_ERR_HIP, "qk_kernel_stats_noalloc_dev: hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = (double*)p;
    }
    *out = g_scratch_ptr[dev];
    return QK_OK;
}
}  // namespace
}  // namespace qk

extern "C" int qk_kernel_stats_noalloc_dev(const double* K, int64_t N, int64_t ldk, const int32_t* labels, double* stats,
                                           void* stream) {
    if (!K || !stats) return fail(QK_ERR_INVALID, "qk_kernel_stats_noalloc_dev: null pointer");
    if (N < 1 || ldk < N) return fail(QK_ERR_INVALID, "qk_kernel_stats_noalloc_dev: bad shape");
    if (N > QK_STATS_NOALLOC_MAX_N)
        return fail(QK_ERR_UNSUPPORTED, "qk_kernel_stats_noalloc_dev: N exceeds QK_STATS_NOALLOC_MAX_N (131072 rows); "
                                        "use qk_kernel_stats_dev with a caller-provided workspace");
    double* ws = nullptr;
    const int rc = scratch_on_current_device(K, &ws);
    if (rc) return rc;
    // same kernels, same block count and reduction tree as qk_kernel_stats_dev: bitwise the same sums.
    // Every partial the final kernel reads is written by the partial kernel of THIS call (blocks entries),
    // so nothing left over from an earlier, larger call is ever read and nothing has to be cleared.
    const int blocks = (int)std::min<int64_t>(N, kStatMaxBlocks);
    hipLaunchKernelGGL(stats_partial_kernel, dim3(blocks), dim3(kStatThreads), 0, (hipStream_t)stream, K, N, ldk, labels, ws);
    hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(kStatThreads), 0, (hipStream_t)stream, ws, blocks, labels, N, stats);
    QK_HIP(hipGetLastError());
    return QK_OK;
}

extern "C" int qk_kernel_noise_dev(double* K, int64_t N, int64_t ldk, int mode, double param, uint64_t seed, void* stream) {
    if (!K) return fail(QK_ERR_INVALID, "qk_kernel_noise_dev: null pointer");
    if (N < 1 || ldk < N) return fail(QK_ERR_INVALID, "qk_kernel_noise_dev: bad shape");
    return launch_noise(K, 0, N, N, ldk, true, mode, param, seed, (hipStream_t)stream);
}

extern "C" int qk_kernel_noise_rect_dev(double* K, int64_t rows, int64_t cols, int64_t ldk, int mode, double param,
                                        uint64_t seed, void* stream) {
    if (!K) return fail(QK_ERR_INVALID, "qk_kernel_noise_rect_dev: null pointer");
    if (rows < 1 || cols < 1 || ldk < cols) return fail(QK_ERR_INVALID, "qk_kernel_noise_rect_dev: bad shape");
    return launch_noise(K, 0, rows, cols, ldk, false, mode, param, seed, (hipStream_t)stream);
}
