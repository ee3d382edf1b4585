// Pairwise fidelity  K[i][j] = |sum_b conj(A[i][b]) B[j][b]|^2  on FP64 MFMA (gfx950).
//
// Psi rows are interleaved complex128.  With a = A[i][b], c = B[j][b]:
//     Re = sum  a.re*c.re + a.im*c.im          Im = sum  a.re*c.im - a.im*c.re
// The MFMA k-slot of a lane (lane >> 4) is simply "which of 4 consecutive amplitudes",
// and each lane feeds the re and im halves of ONE 16-byte LDS read to the MFMAs.
//
// Two arithmetic forms per 16x16 tile and 4 amplitudes (v_mfma_f64_16x16x4_f64):
//   4M  Re += ar*cr; Re += ai*ci; Im += ar*ci; Im += (-ai)*cr      4 MFMAs, 2 accumulators
//   3M  P1 += ar*cr; P2 += ai*ci; P3 += (ar+ai)*(ci-cr)            3 MFMAs, 3 accumulators
//       Re = P1 + P2, Im = P3 + P1 - P2   (Karatsuba; the sums ar+ai, ci-cr cost one
//       VALU add per fragment).  Same result to ~1e-16; 25 % fewer matrix-core cycles.
// The algorithmic count stays 8 real flops per amplitude and entry.
//
// The shipped kernel (gram_dma_kernel, 64 x 64 block tiles, four waves, two workgroups per CU): A and B tiles
// go HBM/L2 -> LDS directly with global_load_lds (no staging registers), double buffered, one barrier per 16
// amplitudes.  The LDS image is row-major with 256-byte rows; the DMA writes linearly, so the bank swizzle sits
// on the SOURCE address (slot s of row r holds amplitude s ^ (r & 15)) and the b128 fragment reads are conflict
// free.  The register-staged kernel below it (gram_kernel, k-major LDS image with one element of padding per
// row) serves circuits under 4 qubits, where a 16-amplitude k-tile does not exist, and A/B runs.
//
// Tile schedule (gram_locate): the tiles that have to be computed are listed band by band (8 tile rows),
// column-major inside a band, and cut into groups of 64 consecutive tiles -- an 8 x 8 macro-tile without dead
// tiles -- dealt round-robin to the 8 XCD labels (block id % 8): the 64 workgroups resident on an XCD share 8 A
// and 8 B row blocks in its L2; symmetric jobs list only tiles on or above the diagonal.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "qk_common.h"

namespace qk {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));  // native vector: stays in registers (HIP's double2 class arrays went to scratch)

struct GramJobDev {
    const double2* A;
    const double2* B;
    double* K;
    double* KT;
    int64_t ma, nb, ldk, ldkt;
    int32_t flags;
    int32_t tiles_m, tiles_n;  // block tiles
    int64_t block_begin;       // first workgroup of this job
    int64_t live;              // number of tiles that have to be computed
};

struct GramArgs {
    GramJobDev job[QK_GRAM_MAX_JOBS];
    int n_jobs;
    int lr, lc;  // log2 of the macro-tile shape in block tiles (rows, columns): 4 x 8 for 128 x 128 tiles
    int64_t D;  // amplitudes per row
};

// Which block tile does this workgroup compute?  Returns false for padding workgroups.
// Workgroups with equal (block id % 8) share an XCD and its L2; the 2^(lr+lc) that run there at a time
// form ONE group of tiles (2^lr A row blocks x 2^lc B row blocks: 8 x 8 for the default 64 x 64 tiles,
// two workgroups per CU), and the 8 XCDs walk neighbouring groups of the same band.
// The tiles that have to be computed are listed band by band (a band = 2^lr tile rows), column-major
// inside a band, and the list is cut into groups of 2^(lr+lc) consecutive tiles -- a macro-tile that holds
// no tile below the diagonal of a symmetric job and no tile beyond the edge.  No workgroup returns early
// (but for the last group of a job), so the workgroups of an XCD keep starting and finishing together and
// keep sharing their A and B row blocks in L2.  (The first version enumerated rectangular macro-tiles and
// let the workgroups of dead tiles return at once; every one of them shifted a CU by a tile time for the
// rest of the launch: -2.3 % at N = 65536, n = 16.)
__device__ __forceinline__ bool gram_locate(const GramArgs& args, GramJobDev& J, int& ti, int& tj) {
    const int64_t bid = blockIdx.x;
    // constant indices only: a dynamic index into the kernel-argument struct would copy it to scratch
    J = args.job[0];
#pragma unroll
    for (int k = 1; k < QK_GRAM_MAX_JOBS; ++k)
        if (k < args.n_jobs && bid >= args.job[k].block_begin) J = args.job[k];
    const int64_t lb = bid - J.block_begin;
    const int xcd = (int)(lb & 7);
    const int64_t seq = lb >> 3;                       // position in this XCD's stream
    const int lr = args.lr, lc = args.lc;
    const int64_t group = (seq >> (lr + lc)) * 8 + xcd;
    int64_t t = (group << (lr + lc)) + (seq & ((1 << (lr + lc)) - 1));   // index in the job's list of live tiles
    if (t >= J.live) return false;
    const int br = 1 << lr, TMt = J.tiles_m, TNt = J.tiles_n;
    if (!(J.flags & QK_GRAM_SYMMETRIC)) {
        const int64_t per_band = (int64_t)br * TNt;
        const int band = (int)(t / per_band);
        const int idx = (int)(t - band * per_band);
        const int rb = min(br, TMt - band * br);
        tj = idx / rb;
        ti = band * br + (idx - tj * rb);
        return true;
    }
    // symmetric (square tiles): band m holds, for the columns c >= m * br, the rows m * br .. min(c, m * br + rb - 1)
    int m = 0, rb = 0;
    for (;; ++m) {
        const int first = m * br;
        if (first >= TMt) return false;
        rb = min(br, TMt - first);
        const int64_t cnt = (int64_t)rb * (TNt - first) - rb * (rb - 1) / 2;
        if (t < cnt) break;
        t -= cnt;
    }
    const int ragged = rb * (rb - 1) / 2;              // tiles of the first rb - 1 columns (1, 2, ... rows)
    int k, r;
    if (t < ragged) {
        k = 0;
        while ((k + 1) * (k + 2) / 2 <= (int)t) ++k;
        r = (int)t - k * (k + 1) / 2;
    } else {
        const int64_t u = t - ragged;
        k = rb - 1 + (int)(u / rb);
        r = (int)(u % rb);
    }
    ti = m * br + r;
    tj = m * br + k;
    return true;
}

// |.|^2 and the stores.  f64 16x16x4 result layout: col = lane & 15, row = (lane >> 4) + 4 * reg.
template <int TM, int TN, bool K3M, int NACC>
__device__ __forceinline__ void gram_epilogue(v4d (&acc)[NACC][TM][TN], const GramJobDev& J, int64_t wrow0,
                                              int64_t wcol0, int lane, bool sym) {
    double* __restrict__ Kp = J.K;
    double* __restrict__ KTp = J.KT;
    const int64_t ldk = J.ldk, ldkt = J.ldkt, ma = J.ma, nb = J.nb;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = wrow0 + m * 16 + (lane >> 4) + 4 * r;
                const int64_t j = wcol0 + n * 16 + (lane & 15);
                if (i >= ma || j >= nb) continue;
                double re, im;
                if constexpr (K3M) {
                    re = acc[0][m][n][r] + acc[1][m][n][r];
                    im = acc[2][m][n][r] + (acc[0][m][n][r] - acc[1][m][n][r]);
                } else {
                    re = acc[0][m][n][r];
                    im = acc[1][m][n][r];
                }
                double v = fma(re, re, im * im);
                if (sym) {
                    if (i > j) continue;      // below the diagonal: written by the mirror store
                    if (i == j) v = 1.0;      // identical statevectors stay exactly 1
                }
                Kp[i * ldk + j] = v;
                if (sym && i != j) Kp[j * ldk + i] = v;
                if (KTp) KTp[j * ldkt + i] = v;
            }
}

// WAVES_M x WAVES_N waves, each TM x TN MFMA tiles of 16x16; BK amplitudes per barrier.
template <int WAVES_M, int WAVES_N, int TM, int TN, int BK, bool K3M>
struct GramCfg {
    static constexpr int THREADS = 64 * WAVES_M * WAVES_N;
    static constexpr int BM = WAVES_M * TM * 16;
    static constexpr int BN = WAVES_N * TN * 16;
    static constexpr int LDA = BM + 1;  // padded k-row lengths (16-B elements)
    static constexpr int LDB = BN + 1;
    static constexpr int BUF = BK * (LDA + LDB);
    static constexpr int PA = BM * BK / THREADS;  // staged elements per thread and k-tile
    static constexpr int PB = BN * BK / THREADS;
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF * sizeof(double2);
    static_assert(BM * BK % THREADS == 0 && BN * BK % THREADS == 0, "staging must divide evenly");
    static_assert(BK % 8 == 0, "BK is a multiple of 8 amplitudes (128-byte runs)");
};

template <int WAVES_M, int WAVES_N, int TM, int TN, int BK, bool K3M>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N) / 4)
gram_kernel(GramArgs args) {
    using C = GramCfg<WAVES_M, WAVES_N, TM, TN, BK, K3M>;
    constexpr int NACC = K3M ? 3 : 2;
    extern __shared__ v2d smem[];  // [2 buffers][A: BK x LDA | B: BK x LDB]
    GramJobDev J;
    int ti, tj;
    if (!gram_locate(args, J, ti, tj)) return;
    const bool sym = (J.flags & QK_GRAM_SYMMETRIC) != 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int64_t D = args.D;
    const int64_t row0 = (int64_t)ti * C::BM, col0 = (int64_t)tj * C::BN;
    const int64_t ma = J.ma, nb = J.nb;
    const v2d* __restrict__ Ag = (const v2d*)J.A;
    const v2d* __restrict__ Bg = (const v2d*)J.B;

    // ---- global -> register staging map: BK consecutive lanes cover BK*16 contiguous bytes of a row.
    // Rows past the end are clamped: their values land in LDS rows whose results are never stored.
    const bool k_in_range = (tid % BK) < D;  // D < BK only for circuits under 3 qubits
    const int ld_k = k_in_range ? (tid % BK) : 0;
    const v2d* gA[C::PA];
    const v2d* gB[C::PB];
    int sA[C::PA], sB[C::PB];
#pragma unroll
    for (int p = 0; p < C::PA; ++p) {
        const int r = (tid + p * C::THREADS) / BK;
        gA[p] = Ag + min(row0 + r, ma - 1) * D + ld_k;
        sA[p] = (tid % BK) * C::LDA + r;
    }
#pragma unroll
    for (int p = 0; p < C::PB; ++p) {
        const int r = (tid + p * C::THREADS) / BK;
        gB[p] = Bg + min(col0 + r, nb - 1) * D + ld_k;
        sB[p] = BK * C::LDA + (tid % BK) * C::LDB + r;
    }

    v4d acc[NACC][TM][TN];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[c][m][n] = (v4d){0, 0, 0, 0};

    const int64_t ktiles = (D + BK - 1) / BK;
    const v2d zero2 = (v2d){0.0, 0.0};
    v2d pa[C::PA], pb[C::PB];
    // prologue: tile 0 -> LDS buffer 0
#pragma unroll
    for (int p = 0; p < C::PA; ++p) pa[p] = k_in_range ? gA[p][0] : zero2;
#pragma unroll
    for (int p = 0; p < C::PB; ++p) pb[p] = k_in_range ? gB[p][0] : zero2;
#pragma unroll
    for (int p = 0; p < C::PA; ++p) smem[sA[p]] = pa[p];
#pragma unroll
    for (int p = 0; p < C::PB; ++p) smem[sB[p]] = pb[p];
    __syncthreads();

    const int frag_k = lane >> 4, frag_r = lane & 15;
    const int a_off = frag_k * C::LDA + wm * (TM * 16) + frag_r;
    const int b_off = BK * C::LDA + frag_k * C::LDB + wn * (TN * 16) + frag_r;
    constexpr int KS = BK / 4;  // MFMA k-steps per k-tile
    if constexpr (K3M) {
        // Software pipeline: the fragments of k-step s+1 are read from LDS after the P1/P2 MFMAs of
        // step s have been issued (their a/b registers are dead by then) and before its P3 MFMAs,
        // so the LDS latency hides behind TM*TN MFMAs (64 cycles each) -- also across the barrier.
        v2d a[TM], b[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) a[m] = smem[a_off + m * 16];
#pragma unroll
        for (int n = 0; n < TN; ++n) b[n] = smem[b_off + n * 16];
        for (int64_t kt = 0; kt < ktiles; ++kt) {
            const int cur = (int)(kt & 1);
            const int64_t off = min(kt + 1, ktiles - 1) * BK;  // last tile re-reads itself (branch-free)
#pragma unroll
            for (int p = 0; p < C::PA; ++p) pa[p] = gA[p][off];
#pragma unroll
            for (int p = 0; p < C::PB; ++p) pb[p] = gB[p][off];
            __builtin_amdgcn_sched_barrier(0);  // keep the loads at the top of the k-tile
            const v2d* S = smem + cur * C::BUF;
            v2d* Sn = smem + (cur ^ 1) * C::BUF;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                double sa[TM], db[TN];
#pragma unroll
                for (int m = 0; m < TM; ++m) sa[m] = a[m].x + a[m].y;
#pragma unroll
                for (int n = 0; n < TN; ++n) db[n] = b[n].y - b[n].x;
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[0][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b[n].x, acc[0][m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[1][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].y, b[n].y, acc[1][m][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk < KS - 1) {
#pragma unroll
                    for (int m = 0; m < TM; ++m) a[m] = S[(kk + 1) * 4 * C::LDA + a_off + m * 16];
#pragma unroll
                    for (int n = 0; n < TN; ++n) b[n] = S[(kk + 1) * 4 * C::LDB + b_off + n * 16];
                } else {
#pragma unroll
                    for (int p = 0; p < C::PA; ++p) Sn[sA[p]] = pa[p];
#pragma unroll
                    for (int p = 0; p < C::PB; ++p) Sn[sB[p]] = pb[p];
                    __syncthreads();
#pragma unroll
                    for (int m = 0; m < TM; ++m) a[m] = Sn[a_off + m * 16];
#pragma unroll
                    for (int n = 0; n < TN; ++n) b[n] = Sn[b_off + n * 16];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[2][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[m], db[n], acc[2][m][n], 0, 0, 0);
            }
        }
    } else {
        for (int64_t kt = 0; kt < ktiles; ++kt) {
            const int cur = (int)(kt & 1);
            // prefetch the next k-tile into registers (the last iteration re-reads its own
            // tile: branch-free, and nothing reads the buffer it is written to afterwards)
            const int64_t off = min(kt + 1, ktiles - 1) * BK;
#pragma unroll
            for (int p = 0; p < C::PA; ++p) pa[p] = gA[p][off];
#pragma unroll
            for (int p = 0; p < C::PB; ++p) pb[p] = gB[p][off];
            // keep the loads up here: hipcc otherwise sinks them to their first use (the LDS
            // writes below) and every k-tile waits out a full HBM/L2 round trip
            __builtin_amdgcn_sched_barrier(0);
            const v2d* S = smem + cur * C::BUF;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                v2d a[TM], b[TN];
#pragma unroll
                for (int m = 0; m < TM; ++m) a[m] = S[kk * 4 * C::LDA + a_off + m * 16];
#pragma unroll
                for (int n = 0; n < TN; ++n) b[n] = S[kk * 4 * C::LDB + b_off + n * 16];
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[0][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b[n].x, acc[0][m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[1][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b[n].y, acc[1][m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[0][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].y, b[n].y, acc[0][m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < TM; ++m) {
                    const double nai = -a[m].y;
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[1][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, b[n].x, acc[1][m][n], 0, 0, 0);
                }
            }
            v2d* Sn = smem + (cur ^ 1) * C::BUF;
#pragma unroll
            for (int p = 0; p < C::PA; ++p) Sn[sA[p]] = pa[p];
#pragma unroll
            for (int p = 0; p < C::PB; ++p) Sn[sB[p]] = pb[p];
            __syncthreads();
        }
    }

    gram_epilogue<TM, TN, K3M>(acc, J, row0 + wm * (TM * 16), col0 + wn * (TN * 16), lane, sym);
}

// ---- LDS-DMA variant ---------------------------------------------------------------
// Same tiling and arithmetic, but the A/B tiles go HBM/L2 -> LDS directly with
// global_load_lds_dwordx4 (no staging registers, no ds_write, no VGPR write-back).
// The DMA writes LDS linearly (wave-uniform base + lane * 16 B), so the LDS image is
// row-major [row][16 slots] with 256-byte rows, and the bank swizzle lives in the SOURCE
// address: slot s of row r holds amplitude k = s ^ (r & 15).  One wave instruction fills
// 4 rows from 4 x 256 contiguous bytes of Psi; a fragment read (16 rows x one k-slot per
// 16-lane group) then touches 16 different slots = all 64 banks once: conflict free.
template <int WAVES_M, int WAVES_N, int TM, int TN, bool K3M>
struct GramDmaCfg {
    static constexpr int BK = 16;
    static constexpr int WAVES = WAVES_M * WAVES_N;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int BM = WAVES_M * TM * 16;
    static constexpr int BN = WAVES_N * TN * 16;
    static constexpr int BUF = (BM + BN) * BK;          // elements (16 B) per buffer
    static constexpr int NA = BM / 4 / WAVES;           // DMA instructions per wave and k-tile
    static constexpr int NB = BN / 4 / WAVES;
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF * 16;
    static_assert(BM % (4 * WAVES) == 0 && BN % (4 * WAVES) == 0, "rows must split evenly over the waves");
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int WAVES_M, int WAVES_N, int TM, int TN, bool K3M>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N) / 4 < 2 ? 2 : (WAVES_M * WAVES_N) / 4)
gram_dma_kernel(GramArgs args) {
    using C = GramDmaCfg<WAVES_M, WAVES_N, TM, TN, K3M>;
    constexpr int NACC = K3M ? 3 : 2;
    constexpr int KS = 4;  // BK / 4
    extern __shared__ v2d smem[];  // [2 buffers][A: BM x 16 | B: BN x 16]
    GramJobDev J;
    int ti, tj;
    if (!gram_locate(args, J, ti, tj)) return;
    const bool sym = (J.flags & QK_GRAM_SYMMETRIC) != 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int64_t D = args.D;
    const int64_t row0 = (int64_t)ti * C::BM, col0 = (int64_t)tj * C::BN;
    const int64_t ma = J.ma, nb = J.nb;
    const v2d* __restrict__ Ag = (const v2d*)J.A;
    const v2d* __restrict__ Bg = (const v2d*)J.B;

    // DMA source pointers (per lane) and LDS destinations (per wave) of this wave's instructions
    const v2d* gA[C::NA];
    const v2d* gB[C::NB];
#pragma unroll
    for (int j = 0; j < C::NA; ++j) {
        const int row = 4 * (wave + C::WAVES * j) + (lane >> 4);
        gA[j] = Ag + min(row0 + row, ma - 1) * D + ((lane & 15) ^ (row & 15));
    }
#pragma unroll
    for (int j = 0; j < C::NB; ++j) {
        const int row = 4 * (wave + C::WAVES * j) + (lane >> 4);
        gB[j] = Bg + min(col0 + row, nb - 1) * D + ((lane & 15) ^ (row & 15));
    }
    // DMA instruction e of this wave (A first, then B); `part` < 0 issues all of them, otherwise only
    // those assigned to k-step `part`: spread over the first KS-1 k-steps so that the ~100-cycle issue
    // cost of each lands between MFMAs instead of in front of them, and the last ones still have a
    // quarter k-tile to land before the barrier.
    auto issue_dma = [&](int buf, int64_t off, int part) {
        v2d* base = smem + buf * C::BUF;
#pragma unroll
        for (int e = 0; e < C::NA + C::NB; ++e) {
            if (part >= 0 && (e % 3) != part) continue;
            if (e < C::NA) {
                const int j = e < C::NA ? e : 0;
                __builtin_amdgcn_global_load_lds((gptr_t)(gA[j] + off), (lptr_t)(base + (wave + C::WAVES * j) * 64), 16, 0, 0);
            } else {
                const int j = e >= C::NA ? e - C::NA : 0;
                __builtin_amdgcn_global_load_lds((gptr_t)(gB[j] + off),
                                                 (lptr_t)(base + C::BM * 16 + (wave + C::WAVES * j) * 64), 16, 0, 0);
            }
        }
    };

    v4d acc[NACC][TM][TN];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[c][m][n] = (v4d){0, 0, 0, 0};

    const int64_t ktiles = D / C::BK;
    issue_dma(0, 0, -1);
    __syncthreads();

    // fragment addresses: element (row * 16 + ((4 kk + lane>>4) ^ (lane & 15)))
    const int frag_k = lane >> 4, frag_r = lane & 15;
    const int a_row = (wm * (TM * 16) + frag_r) * 16;
    const int b_row = C::BM * 16 + (wn * (TN * 16) + frag_r) * 16;
    int xs[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) xs[kk] = (4 * kk + frag_k) ^ frag_r;

    v2d a[TM], b[TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) a[m] = smem[a_row + m * 256 + xs[0]];
#pragma unroll
    for (int n = 0; n < TN; ++n) b[n] = smem[b_row + n * 256 + xs[0]];
    for (int64_t kt = 0; kt < ktiles; ++kt) {
        const int cur = (int)(kt & 1);
        // next k-tile straight into the other buffer (nobody reads it during this k-tile);
        // the last iteration re-reads its own tile: branch-free and harmless
        const int64_t noff = min(kt + 1, ktiles - 1) * C::BK;
        const v2d* S = smem + cur * C::BUF;
        const v2d* Sn = smem + (cur ^ 1) * C::BUF;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if constexpr (K3M) {
                // software pipeline: fragments of the next k-step are read after P1/P2 have been
                // issued (a, b are dead then) and before P3, also across the barrier
                double sa[TM], db[TN];
#pragma unroll
                for (int m = 0; m < TM; ++m) sa[m] = a[m].x + a[m].y;
#pragma unroll
                for (int n = 0; n < TN; ++n) db[n] = b[n].y - b[n].x;
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[0][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b[n].x, acc[0][m][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk < 3) issue_dma(cur ^ 1, noff, kk);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[1][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].y, b[n].y, acc[1][m][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk < KS - 1) {
#pragma unroll
                    for (int m = 0; m < TM; ++m) a[m] = S[a_row + m * 256 + xs[kk + 1 < KS ? kk + 1 : 0]];
#pragma unroll
                    for (int n = 0; n < TN; ++n) b[n] = S[b_row + n * 256 + xs[kk + 1 < KS ? kk + 1 : 0]];
                } else {
                    __syncthreads();
#pragma unroll
                    for (int m = 0; m < TM; ++m) a[m] = Sn[a_row + m * 256 + xs[0]];
#pragma unroll
                    for (int n = 0; n < TN; ++n) b[n] = Sn[b_row + n * 256 + xs[0]];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[2][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[m], db[n], acc[2][m][n], 0, 0, 0);
            } else {
                if (kk < 3) issue_dma(cur ^ 1, noff, kk);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[0][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b[n].x, acc[0][m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[1][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b[n].y, acc[1][m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[0][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].y, b[n].y, acc[0][m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < TM; ++m) {
                    const double nai = -a[m].y;
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[1][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, b[n].x, acc[1][m][n], 0, 0, 0);
                }
                if (kk < KS - 1) {
#pragma unroll
                    for (int m = 0; m < TM; ++m) a[m] = S[a_row + m * 256 + xs[kk + 1 < KS ? kk + 1 : 0]];
#pragma unroll
                    for (int n = 0; n < TN; ++n) b[n] = S[b_row + n * 256 + xs[kk + 1 < KS ? kk + 1 : 0]];
                } else {
                    __syncthreads();
#pragma unroll
                    for (int m = 0; m < TM; ++m) a[m] = Sn[a_row + m * 256 + xs[0]];
#pragma unroll
                    for (int n = 0; n < TN; ++n) b[n] = Sn[b_row + n * 256 + xs[0]];
                }
            }
        }
    }
    gram_epilogue<TM, TN, K3M>(acc, J, row0 + wm * (TM * 16), col0 + wn * (TN * 16), lane, sym);
}

// ---- FP32 variant (optional precision of BASELINE.json north_star; tolerance 1e-5) -------------
// Psi rows are interleaved complex64.  Same LDS-DMA geometry with 16-byte units: one unit holds TWO
// amplitudes, a k-tile is 32 amplitudes, and one b128 fragment read feeds two k-steps of
// v_mfma_f32_16x16x4_f32 (which amplitude sits in which MFMA k-slot is free as long as A and B
// agree: lane group g of double-step kk takes the pair 4 kk + g).  3-multiplication product as
// above.  FP32 accumulation over 2^16 amplitudes would drift past 1e-5, so the three running
// accumulators only ever hold 1024 amplitudes: every 32 k-tiles they are folded into a second
// level (Re, Im) and cleared -- 64 second-level additions instead of 16384 first-level ones.
typedef float v4f __attribute__((ext_vector_type(4)));

// f32 16x16x4 result layout: col = lane & 15, row = 4 * (lane >> 4) + reg.
template <int TM, int TN>
__device__ __forceinline__ void gram32_epilogue(v4f (&re)[TM][TN], v4f (&im)[TM][TN], const GramJobDev& J, int64_t wrow0,
                                                int64_t wcol0, int lane, bool sym) {
    double* __restrict__ Kp = J.K;
    double* __restrict__ KTp = J.KT;
    const int64_t ldk = J.ldk, ldkt = J.ldkt, ma = J.ma, nb = J.nb;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = wrow0 + m * 16 + 4 * (lane >> 4) + r;
                const int64_t j = wcol0 + n * 16 + (lane & 15);
                if (i >= ma || j >= nb) continue;
                const double x = (double)re[m][n][r], y = (double)im[m][n][r];
                double v = fma(x, x, y * y);
                if (sym) {
                    if (i > j) continue;
                    if (i == j) v = 1.0;
                }
                Kp[i * ldk + j] = v;
                if (sym && i != j) Kp[j * ldk + i] = v;
                if (KTp) KTp[j * ldkt + i] = v;
            }
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N) / 4 < 2 ? 2 : (WAVES_M * WAVES_N) / 4)
gram32_dma_kernel(GramArgs args) {
    using C = GramDmaCfg<WAVES_M, WAVES_N, TM, TN, true>;   // BK = 16 units = 32 amplitudes
    constexpr int KS = 4;
    constexpr int FLUSH = 32;  // k-tiles per first-level accumulation run (1024 amplitudes)
    extern __shared__ v4f smem32[];
    GramJobDev J;
    int ti, tj;
    if (!gram_locate(args, J, ti, tj)) return;
    const bool sym = (J.flags & QK_GRAM_SYMMETRIC) != 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int64_t U = args.D / 2;   // 16-byte units per row
    const int64_t row0 = (int64_t)ti * C::BM, col0 = (int64_t)tj * C::BN;
    const int64_t ma = J.ma, nb = J.nb;
    const v4f* __restrict__ Ag = (const v4f*)J.A;
    const v4f* __restrict__ Bg = (const v4f*)J.B;

    const v4f* gA[C::NA];
    const v4f* gB[C::NB];
#pragma unroll
    for (int j = 0; j < C::NA; ++j) {
        const int row = 4 * (wave + C::WAVES * j) + (lane >> 4);
        gA[j] = Ag + min(row0 + row, ma - 1) * U + ((lane & 15) ^ (row & 15));
    }
#pragma unroll
    for (int j = 0; j < C::NB; ++j) {
        const int row = 4 * (wave + C::WAVES * j) + (lane >> 4);
        gB[j] = Bg + min(col0 + row, nb - 1) * U + ((lane & 15) ^ (row & 15));
    }
    auto issue_dma = [&](int buf, int64_t off, int part) {
        v4f* base = smem32 + buf * C::BUF;
#pragma unroll
        for (int e = 0; e < C::NA + C::NB; ++e) {
            if (part >= 0 && (e % 3) != part) continue;
            if (e < C::NA) {
                const int j = e < C::NA ? e : 0;
                __builtin_amdgcn_global_load_lds((gptr_t)(gA[j] + off), (lptr_t)(base + (wave + C::WAVES * j) * 64), 16, 0, 0);
            } else {
                const int j = e >= C::NA ? e - C::NA : 0;
                __builtin_amdgcn_global_load_lds((gptr_t)(gB[j] + off),
                                                 (lptr_t)(base + C::BM * 16 + (wave + C::WAVES * j) * 64), 16, 0, 0);
            }
        }
    };

    v4f acc[3][TM][TN], hre[TM][TN], him[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            acc[0][m][n] = acc[1][m][n] = acc[2][m][n] = (v4f){0, 0, 0, 0};
            hre[m][n] = him[m][n] = (v4f){0, 0, 0, 0};
        }
    auto flush = [&]() {
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                hre[m][n] += acc[0][m][n] + acc[1][m][n];
                him[m][n] += acc[2][m][n] + (acc[0][m][n] - acc[1][m][n]);
                acc[0][m][n] = acc[1][m][n] = acc[2][m][n] = (v4f){0, 0, 0, 0};
            }
    };

    const int64_t ktiles = U / C::BK;
    issue_dma(0, 0, -1);
    __syncthreads();

    const int frag_k = lane >> 4, frag_r = lane & 15;
    const int a_row = (wm * (TM * 16) + frag_r) * 16;
    const int b_row = C::BM * 16 + (wn * (TN * 16) + frag_r) * 16;
    int xs[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) xs[kk] = (4 * kk + frag_k) ^ frag_r;

    v4f a[TM], b[TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) a[m] = smem32[a_row + m * 256 + xs[0]];
#pragma unroll
    for (int n = 0; n < TN; ++n) b[n] = smem32[b_row + n * 256 + xs[0]];
    for (int64_t kt = 0; kt < ktiles; ++kt) {
        const int cur = (int)(kt & 1);
        const int64_t noff = min(kt + 1, ktiles - 1) * C::BK;
        const v4f* S = smem32 + cur * C::BUF;
        const v4f* Sn = smem32 + (cur ^ 1) * C::BUF;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            float sa[2][TM], db[2][TN];
#pragma unroll
            for (int m = 0; m < TM; ++m) { sa[0][m] = a[m].x + a[m].y; sa[1][m] = a[m].z + a[m].w; }
#pragma unroll
            for (int n = 0; n < TN; ++n) { db[0][n] = b[n].y - b[n].x; db[1][n] = b[n].w - b[n].z; }
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[0][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b[n].x, acc[0][m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk < 3) issue_dma(cur ^ 1, noff, kk);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[1][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, b[n].y, acc[1][m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[2][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0][m], db[0][n], acc[2][m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[0][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].z, b[n].z, acc[0][m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[1][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].w, b[n].w, acc[1][m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // fragments of the next step: a, b are dead from here (the last products use sa / db)
            if (kk < KS - 1) {
#pragma unroll
                for (int m = 0; m < TM; ++m) a[m] = S[a_row + m * 256 + xs[kk + 1 < KS ? kk + 1 : 0]];
#pragma unroll
                for (int n = 0; n < TN; ++n) b[n] = S[b_row + n * 256 + xs[kk + 1 < KS ? kk + 1 : 0]];
            } else {
                __syncthreads();
#pragma unroll
                for (int m = 0; m < TM; ++m) a[m] = Sn[a_row + m * 256 + xs[0]];
#pragma unroll
                for (int n = 0; n < TN; ++n) b[n] = Sn[b_row + n * 256 + xs[0]];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[2][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1][m], db[1][n], acc[2][m][n], 0, 0, 0);
        }
        if ((kt & (FLUSH - 1)) == FLUSH - 1) flush();
    }
    flush();
    gram32_epilogue<TM, TN>(hre, him, J, row0 + wm * (TM * 16), col0 + wn * (TN * 16), lane, sym);
}

// D < 32 (under 5 qubits): one thread per entry, FP32 products accumulated in order.
__global__ void __launch_bounds__(256) gram32_small_kernel(GramArgs args) {
    const GramJobDev& J = args.job[0];
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= J.ma * J.nb) return;
    const int64_t i = e / J.nb, j = e % J.nb;
    const bool sym = (J.flags & QK_GRAM_SYMMETRIC) != 0;
    // symmetric block: both (i, j) and (j, i) are computed from the ordered pair, so they agree bitwise
    const int64_t p = sym && i > j ? j : i, q = sym && i > j ? i : j;
    const float2* a = (const float2*)J.A + p * args.D;
    const float2* b = (const float2*)J.B + q * args.D;
    float re = 0.f, im = 0.f;
    for (int64_t k = 0; k < args.D; ++k) {
        re = fmaf(a[k].x, b[k].x, fmaf(a[k].y, b[k].y, re));
        im = fmaf(a[k].x, b[k].y, fmaf(-a[k].y, b[k].x, im));
    }
    double v = fma((double)re, (double)re, (double)im * (double)im);
    if (sym && i == j) v = 1.0;
    J.K[i * J.ldk + j] = v;
    if (J.KT) J.KT[j * J.ldkt + i] = v;
}

// ---- variants ------------------------------------------------------------------
struct Variant {
    const char* name;
    int bm, bn, threads, min_d;
    int lr, lc;  // macro-tile shape (log2 block tiles): sized so that one XCD's resident workgroups form about one macro-tile
    size_t lds;
    const void* fn;
    void (*launch)(const GramArgs&, unsigned, hipStream_t);
};

template <int WAVES_M, int WAVES_N, int TM, int TN, int BK, bool K3M>
void launch_variant(const GramArgs& args, unsigned blocks, hipStream_t stream) {
    using C = GramCfg<WAVES_M, WAVES_N, TM, TN, BK, K3M>;
    hipLaunchKernelGGL((gram_kernel<WAVES_M, WAVES_N, TM, TN, BK, K3M>), dim3(blocks), dim3(C::THREADS), C::LDS_BYTES,
                       stream, args);
}

#define QK_VARIANT(NAME, WM, WN, TM, TN, BK, K3M)                                                          \
    Variant {                                                                                              \
        NAME, GramCfg<WM, WN, TM, TN, BK, K3M>::BM, GramCfg<WM, WN, TM, TN, BK, K3M>::BN,                  \
            GramCfg<WM, WN, TM, TN, BK, K3M>::THREADS, 1, 2, 3, GramCfg<WM, WN, TM, TN, BK, K3M>::LDS_BYTES, \
            (const void*)gram_kernel<WM, WN, TM, TN, BK, K3M>, launch_variant<WM, WN, TM, TN, BK, K3M>     \
    }

template <int WAVES_M, int WAVES_N, int TM, int TN, bool K3M>
void launch_dma_variant(const GramArgs& args, unsigned blocks, hipStream_t stream) {
    using C = GramDmaCfg<WAVES_M, WAVES_N, TM, TN, K3M>;
    hipLaunchKernelGGL((gram_dma_kernel<WAVES_M, WAVES_N, TM, TN, K3M>), dim3(blocks), dim3(C::THREADS), C::LDS_BYTES,
                       stream, args);
}

#define QK_DMA_VARIANT(NAME, WM, WN, TM, TN, K3M, LR, LC)                                                  \
    Variant {                                                                                              \
        NAME, GramDmaCfg<WM, WN, TM, TN, K3M>::BM, GramDmaCfg<WM, WN, TM, TN, K3M>::BN,                    \
            GramDmaCfg<WM, WN, TM, TN, K3M>::THREADS, 16, LR, LC, GramDmaCfg<WM, WN, TM, TN, K3M>::LDS_BYTES, \
            (const void*)gram_dma_kernel<WM, WN, TM, TN, K3M>, launch_dma_variant<WM, WN, TM, TN, K3M>     \
    }

static const Variant kVariants[] = {
    QK_VARIANT("4m_128x128_w8_bk8", 2, 4, 4, 2, 8, false),
    QK_VARIANT("4m_128x128_w8_bk16", 2, 4, 4, 2, 16, false),
    QK_DMA_VARIANT("dma3m_128x128_w8", 2, 4, 4, 2, true, 2, 3),
    QK_DMA_VARIANT("dma4m_128x128_w8", 2, 4, 4, 2, false, 2, 3),
    QK_DMA_VARIANT("dma3m_64x64_w4", 2, 2, 2, 2, true, 2, 3),    // two independent workgroups per CU
    QK_DMA_VARIANT("dma3m_64x64_w4_g64", 2, 2, 2, 2, true, 3, 3),  // ... whose 64 tiles per XCD form ONE 8 x 8 group
    QK_DMA_VARIANT("dma3m_64x64_w4_g4x16", 2, 2, 2, 2, true, 2, 4),
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
// Default, from in-process A/B runs (tools/gram_ab.py, tools/shard_ab.py): dma3m_64x64_w4_g64 for every
// shape.  Two independent 4-wave workgroups per CU: one keeps the matrix pipe busy while the other sits at
// its barrier, and small problems split into 4x as many tiles.  With the packed tile schedule (gram_locate)
// it also wins at the headline size (N = 65536, n = 16: 12.50 s against 13.05 s for dma3m_128x128_w8 on
// the same box; under the older macro-tile schedule with its dead tiles the 128 x 128 tiles were 1.3 %
// ahead there).  N = 4096, n = 8: +34 % over 128 x 128.
constexpr int kDefaultVariant = 5;   // dma3m_64x64_w4_g64

static std::atomic<int> g_variant{-1};       // forced by QK_GRAM_VARIANT / qk_gram_set_variant; -1 = the default
static std::atomic<bool> g_env_read{false};
static DeviceOnce g_attr_set[kNumVariants];

static int pick_variant(const qk_gram_job* jobs, int n_jobs, int64_t D) {
    if (!g_env_read.exchange(true)) {
        if (const char* e = std::getenv("QK_GRAM_VARIANT"))
            for (int v = 0; v < kNumVariants; ++v)
                if (std::strcmp(e, kVariants[v].name) == 0) g_variant = v;
    }
    const int forced = g_variant.load();
    if (forced >= 0) return forced;
    (void)jobs; (void)n_jobs; (void)D;
    return kDefaultVariant;
}

}  // namespace qk

// Development hooks (not in the public header).  qk_gram_set_variant: force a kernel variant by name for
// A/B timing ("auto" or NULL: back to the per-call choice).  qk_gram_variant_for: the variant a block
// of `rows` rows would run on (bench.py labels its roofline line with it).
extern "C" int qk_gram_set_variant(const char* name) {
    using namespace qk;
    g_env_read = true;
    if (!name || std::strcmp(name, "auto") == 0) {
        g_variant = -1;
        return QK_OK;
    }
    for (int v = 0; v < kNumVariants; ++v)
        if (std::strcmp(name, kVariants[v].name) == 0) {
            g_variant = v;
            return QK_OK;
        }
    return fail(QK_ERR_INVALID, "qk_gram_set_variant: unknown variant");
}

extern "C" const char* qk_gram_variant_for(int64_t rows, int64_t D) {
    using namespace qk;
    qk_gram_job j;
    std::memset(&j, 0, sizeof(j));
    j.ma = j.nb = rows;
    int vi = pick_variant(&j, 1, D);
    if (D < kVariants[vi].min_d) vi = 0;
    return kVariants[vi].name;
}

namespace qk {
// validate the jobs and lay them out on the tile schedule of (bm x bn) block tiles
static int build_gram_args(const qk_gram_job* jobs, int n_jobs, int64_t D, int bm, int bn, int lr, int lc, const char* who,
                           GramArgs& args, int64_t& blocks) {
    if (!jobs || n_jobs < 1 || n_jobs > QK_GRAM_MAX_JOBS) return fail(QK_ERR_INVALID, std::string(who) + ": need 1..16 jobs");
    if (D < 1 || (D & (D - 1)) != 0) return fail(QK_ERR_INVALID, std::string(who) + ": D must be a power of two");
    std::memset(&args, 0, sizeof(args));
    args.D = D;
    args.lr = lr;
    args.lc = lc;
    blocks = 0;
    int nj = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const qk_gram_job& j = jobs[k];
        if (j.ma < 0 || j.nb < 0) return fail(QK_ERR_INVALID, std::string(who) + ": negative block size");
        if (j.ma == 0 || j.nb == 0) continue;
        if (!j.A || !j.B || !j.K) return fail(QK_ERR_INVALID, std::string(who) + ": null pointer in job");
        if (j.ldk < j.nb || (j.KT && j.ldkt < j.ma))
            return fail(QK_ERR_INVALID, std::string(who) + ": leading dimension smaller than the block");
        const bool sym = (j.flags & QK_GRAM_SYMMETRIC) != 0;
        if (sym && (j.ma != j.nb || j.A != j.B))
            return fail(QK_ERR_INVALID, std::string(who) + ": a symmetric job needs A == B and ma == nb");
        GramJobDev& d = args.job[nj++];
        d.A = (const double2*)j.A;
        d.B = (const double2*)j.B;
        d.K = j.K;
        d.KT = j.KT;
        d.ma = j.ma; d.nb = j.nb; d.ldk = j.ldk; d.ldkt = j.ldkt;
        d.flags = j.flags;
        d.tiles_m = (int32_t)((j.ma + bm - 1) / bm);
        d.tiles_n = (int32_t)((j.nb + bn - 1) / bn);
        d.block_begin = blocks;
        if (sym && bm != bn) return fail(QK_ERR_INVALID, std::string(who) + ": symmetric jobs need square tiles");
        const int br = 1 << lr;
        if (sym) {
            d.live = 0;
            for (int first = 0; first < d.tiles_m; first += br) {
                const int rb = std::min(br, d.tiles_m - first);
                d.live += (int64_t)rb * (d.tiles_n - first) - rb * (rb - 1) / 2;
            }
        } else {
            d.live = (int64_t)d.tiles_m * d.tiles_n;
        }
        const int64_t per_group = (int64_t)1 << (lr + lc);
        const int64_t groups = (d.live + per_group - 1) / per_group;
        blocks += ((groups + 7) / 8) * 8 * per_group;  // whole rounds of 8 groups, one per XCD label
    }
    args.n_jobs = nj;
    if (blocks > 0x7fffffffLL) return fail(QK_ERR_UNSUPPORTED, std::string(who) + ": too many tiles; split the call");
    return QK_OK;
}
}  // namespace qk

extern "C" int qk_gram_jobs_dev(const qk_gram_job* jobs, int n_jobs, int64_t D, void* stream) {
    using namespace qk;
    int vi = pick_variant(jobs, n_jobs, D);
    if (D < kVariants[vi].min_d) vi = 0;  // LDS-DMA tiles need 16 amplitudes per k-tile; tiny circuits use variant 0
    const Variant& V = kVariants[vi];
    GramArgs args;
    int64_t blocks = 0;
    const int rc = build_gram_args(jobs, n_jobs, D, V.bm, V.bn, V.lr, V.lc, "qk_gram_jobs_dev", args, blocks);
    if (rc) return rc;
    if (args.n_jobs == 0) return QK_OK;
    if (g_attr_set[vi].pending()) {
        QK_HIP(hipFuncSetAttribute(V.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)V.lds));
        g_attr_set[vi].done();
    }
    V.launch(args, (unsigned)blocks, (hipStream_t)stream);
    QK_HIP(hipGetLastError());
    return QK_OK;
}

namespace qk {
template <int WM, int WN, int TM, int TN>
static int launch_gram32(const qk_gram_job* jobs, int n_jobs, int64_t D, hipStream_t stream) {
    using C = GramDmaCfg<WM, WN, TM, TN, true>;
    GramArgs args;
    int64_t blocks = 0;
    const int rc = build_gram_args(jobs, n_jobs, D, C::BM, C::BN, C::BM == 64 ? 3 : 2, 3, "qk_gram32_jobs_dev", args, blocks);   // 64 x 64: two workgroups per CU = 8 x 8 groups
    if (rc) return rc;
    if (args.n_jobs == 0) return QK_OK;
    static DeviceOnce attr_set;
    if (attr_set.pending()) {
        QK_HIP(hipFuncSetAttribute((const void*)gram32_dma_kernel<WM, WN, TM, TN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)C::LDS_BYTES));
        attr_set.done();
    }
    hipLaunchKernelGGL((gram32_dma_kernel<WM, WN, TM, TN>), dim3((unsigned)blocks), dim3(C::THREADS), C::LDS_BYTES, stream, args);
    QK_HIP(hipGetLastError());
    return QK_OK;
}
static std::atomic<int> g_tile32{0};   // 0 = choose per call, 64 / 128 = forced (QK_GRAM32_TILE, A/B runs)
}  // namespace qk

extern "C" int qk_gram32_jobs_dev(const qk_gram_job* jobs, int n_jobs, int64_t D, void* stream) {
    using namespace qk;
    if (D >= 1 && D < 32 && (D & (D - 1)) == 0) {   // under 5 qubits: one launch per job, one thread per entry
        GramArgs args;
        int64_t blocks = 0;
        const int rc = build_gram_args(jobs, n_jobs, D, 128, 128, 2, 3, "qk_gram32_jobs_dev", args, blocks);
        if (rc) return rc;
        for (int k = 0; k < args.n_jobs; ++k) {
            GramArgs one;
            std::memset(&one, 0, sizeof(one));
            one.D = D;
            one.n_jobs = 1;
            one.job[0] = args.job[k];
            const int64_t entries = one.job[0].ma * one.job[0].nb;
            if ((entries + 255) / 256 > 0x7fffffffLL) return fail(QK_ERR_UNSUPPORTED, "qk_gram32_jobs_dev: block too large");
            hipLaunchKernelGGL(gram32_small_kernel, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, (hipStream_t)stream, one);
        }
        QK_HIP(hipGetLastError());
        return QK_OK;
    }
    static std::atomic<bool> env_read{false};
    if (!env_read.exchange(true)) {
        if (const char* e = std::getenv("QK_GRAM32_TILE")) g_tile32 = std::atoi(e);
    }
    const bool big = g_tile32.load() == 128;   // 64 x 64 tiles measured faster at every size (168 vs 166 TFLOP/s at N = 65536, n = 16)
    return big ? launch_gram32<2, 4, 4, 2>(jobs, n_jobs, D, (hipStream_t)stream)
               : launch_gram32<2, 2, 2, 2>(jobs, n_jobs, D, (hipStream_t)stream);
}

extern "C" int qk_gram_dev(const void* psi_x, int64_t N, const void* psi_y, int64_t M, int64_t D,
                           double* K_dev, int64_t ldk, void* stream) {
    qk_gram_job j;
    std::memset(&j, 0, sizeof(j));
    j.A = psi_x;
    j.B = psi_y ? psi_y : psi_x;
    j.K = K_dev;
    j.KT = nullptr;
    j.ma = N;
    j.nb = psi_y ? M : N;
    j.ldk = ldk;
    j.flags = psi_y ? 0 : QK_GRAM_SYMMETRIC;
    return qk_gram_jobs_dev(&j, 1, D, stream);
}

extern "C" int qk_gram32_dev(const void* psi_x, int64_t N, const void* psi_y, int64_t M, int64_t D,
                             double* K_dev, int64_t ldk, void* stream) {
    qk_gram_job j;
    std::memset(&j, 0, sizeof(j));
    j.A = psi_x;
    j.B = psi_y ? psi_y : psi_x;
    j.K = K_dev;
    j.KT = nullptr;
    j.ma = N;
    j.nb = psi_y ? M : N;
    j.ldk = ldk;
    j.flags = psi_y ? 0 : QK_GRAM_SYMMETRIC;
    return qk_gram32_jobs_dev(&j, 1, D, stream);
}
