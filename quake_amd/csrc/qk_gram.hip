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
// The shipped kernels (gram_ring_kernel<3> for rows of up to 1024 amplitudes, gram_ring_kernel<5> above; 64 x 64
// block tiles, four waves): A and B tiles go HBM/L2 -> LDS directly with global_load_lds (no staging registers)
// in stages of 8 amplitudes through a ring of 3 or 5 buffers, the barrier of a stage waiting for the next stage
// only (partial vmcnt), and the k-loop holds no vector-ALU instruction beside the four fragment sums of the
// 3-multiplication product -- the FP64 MFMA and the FP64 vector ALU share their multipliers, every other VALU
// instruction there is lost matrix time (profiles/r02_gram_ablation.md).  Two more kernels are kept
// for the shapes the rings do not take: gram_dma_kernel (the round-1 kernel: k-tiles of 16 amplitudes, double
// buffered, 256-byte LDS rows with slot s of row r holding amplitude s ^ (r & 15); 64-bit addressing) for rows
// beyond 2^22 amplitudes, and the register-staged gram_kernel (k-major LDS image with one element of padding per
// row) for circuits under 3 qubits.  All of them give the same bits.  (The kernels that only A/B runs used --
// 128 x 128 tiles, the "lean" double-buffered loop, other macro-tile shapes -- and the -DQK_GRAM_ABLATE measurement
// builds left the source after round 2; profiles/r02_gram_ablation.md names the commit that holds them.)
//
// Tile schedule (gram_locate): the tiles that have to be computed are listed band by band (8 tile rows),
// column-major inside a band, and cut into groups of 64 consecutive tiles -- an 8 x 8 macro-tile without dead
// tiles -- dealt round-robin to the 8 XCD labels (block id % 8): the 64 workgroups resident on an XCD share 8 A
// and 8 B row blocks in its L2; symmetric jobs list only tiles on or above the diagonal.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

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
    int64_t tile_begin;        // position of this job's first tile in the launch's list of live tiles
    int64_t live;              // number of tiles that have to be computed
};

struct GramArgs {
    GramJobDev job[QK_GRAM_MAX_JOBS];
    int n_jobs;
    uint32_t total_live;   // live tiles of all jobs together: ONE list, cut into groups without regard to job boundaries
    int lr, lc;  // log2 of the macro-tile shape in block tiles (rows, columns): 4 x 8 for 128 x 128 tiles
    int64_t D;  // amplitudes per row
    int64_t ld; // row stride of A and B in amplitudes (>= D)
    unsigned int* sync;   // ring kernel, long rows: one arrival counter per cohort of 32 workgroups (null: no pacing)
    int sync_shift;       // log2 of the trips of the stage loop (NST stages each) between two meetings of a cohort
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
//
// Everything here is wave-uniform and written for the scalar unit: 32-bit unsigned arithmetic (the host refuses
// launches of 2^31 positions or more), no 64-bit compares and no divisions -- the scalar unit has neither, and
// what the compiler moves to the vector ALU instead waits there behind the MFMAs of the CU's older waves: the
// first version of this function (int64, t / per_band, u / rb, a 16-way select over the job records) took 26 us
// of an 88 us tile slot at 8 qubits (timeline build, tools/gram_timeline.py).
__device__ __forceinline__ uint32_t gram_div_small(uint32_t x, uint32_t d) {   // d = 1 .. 7: constant divisors
    switch (d) {
        case 1: return x;
        case 2: return x >> 1;
        case 3: return x / 3u;
        case 4: return x >> 2;
        case 5: return x / 5u;
        case 6: return x / 6u;
        default: return x / 7u;
    }
}

// Position `lb` of the launch (block id) -> index in the launch's list of live tiles; false for padding positions.
// The jobs of a launch form ONE list (job after job): a group of 2^(lr+lc) tiles may straddle two jobs, but no
// round of 8 groups is left partial between jobs.  (Each job padded to whole rounds of its own -- the first version --
// left the 32 tiles of a small symmetric block running alone next to 480 empty positions, after which the next job's
// workgroups start in two shifts and the cohorts of an XCD no longer walk their rows together for the rest of the
// launch: a band of the host path or a rank's share ran 5-6 % below the rate of one large job, tools/gram_shapes.py.)
__device__ __forceinline__ bool gram_position_to_tile(const uint32_t live, const uint32_t lb, const int lr, const int lc, uint32_t& t) {
    const uint32_t xcd = lb & 7u;
    const uint32_t seq = lb >> 3;                      // position in this XCD's stream
    const int g = lr + lc;
    const uint32_t full = live >> (g + 3);             // rounds of 8 whole groups
    if ((seq >> g) < full) {
        t = ((((seq >> g) << 3) + xcd) << g) + (seq & ((1u << g) - 1u));
        return true;
    }
    // The last, partial round is dealt out evenly, an eighth to each XCD (dealt by whole groups, 2080 tiles
    // = 32.5 groups of 64 at N = 4096 left one XCD with 288 tiles and seven with 256: the launch took 9 tile
    // times per CU instead of 8.1).
    const uint32_t rest = live - (full << (g + 3));
    const uint32_t chunk = (rest + 7u) >> 3;
    const uint32_t within = seq - (full << g);
    if (within >= chunk) return false;
    t = (full << (g + 3)) + xcd * chunk + within;
    return t < live;
}

// Tile number t of job J's list of live tiles -> block tile (ti, tj).
__device__ __forceinline__ void gram_tile_of(const GramJobDev& J, uint32_t t, const int lr, const int lc, int& ti, int& tj) {
    const uint32_t br = 1u << lr, TMt = (uint32_t)J.tiles_m, TNt = (uint32_t)J.tiles_n;
    uint32_t lo = 0, hi = ((TMt + br - 1u) >> lr) - 1u;    // the band: binary search, every band before `hi` is full
    if (!(J.flags & QK_GRAM_SYMMETRIC)) {
        const uint32_t per_band = TNt << lr;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            if (mid * per_band <= t) lo = mid; else hi = mid - 1u;
        }
        const uint32_t idx = t - lo * per_band;
        const uint32_t rb = min(br, TMt - (lo << lr));
        const uint32_t c = rb == br ? idx >> lr : gram_div_small(idx, rb);
        tj = (int)c;
        ti = (int)((lo << lr) + (idx - c * rb));
        return;
    }
    // symmetric (square tiles): band m holds, for the columns c >= m * br, the rows m * br .. min(c, m * br + rb - 1):
    // rb (TNt - m br) - rb (rb - 1) / 2 tiles; m full bands hold m (br TNt - br (br - 1) / 2) - br^2 m (m - 1) / 2
    const uint32_t per_full = (TNt << lr) - ((br * (br - 1u)) >> 1);
    auto before = [&](uint32_t m) { return m * per_full - (((m * (m - 1u)) >> 1) << (2 * lr)); };
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        if (before(mid) <= t) lo = mid; else hi = mid - 1u;
    }
    t -= before(lo);
    const uint32_t first = lo << lr;
    const uint32_t rb = min(br, TMt - first);
    const uint32_t ragged = (rb * (rb - 1u)) >> 1;    // tiles of the first rb - 1 columns (1, 2, ... rows)
    uint32_t k, r;
    if (t < ragged) {
        k = 0;
        while (((k + 1u) * (k + 2u)) >> 1 <= t) ++k;
        r = t - ((k * (k + 1u)) >> 1);
    } else {
        const uint32_t u = t - ragged;
        const uint32_t q = rb == br ? u >> lr : gram_div_small(u, rb);
        k = rb - 1u + q;
        r = u - q * rb;
    }
    ti = (int)(first + r);
    tj = (int)(first + k);
}

// The job record is read through a pointer to the kernel-argument segment: through the by-value struct the
// compiler loads all 16 records at the top of the kernel (27 scalar loads; 700 spilled SGPRs where the persistent
// kernel locates its next tile with a full register file); through the pointer it loads the one it needs.
typedef const __attribute__((address_space(4))) GramArgs* gram_kargs_t;
__device__ __forceinline__ bool gram_locate(gram_kargs_t ap, const uint32_t bid, GramJobDev& J, int& ti, int& tj) {
    uint32_t t;
    if (!gram_position_to_tile(ap->total_live, bid, ap->lr, ap->lc, t)) return false;
    const int nj = ap->n_jobs;
    int k = 0;
    for (int q = 1; q < nj; ++q)
        if (t >= (uint32_t)ap->job[q].tile_begin) k = q;
    const __attribute__((address_space(4))) GramJobDev* s = &ap->job[k];
    J.A = s->A; J.B = s->B; J.K = s->K; J.KT = s->KT;
    J.ma = s->ma; J.nb = s->nb; J.ldk = s->ldk; J.ldkt = s->ldkt;
    J.flags = s->flags; J.tiles_m = s->tiles_m; J.tiles_n = s->tiles_n;
    J.tile_begin = s->tile_begin; J.live = s->live;
    gram_tile_of(J, t - (uint32_t)J.tile_begin, ap->lr, ap->lc, ti, tj);
    return true;
}
// for a kernel whose only argument is GramArgs (offset 0 of the segment)
__device__ __forceinline__ bool gram_locate(const GramArgs&, const int64_t bid, GramJobDev& J, int& ti, int& tj) {
    return gram_locate((gram_kargs_t)__builtin_amdgcn_kernarg_segment_ptr(), (uint32_t)bid, J, ti, tj);
}

// |.|^2 and the stores.  f64 16x16x4 result layout: col = lane & 15, row = (lane >> 4) + 4 * reg.
template <int TM, int TN, bool K3M, int NACC>
__device__ __forceinline__ void gram_epilogue(v4d (&acc)[NACC][TM][TN], const GramJobDev& J, int64_t wrow0,
                                              int64_t wcol0, int lane, bool sym) {
    double* __restrict__ Kp = J.K;
    double* __restrict__ KTp = J.KT;
    const int64_t ldk = J.ldk, ldkt = J.ldkt, ma = J.ma, nb = J.nb;
    if constexpr (K3M) {
        // A wave whose 16 TM x 16 TN entries all lie inside the block and (symmetric job) strictly above the
        // diagonal -- all but the edge and diagonal tiles -- needs no test per entry, and its addresses are a
        // wave-uniform base (scalar registers and scalar arithmetic) plus one 32-bit lane offset per store
        // pattern.  That leaves five FP64 operations per entry for the vector ALU, which the waves of the other
        // workgroups on this CU need for their MFMAs (the general path below: 330 vector instructions per wave,
        // 10 % of a tile's time at 8 qubits).
        constexpr int64_t kLd32 = (int64_t)1 << 24;   // 15 rows x ld x 8 B stay below 2^31
        const bool inside = wrow0 + TM * 16 <= ma && wcol0 + TN * 16 <= nb && (!sym || wrow0 + TM * 16 <= wcol0) &&
                            ldk <= kLd32 && (!KTp || ldkt <= kLd32);
        if (__builtin_amdgcn_readfirstlane(inside)) {
            const uint32_t lr4 = (uint32_t)(lane >> 4), lc16 = (uint32_t)(lane & 15);
            uint32_t off_d = (lr4 * (uint32_t)ldk + lc16) * 8u;        // K[i][j]
            uint32_t off_m = (lc16 * (uint32_t)ldk + lr4) * 8u;        // K[j][i]
            uint32_t off_t = (lc16 * (uint32_t)ldkt + lr4) * 8u;       // KT[j][i]
            char* const bd = (char*)Kp + (wrow0 * ldk + wcol0) * 8;
            char* const bm = (char*)Kp + (wcol0 * ldk + wrow0) * 8;
            char* const bt = (char*)KTp + (wcol0 * ldkt + wrow0) * 8;
            auto put = [&](char* base, uint32_t& off, double v) __attribute__((always_inline)) {
                asm volatile("" : "+v"(off));   // keeps zext(off) at the store: selects the SGPR-base form
                *(double*)(base + (uint64_t)off) = v;
            };
            auto all = [&](auto mirror_c, auto kt_c) __attribute__((always_inline)) {
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double re = acc[0][m][n][r] + acc[1][m][n][r];
                            const double im = acc[2][m][n][r] + (acc[0][m][n][r] - acc[1][m][n][r]);
                            const double v = fma(re, re, im * im);
                            put(bd + ((m * 16 + 4 * r) * ldk + n * 16) * 8, off_d, v);
                            if constexpr (decltype(mirror_c)::value) put(bm + (n * 16 * ldk + m * 16 + 4 * r) * 8, off_m, v);
                            if constexpr (decltype(kt_c)::value) put(bt + (n * 16 * ldkt + m * 16 + 4 * r) * 8, off_t, v);
                        }
            };
            using std::true_type;
            using std::false_type;
            if (sym && KTp) all(true_type{}, true_type{});
            else if (sym) all(true_type{}, false_type{});
            else if (KTp) all(false_type{}, true_type{});
            else all(false_type{}, false_type{});
            return;
        }
    }
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
    if (!gram_locate(args, (int64_t)blockIdx.x, J, ti, tj)) return;
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

constexpr int kSyncCounters = 1 << 18;
static __device__ unsigned int g_sync_counters[kSyncCounters];   // pacing counters of the ring kernel (see there)

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
    if (!gram_locate(args, (int64_t)blockIdx.x, J, ti, tj)) return;
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

// ---- "ring" LDS-DMA kernel: stages of 8 amplitudes in a ring of NST 16 KiB buffers, with an inner loop stripped of
// vector-ALU work.  Ablation builds of gram_dma_kernel in round 2 showed that every VALU instruction a wave issues
// between its MFMAs costs matrix-pipe time (the FP64 MFMA and the FP64 vector ALU are the same multipliers; 4 v_add_f64
// per k-step = 2.4 %, the 64-bit address adds and readfirstlanes around each DMA instruction another 3.6 %).  Here
//   * DMA sources are  SGPR base (advanced per stage on the scalar ALU) + a per-lane 32-bit offset that never
//     changes, the LDS destinations are scalar (wave index through readfirstlane once),
//   * fragment reads use precomputed per-lane LDS addresses with immediate offsets; the stage loop is unrolled over
//     the ring so that the buffer a stage lives in is a compile-time constant,
//   * loop control is 32-bit scalar,
// which leaves the TM + TN fragment sums as the only vector-ALU instructions of a k-step.
// The DMA of stage h + NST - 1 is issued while stage h is computed and only stage h + 1 is waited for
// (s_waitcnt vmcnt(4 (NST - 2)): the younger DMA instructions of the wave stay in flight across the barrier), so a
// load has NST - 1 stages to arrive instead of the remainder of one k-tile.  NST = 3: 48 KiB of LDS, THREE workgroups
// per CU (three waves per SIMD) -- for short rows, where a tile is over in tens of microseconds and its prologue,
// epilogue and the restart of the slot (measured with round 2's timeline build: 1, 3.5 and 6 us of a 49 us tile at D = 256)
// have to hide behind the other workgroups of the CU.  NST = 5: 80 KiB, two workgroups per CU, four stages of
// prefetch -- for long rows, where the working set of a third workgroup costs more in L2 than it hides.
// LDS image of a stage: [A: 64 rows | B: 64 rows] x 128 bytes; one DMA instruction fills 8 rows; slot s of row r
// holds amplitude s ^ ((r >> 1) & 7), which spreads the 16 rows x one k-slot of a fragment read over all 64 banks.
template <int NST>
__global__ void __launch_bounds__(256, NST == 3 ? 3 : 2)
gram_ring_kernel(GramArgs args) {
    constexpr int TM = 2, TN = 2, WAVES_N = 2, WAVES = 4, BM = 64, BN = 64;
    constexpr int STAGE = (BM + BN) * 128;          // bytes
    constexpr int NA = BM / 8 / WAVES, NB = BN / 8 / WAVES, NE = NA + NB;
    constexpr int INFLIGHT = NE * (NST - 2);        // DMA instructions of a wave that may stay in flight at a barrier
    static_assert(NST == 3 || NST == 5, "ring of 3 or 5 stages");
    extern __shared__ v2d smem[];  // [NST stages][A | B]
    // Outside the stage loop a wave runs at raised priority: next to waves that keep the matrix pipe full, a wave
    // at equal priority is granted about one instruction per MFMA (27 ns) whatever its type -- round 2's timeline build
    // showed 10-26 us for the few hundred scalar instructions that find the tile, and as much for the epilogue,
    // during which the workgroup's LDS ring sits idle.
    __builtin_amdgcn_s_setprio(3);
    GramJobDev J;
    int ti, tj;
    if (!gram_locate(args, (int64_t)blockIdx.x, J, ti, tj)) return;
    const bool sym = (J.flags & QK_GRAM_SYMMETRIC) != 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const uint32_t D = (uint32_t)args.D, LD = (uint32_t)args.ld;   // the host checks ld <= 2^22: 63 rows x ld x 16 B fit 32 bits
    const int64_t row0 = (int64_t)ti * BM, col0 = (int64_t)tj * BN;

    uint32_t vo[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int j = e < NA ? e : e - NA;
        const int row = 8 * (wave + WAVES * j) + (lane >> 3);
        const uint32_t room = (uint32_t)((e < NA ? J.ma - row0 : J.nb - col0) - 1);   // rows of the block past this tile's first (the host checks ma, nb < 2^31); rows beyond the edge re-read the last one
        const uint32_t rr = min((uint32_t)row, room);
        vo[e] = (rr * LD + (uint32_t)((lane & 7) ^ ((row >> 1) & 7))) * 16u;
    }
    const char* Ab = (const char*)J.A + row0 * (int64_t)LD * 16;   // wave-uniform: scalar registers
    const char* Bb = (const char*)J.B + col0 * (int64_t)LD * 16;
    char* const lds = (char*)smem;
    auto dma = [&](int buf, uint32_t koff, int e) __attribute__((always_inline)) {
        const int j = e < NA ? e : e - NA;
        const char* src = (e < NA ? Ab : Bb) + koff;
        char* dst = lds + buf * STAGE + (e < NA ? 0 : BM * 128) + (wave + WAVES * j) * 1024;
        asm volatile("" : "+v"(vo[e]));   // keeps zext(lane offset) in the loop body: selects the SGPR-base form
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (uint64_t)vo[e]), (lptr_t)dst, 16, 0, 0);
    };
    // stage h + 1 has landed for this wave (its own DMA instructions complete in order), then for all of them
    auto wait_next_stage = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(INFLIGHT) : "memory");
    };

    v4d acc[3][TM][TN];
    const int stages = (int)(D >> 3);
    const int lastst = stages - 1;
    // Pacing (long rows only).  The 32 workgroups that start together on an XCD -- half a group: the two workgroups
    // of a CU run half a tile apart -- share their A and B row blocks in that XCD's L2 only while they are at the
    // same place in their rows: 4 MiB hold 13-20 us of the k-loop, and left alone their starts spread over 40 us
    // and drift (on some boxes of the pool down to a 24 % hit rate; the fabric traffic costs 4-5 % of clock on a
    // power-bound chip).  At its start and every 2^sync_shift trips of the stage loop thread 0 reports to the
    // cohort's counter and waits -- bounded: a missing member costs time, never progress -- until all of them have.
    unsigned int* sync_cnt = nullptr;
    unsigned int sync_members = 0, sync_epoch = 0;
    if (NST == 5 && args.sync) {
        // the cohort: 32 consecutive positions of one XCD label; how many of them hold a tile (as in gram_tile_of, 8 x 8 groups)
        const uint32_t lb = blockIdx.x, seq = lb >> 3, xcd = lb & 7u;
        const uint32_t live = args.total_live, full = live >> 9;      // (as in gram_position_to_tile: the launch's one list)
        sync_members = 32;
        if ((seq >> 6) >= full) {
            const uint32_t rest = live - (full << 9), chunk = (rest + 7u) >> 3;
            const uint32_t first = (seq - (full << 6)) & ~31u;
            const uint32_t mine = min(chunk, rest > xcd * chunk ? rest - xcd * chunk : 0u);
            sync_members = mine > first ? min(32u, mine - first) : 0u;
        }
        sync_cnt = args.sync + (((int64_t)blockIdx.x >> 8) * 8 + (blockIdx.x & 7));
    }
    auto meet = [&](int limit) __attribute__((always_inline)) {
        if (threadIdx.x == 0) {
            ++sync_epoch;
            const unsigned int target = sync_members * sync_epoch;
            atomicAdd(sync_cnt, 1u);
            for (int spin = 0; spin < limit; ++spin) {       // x s_sleep(8) = 512 cycles
                if (__hip_atomic_load(sync_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
    };
    if (sync_cnt) meet(400);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
#pragma unroll
        for (int e = 0; e < NE; ++e) dma(s, (uint32_t)min(s, lastst) * 128u, e);

    const int frag_k = lane >> 4, frag_r = lane & 15;
    typedef const __attribute__((address_space(3))) v2d* lfrag_t;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
    uint32_t fa[2], fb[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int x = (4 * kk + frag_k) ^ (frag_r >> 1);
        fa[kk] = lds0 + (uint32_t)((wm * (TM * 16) + frag_r) * 128 + x * 16);
        fb[kk] = lds0 + (uint32_t)(BM * 128 + (wn * (TN * 16) + frag_r) * 128 + x * 16);
        asm volatile("" : "+v"(fa[kk]), "+v"(fb[kk]));   // opaque: otherwise the sums are redone per read
    }
    v2d a[TM], b[TN];
    auto read_frags = [&](int buf, int kk) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < TM; ++m) a[m] = *(lfrag_t)(size_t)(fa[kk] + (uint32_t)(buf * STAGE + m * 2048));
#pragma unroll
        for (int n = 0; n < TN; ++n) b[n] = *(lfrag_t)(size_t)(fb[kk] + (uint32_t)(buf * STAGE + n * 2048));
    };

    // stage hh out of buffer S; the DMA of stage hh + NST - 1 goes into buffer S - 1 (mod NST), which every wave
    // left behind at the last barrier.  Past the end the DMA re-reads the last stage into a buffer nobody reads
    // any more: branch-free and harmless.
    auto stage = [&](auto s_c, int hh) __attribute__((always_inline)) {
        constexpr int S = decltype(s_c)::value;
        const uint32_t koff = (uint32_t)min(hh + NST - 1, lastst) * 128u;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            double sa[TM], db[TN];
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[0][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b[n].x, acc[0][m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < TM; ++m) sa[m] = a[m].x + a[m].y;
#pragma unroll
            for (int n = 0; n < TN; ++n) db[n] = b[n].y - b[n].x;
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if ((e & 1) == kk) dma((S + NST - 1) % NST, koff, e);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[1][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].y, b[n].y, acc[1][m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk == 0) {
                read_frags(S, 1);
            } else {
                wait_next_stage();
                read_frags((S + 1) % NST, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[2][m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[m], db[n], acc[2][m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using std::integral_constant;
    const int meet_mask = (1 << args.sync_shift) - 1;
    wait_next_stage();   // here: stage 0
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[c][m][n] = (v4d){0, 0, 0, 0};
    read_frags(0, 0);
    __builtin_amdgcn_s_setprio(0);
    int h = 0;
    for (int trip = 1; h + NST <= stages; h += NST, ++trip) {
        if (sync_cnt && (trip & meet_mask) == 0) meet(40);      // scalar test; the other waves meet thread 0 at the next barrier
        stage(integral_constant<int, 0>{}, h);
        stage(integral_constant<int, 1>{}, h + 1);
        stage(integral_constant<int, 2>{}, h + 2);
        if constexpr (NST == 5) {
            stage(integral_constant<int, 3>{}, h + 3);
            stage(integral_constant<int, 4>{}, h + 4);
        }
    }
    // the stage count is a power of two: 1 .. NST - 1 stages are left, starting in buffer 0 again
    if (h < stages) stage(integral_constant<int, 0>{}, h);
    if (h + 1 < stages) stage(integral_constant<int, 1>{}, h + 1);
    if constexpr (NST == 5) {
        if (h + 2 < stages) stage(integral_constant<int, 2>{}, h + 2);
        if (h + 3 < stages) stage(integral_constant<int, 3>{}, h + 3);
    }
    __builtin_amdgcn_s_setprio(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may still be writing this workgroup's LDS when it ends
    gram_epilogue<TM, TN, true>(acc, J, row0 + wm * (TM * 16), col0 + wn * (TN * 16), lane, sym);
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
    if (!gram_locate(args, (int64_t)blockIdx.x, J, ti, tj)) return;
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

// ---- FP32 ring kernel: gram_ring_kernel's loop and LDS ring (stages of 8 units = 16 amplitudes) with the FP32
// arithmetic of gram32_dma_kernel above (one 16-byte unit feeds two MFMA k-steps, 3-multiplication product, the
// running accumulators folded into (Re, Im) every 1024 amplitudes).
template <int NST>
__global__ void __launch_bounds__(256, NST == 3 ? 3 : 2)
gram32_ring_kernel(GramArgs args) {
    constexpr int TM = 2, TN = 2, WAVES_N = 2, WAVES = 4, BM = 64, BN = 64;
    constexpr int STAGE = (BM + BN) * 128;          // bytes
    constexpr int NA = BM / 8 / WAVES, NB = BN / 8 / WAVES, NE = NA + NB;
    constexpr int INFLIGHT = NE * (NST - 2);
    constexpr int FLUSH = 64;                       // stages per first-level accumulation run (1024 amplitudes)
    static_assert(NST == 3 || NST == 5, "ring of 3 or 5 stages");
    extern __shared__ v4f smem32[];
    GramJobDev J;
    int ti, tj;
    __builtin_amdgcn_s_setprio(3);   // as in gram_ring_kernel: raised priority outside the stage loop
    if (!gram_locate(args, (int64_t)blockIdx.x, J, ti, tj)) return;
    const bool sym = (J.flags & QK_GRAM_SYMMETRIC) != 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const uint32_t U = (uint32_t)(args.D >> 1);     // 16-byte units per row; the host checks D <= 2^22
    const int64_t row0 = (int64_t)ti * BM, col0 = (int64_t)tj * BN;
    const int64_t ma = J.ma, nb = J.nb;

    uint32_t vo[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int j = e < NA ? e : e - NA;
        const int row = 8 * (wave + WAVES * j) + (lane >> 3);
        const int64_t first = e < NA ? row0 : col0, last = (e < NA ? ma : nb) - 1;
        const uint32_t rr = (uint32_t)(min(first + row, last) - first);
        vo[e] = (rr * U + (uint32_t)((lane & 7) ^ ((row >> 1) & 7))) * 16u;
    }
    const char* Ab = (const char*)J.A + row0 * (int64_t)U * 16;
    const char* Bb = (const char*)J.B + col0 * (int64_t)U * 16;
    char* const lds = (char*)smem32;
    auto dma = [&](int buf, uint32_t koff, int e) __attribute__((always_inline)) {
        const int j = e < NA ? e : e - NA;
        const char* src = (e < NA ? Ab : Bb) + koff;
        char* dst = lds + buf * STAGE + (e < NA ? 0 : BM * 128) + (wave + WAVES * j) * 1024;
        asm volatile("" : "+v"(vo[e]));
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (uint64_t)vo[e]), (lptr_t)dst, 16, 0, 0);
    };
    auto wait_next_stage = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(INFLIGHT) : "memory");
    };

    v4f acc[3][TM][TN], hre[TM][TN], him[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            acc[0][m][n] = acc[1][m][n] = acc[2][m][n] = (v4f){0, 0, 0, 0};
            hre[m][n] = him[m][n] = (v4f){0, 0, 0, 0};
        }
    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                hre[m][n] += acc[0][m][n] + acc[1][m][n];
                him[m][n] += acc[2][m][n] + (acc[0][m][n] - acc[1][m][n]);
                acc[0][m][n] = acc[1][m][n] = acc[2][m][n] = (v4f){0, 0, 0, 0};
            }
    };

    const int stages = (int)(U >> 3);
    const int lastst = stages - 1;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
#pragma unroll
        for (int e = 0; e < NE; ++e) dma(s, (uint32_t)min(s, lastst) * 128u, e);
    wait_next_stage();

    const int frag_k = lane >> 4, frag_r = lane & 15;
    typedef const __attribute__((address_space(3))) v4f* lfrag_t;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem32;
    uint32_t fa[2], fb[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int x = (4 * kk + frag_k) ^ (frag_r >> 1);
        fa[kk] = lds0 + (uint32_t)((wm * (TM * 16) + frag_r) * 128 + x * 16);
        fb[kk] = lds0 + (uint32_t)(BM * 128 + (wn * (TN * 16) + frag_r) * 128 + x * 16);
        asm volatile("" : "+v"(fa[kk]), "+v"(fb[kk]));
    }
    v4f a[TM], b[TN];
    auto read_frags = [&](int buf, int kk) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < TM; ++m) a[m] = *(lfrag_t)(size_t)(fa[kk] + (uint32_t)(buf * STAGE + m * 2048));
#pragma unroll
        for (int n = 0; n < TN; ++n) b[n] = *(lfrag_t)(size_t)(fb[kk] + (uint32_t)(buf * STAGE + n * 2048));
    };
    read_frags(0, 0);

    auto stage = [&](auto s_c, int hh) __attribute__((always_inline)) {
        constexpr int S = decltype(s_c)::value;
        const uint32_t koff = (uint32_t)min(hh + NST - 1, lastst) * 128u;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float sa[2][TM], db[2][TN];
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[0][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b[n].x, acc[0][m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < TM; ++m) { sa[0][m] = a[m].x + a[m].y; sa[1][m] = a[m].z + a[m].w; }
#pragma unroll
            for (int n = 0; n < TN; ++n) { db[0][n] = b[n].y - b[n].x; db[1][n] = b[n].w - b[n].z; }
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if ((e & 1) == kk) dma((S + NST - 1) % NST, koff, e);
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
            if (kk == 0) {
                read_frags(S, 1);
            } else {
                wait_next_stage();
                read_frags((S + 1) % NST, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[2][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1][m], db[1][n], acc[2][m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (((hh + 1) & (FLUSH - 1)) == 0) flush();
    };
    using std::integral_constant;
    __builtin_amdgcn_s_setprio(0);
    int h = 0;
    for (; h + NST <= stages; h += NST) {
        stage(integral_constant<int, 0>{}, h);
        stage(integral_constant<int, 1>{}, h + 1);
        stage(integral_constant<int, 2>{}, h + 2);
        if constexpr (NST == 5) {
            stage(integral_constant<int, 3>{}, h + 3);
            stage(integral_constant<int, 4>{}, h + 4);
        }
    }
    if (h < stages) stage(integral_constant<int, 0>{}, h);
    if (h + 1 < stages) stage(integral_constant<int, 1>{}, h + 1);
    if constexpr (NST == 5) {
        if (h + 2 < stages) stage(integral_constant<int, 2>{}, h + 2);
        if (h + 3 < stages) stage(integral_constant<int, 3>{}, h + 3);
    }
    __builtin_amdgcn_s_setprio(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

template <int NST>
void launch_ring_variant(const GramArgs& args, unsigned blocks, hipStream_t stream) {
    hipLaunchKernelGGL((gram_ring_kernel<NST>), dim3(blocks), dim3(256), NST * 128 * 128, stream, args);
}

static const Variant kVariants[] = {
    QK_VARIANT("4m_128x128_w8_bk8", 2, 4, 4, 2, 8, false),          // under 3 qubits (rows shorter than a DMA stage)
    QK_DMA_VARIANT("dma3m_64x64_w4_g64", 2, 2, 2, 2, true, 3, 3),   // rows beyond 2^22 amplitudes (64-bit addressing)
    Variant{"ring3_64x64_w4", 64, 64, 256, 8, 3, 3, 3 * 128 * 128, (const void*)gram_ring_kernel<3>, launch_ring_variant<3>},   // three workgroups per CU
    Variant{"ring5_64x64_w4", 64, 64, 256, 8, 3, 3, 5 * 128 * 128, (const void*)gram_ring_kernel<5>, launch_ring_variant<5>},   // two per CU, four stages of prefetch
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
// Default, from in-process A/B runs (tools/gram_ab.py): the ring kernels, three stages and three
// workgroups per CU for rows of up to 1024 amplitudes, five stages and two workgroups per CU above.  All variants
// give bitwise identical results (same tiles, same summation order).
constexpr int kSmallVariant = 0, kFallbackVariant = 1, kRing3Variant = 2, kRing5Variant = 3;

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
    (void)jobs; (void)n_jobs;
    return D <= 1024 ? kRing3Variant : kRing5Variant;
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
    if (D < kVariants[vi].min_d) vi = kSmallVariant;
    if (D > ((int64_t)1 << 22) && vi >= kRing3Variant) vi = kFallbackVariant;
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
    args.ld = D;
    args.lr = lr;
    args.lc = lc;
    blocks = 0;
    int64_t total = 0;     // live tiles so far
    int nj = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const qk_gram_job& j = jobs[k];
        if (j.ma < 0 || j.nb < 0) return fail(QK_ERR_INVALID, std::string(who) + ": negative block size");
        if (j.ma > 0x7fffffffLL || j.nb > 0x7fffffffLL) return fail(QK_ERR_UNSUPPORTED, std::string(who) + ": more than 2^31 - 1 rows in a block");
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
        d.tile_begin = total;
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
        total += d.live;
    }
    args.n_jobs = nj;
    const int64_t per_group = (int64_t)1 << (lr + lc);
    const int64_t groups = (total + per_group - 1) / per_group;
    blocks = ((groups + 7) / 8) * 8 * per_group;  // whole rounds of 8 groups, one per XCD label, over ALL jobs' tiles
    if (total > 0x7fffffffLL) return fail(QK_ERR_UNSUPPORTED, std::string(who) + ": too many tiles; split the call");
    args.total_live = (uint32_t)total;
    if (blocks > 0x7fffffffLL) return fail(QK_ERR_UNSUPPORTED, std::string(who) + ": too many tiles; split the call");
    return QK_OK;
}
}  // namespace qk

extern "C" int qk_gram_jobs_dev(const qk_gram_job* jobs, int n_jobs, int64_t D, void* stream) {
    using namespace qk;
    int vi = pick_variant(jobs, n_jobs, D);
    if (D < kVariants[vi].min_d) vi = kSmallVariant;   // a DMA stage is 8 amplitudes (16 for the fallback kernel)
    if (D > ((int64_t)1 << 22) && vi >= kRing3Variant) vi = kFallbackVariant;   // 32-bit lane offsets of the rings
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
    // Pacing counters of the ring kernel (see gram_ring_kernel): one per cohort of 32 workgroups, cleared in stream
    // order before the launch.  The table is one static array per device, handed out as a ring: every launch takes
    // the next slice of it (a ticket drawn here, under a mutex, in the order the host enqueues), so launches that
    // are in flight together on one device -- two streams of a host application, the chromosome batches of a GA
    // next to something else -- never clear or count on each other's counters.  A slice is reused only after the
    // whole ring (2^18 counters = 16 launches of the headline size) has been handed out since; even then a clash
    // costs bounded waits and nothing else: the counters pace, they do not guard data.
    // QK_GRAM_PACE=0 turns pacing off, any other value is log2 of the stage-loop trips between meetings;
    // QK_GRAM_PACE_SHARED=1 gives every launch the slice at offset 0, the behaviour before round 3 (A/B runs).
    static const int pace_env = std::getenv("QK_GRAM_PACE") ? std::atoi(std::getenv("QK_GRAM_PACE")) : 7;
    static const bool pace_shared = std::getenv("QK_GRAM_PACE_SHARED") && std::atoi(std::getenv("QK_GRAM_PACE_SHARED")) != 0;
    args.sync_shift = pace_env > 0 ? pace_env : 7;
    const int64_t need = ((blocks / 32 + 8) + 63) / 64 * 64;
    if (pace_env > 0 && vi == kRing5Variant && D >= 4096 && need <= kSyncCounters) {
        static std::mutex mu;
        static unsigned int* table[64] = {};
        static int64_t next[64] = {};
        int dev = 0;
        QK_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64) return fail(QK_ERR_UNSUPPORTED, "qk_gram_jobs_dev: device index out of range");
        unsigned int* cnt = nullptr;
        int64_t off = 0;
        {
            std::lock_guard<std::mutex> lock(mu);
            if (!table[dev]) QK_HIP(hipGetSymbolAddress((void**)&table[dev], HIP_SYMBOL(g_sync_counters)));
            cnt = table[dev];
            if (!pace_shared) {
                if (next[dev] + need > kSyncCounters) next[dev] = 0;
                off = next[dev];
                next[dev] += need;
            }
        }
        QK_HIP(hipMemsetAsync(cnt + off, 0, (size_t)need * sizeof(unsigned int), (hipStream_t)stream));
        args.sync = cnt + off;
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
template <int NST>
static int launch_gram32_ring(const qk_gram_job* jobs, int n_jobs, int64_t D, hipStream_t stream) {
    GramArgs args;
    int64_t blocks = 0;
    const int rc = build_gram_args(jobs, n_jobs, D, 64, 64, 3, 3, "qk_gram32_jobs_dev", args, blocks);
    if (rc) return rc;
    if (args.n_jobs == 0) return QK_OK;
    constexpr int lds = NST * 128 * 128;
    static DeviceOnce attr_set;
    if (attr_set.pending()) {
        QK_HIP(hipFuncSetAttribute((const void*)gram32_ring_kernel<NST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set.done();
    }
    hipLaunchKernelGGL((gram32_ring_kernel<NST>), dim3((unsigned)blocks), dim3(256), lds, stream, args);
    QK_HIP(hipGetLastError());
    return QK_OK;
}
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
    // the ring kernels, as in FP64; rows beyond 2^22 amplitudes: the double-buffered kernel with 64-bit addressing
    if (D > ((int64_t)1 << 22)) return launch_gram32<2, 2, 2, 2>(jobs, n_jobs, D, (hipStream_t)stream);
    return D <= 1024 ? launch_gram32_ring<3>(jobs, n_jobs, D, (hipStream_t)stream)
                     : launch_gram32_ring<5>(jobs, n_jobs, D, (hipStream_t)stream);
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
