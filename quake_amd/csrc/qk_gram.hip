// Pairwise fidelity  K[i][j] = |sum_b conj(A[i][b]) B[j][b]|^2  on FP64 MFMA (gfx950).
//
// Psi rows are interleaved complex128, i.e. real vectors of length 2D.  With
// a = A[i][b], c = B[j][b]:
//     Re = sum  a.re*c.re + a.im*c.im          Im = sum  a.re*c.im - a.im*c.re
// so one 16x16 output tile takes four v_mfma_f64_16x16x4_f64 per 4 amplitudes
// (8 real flops per amplitude and entry = the algorithmic count of SURVEY.md 8d);
// the MFMA k-slot of a lane (lane >> 4) is simply "which of the 4 amplitudes", and
// each lane feeds the re and the im half of ONE 16-byte LDS read to the four MFMAs.
//
// Block tile 128 x 128, 8 waves (2 x 4), wave tile 64 x 32 (4 x 2 MFMA tiles, Re
// and Im accumulators = 128 VGPRs), 2 waves per SIMD, one workgroup per CU.  A and
// B tiles are staged global -> registers -> LDS, double buffered, one barrier per
// BK amplitudes.  LDS images are k-major ([amplitude][row], 16-B elements, one
// element of padding per k row) so fragment reads are conflict free.
//
// Tile schedule: jobs are laid out in 16x16-tile super-tiles of 256 workgroups;
// inside one, the 8 XCD labels (block id % 8) each own a 4 x 8 sub-block, so the
// workgroups that share an L2 share A rows (8x) and B rows (4x).
#include <hip/hip_runtime.h>

#include <cstring>

#include "qk_common.h"

namespace qk {

constexpr int BM = 128, BN = 128, BK = 8;  // BK in amplitudes
constexpr int GRAM_THREADS = 512;
constexpr int LDA = BM + 1;                // padded k-row length (16-B elements)
constexpr int SUPER = 16;                  // super-tile side, in tiles

typedef double v4d __attribute__((ext_vector_type(4)));

struct GramJobDev {
    const double2* A;
    const double2* B;
    double* K;
    double* KT;
    int64_t ma, nb, ldk, ldkt;
    int32_t flags;
    int32_t tiles_m, tiles_n;  // 128-row tiles
    int32_t sup_m, sup_n;      // super-tiles per side
    int64_t block_begin;       // first workgroup of this job
};

struct GramArgs {
    GramJobDev job[QK_GRAM_MAX_JOBS];
    int n_jobs;
    int64_t D;  // amplitudes per row
};

__global__ void __launch_bounds__(GRAM_THREADS, 2) gram_kernel(GramArgs args) {
    extern __shared__ double2 smem[];  // [2 buffers][A | B][BK][LDA]
    // ---- which tile ---------------------------------------------------------
    const int64_t bid = blockIdx.x;
    int ji = 0;
    for (int k = 1; k < args.n_jobs; ++k)
        if (bid >= args.job[k].block_begin) ji = k;
    const GramJobDev& J = args.job[ji];
    const int64_t lb = bid - J.block_begin;
    const int su = (int)(lb >> 8);
    const int slot = (int)(lb & 255);
    const int xcd = slot & 7, idx = slot >> 3;
    const int sr = (xcd >> 1) * 4 + (idx >> 3);
    const int sc = (xcd & 1) * 8 + (idx & 7);
    const bool sym = (J.flags & QK_GRAM_SYMMETRIC) != 0;
    int SI, SJ;
    if (sym) {  // su enumerates super-tiles (I, J >= I) row by row
        int I = 0, rem = su, rowlen = J.sup_n;
        while (rem >= rowlen) { rem -= rowlen; --rowlen; ++I; }
        SI = I;
        SJ = I + rem;
    } else {
        SI = su / J.sup_n;
        SJ = su % J.sup_n;
    }
    const int ti = SI * SUPER + sr, tj = SJ * SUPER + sc;
    if (ti >= J.tiles_m || tj >= J.tiles_n || (sym && tj < ti)) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int64_t D = args.D;
    const int64_t row0 = (int64_t)ti * BM, col0 = (int64_t)tj * BN;

    // ---- global -> register staging map: lanes 0..7 cover 128 contiguous bytes of a row
    const int ld_k = tid & 7;
    const int ld_r = tid >> 3;  // 0..63 (+64 for the second element)
    // clamp rows: values of clamped rows land in LDS rows whose results are never stored
    const int64_t ra0_ = min(row0 + ld_r, J.ma - 1), ra1_ = min(row0 + ld_r + 64, J.ma - 1);
    const int64_t rb0_ = min(col0 + ld_r, J.nb - 1), rb1_ = min(col0 + ld_r + 64, J.nb - 1);
    const bool k_in_range = ld_k < D;  // D < BK only for circuits under 3 qubits
    const int ldk_c = k_in_range ? ld_k : 0;
    const double2* gA0 = J.A + ra0_ * D + ldk_c;
    const double2* gA1 = J.A + ra1_ * D + ldk_c;
    const double2* gB0 = J.B + rb0_ * D + ldk_c;
    const double2* gB1 = J.B + rb1_ * D + ldk_c;

    double2* As0 = smem;
    double2* Bs0 = smem + BK * LDA;
    constexpr int BUF = 2 * BK * LDA;
    const int st_off = ld_k * LDA + ld_r;

    v4d accR[4][2], accI[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            accR[m][n] = (v4d){0, 0, 0, 0};
            accI[m][n] = (v4d){0, 0, 0, 0};
        }

    const int64_t ktiles = (D + BK - 1) / BK;
    const double2 zero2 = make_double2(0.0, 0.0);
    // prologue: tile 0 -> LDS buffer 0
    double2 pa0 = gA0[0], pa1 = gA1[0], pb0 = gB0[0], pb1 = gB1[0];
    if (!k_in_range) { pa0 = zero2; pa1 = zero2; pb0 = zero2; pb1 = zero2; }
    As0[st_off] = pa0;
    As0[st_off + 64] = pa1;
    Bs0[st_off] = pb0;
    Bs0[st_off + 64] = pb1;
    __syncthreads();

    const int frag_k = lane >> 4, frag_r = lane & 15;
    const int a_off = frag_k * LDA + wm * 64 + frag_r;
    const int b_off = frag_k * LDA + wn * 32 + frag_r;
    for (int64_t kt = 0; kt < ktiles; ++kt) {
        const int cur = (int)(kt & 1);
        // prefetch the next k-tile into registers (the last iteration re-reads its own
        // tile: branch-free, and nothing reads the buffer it is written to afterwards)
        const int64_t off = min(kt + 1, ktiles - 1) * BK;
        pa0 = gA0[off];
        pa1 = gA1[off];
        pb0 = gB0[off];
        pb1 = gB1[off];
        // keep the loads up here: hipcc otherwise sinks them to their first use (the LDS
        // writes below) and every k-tile waits out a full HBM/L2 round trip
        __builtin_amdgcn_sched_barrier(0);
        const double2* As = As0 + cur * BUF;
        const double2* Bs = Bs0 + cur * BUF;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double2 a[4], b[2];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = As[kk * 4 * LDA + a_off + m * 16];
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = Bs[kk * 4 * LDA + b_off + n * 16];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    accR[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b[n].x, accR[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    accI[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b[n].y, accI[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    accR[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].y, b[n].y, accR[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const double nai = -a[m].y;
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    accI[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, b[n].x, accI[m][n], 0, 0, 0);
            }
        }
        double2* An = As0 + (cur ^ 1) * BUF;
        double2* Bn = Bs0 + (cur ^ 1) * BUF;
        An[st_off] = pa0;
        An[st_off + 64] = pa1;
        Bn[st_off] = pb0;
        Bn[st_off + 64] = pb1;
        __syncthreads();
    }

    // ---- epilogue: |.|^2, direct + mirrored stores ----------------------------
    // f64 16x16x4 result layout: col = lane & 15, row = (lane >> 4) + 4 * reg
    const bool diag_tile = sym && (ti == tj);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = row0 + wm * 64 + m * 16 + (lane >> 4) + 4 * r;
                const int64_t j = col0 + wn * 32 + n * 16 + (lane & 15);
                if (i >= J.ma || j >= J.nb) continue;
                const double re = accR[m][n][r], im = accI[m][n][r];
                double v = fma(re, re, im * im);
                if (diag_tile) {
                    if (i > j) continue;      // lower half of a diagonal tile comes from its mirror
                    if (i == j) v = 1.0;      // identical statevectors stay exactly 1
                }
                J.K[i * J.ldk + j] = v;
                if (sym && i != j) J.K[j * J.ldk + i] = v;
                if (J.KT) J.KT[j * J.ldkt + i] = v;
            }
}

}  // namespace qk

extern "C" int qk_gram_jobs_dev(const qk_gram_job* jobs, int n_jobs, int64_t D, void* stream) {
    using namespace qk;
    if (!jobs || n_jobs < 1 || n_jobs > QK_GRAM_MAX_JOBS)
        return fail(QK_ERR_INVALID, "qk_gram_jobs_dev: need 1..16 jobs");
    if (D < 1) return fail(QK_ERR_INVALID, "qk_gram_jobs_dev: D must be >= 1");
    if (D >= BK && (D % BK) != 0) return fail(QK_ERR_INVALID, "qk_gram_jobs_dev: D must be a power of two");
    GramArgs args;
    std::memset(&args, 0, sizeof(args));
    args.D = D;
    int64_t blocks = 0;
    int nj = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const qk_gram_job& j = jobs[k];
        if (j.ma < 0 || j.nb < 0) return fail(QK_ERR_INVALID, "qk_gram_jobs_dev: negative block size");
        if (j.ma == 0 || j.nb == 0) continue;
        if (!j.A || !j.B || !j.K) return fail(QK_ERR_INVALID, "qk_gram_jobs_dev: null pointer in job");
        if (j.ldk < j.nb || (j.KT && j.ldkt < j.ma))
            return fail(QK_ERR_INVALID, "qk_gram_jobs_dev: leading dimension smaller than the block");
        const bool sym = (j.flags & QK_GRAM_SYMMETRIC) != 0;
        if (sym && (j.ma != j.nb || j.A != j.B))
            return fail(QK_ERR_INVALID, "qk_gram_jobs_dev: a symmetric job needs A == B and ma == nb");
        GramJobDev& d = args.job[nj++];
        d.A = (const double2*)j.A;
        d.B = (const double2*)j.B;
        d.K = j.K;
        d.KT = j.KT;
        d.ma = j.ma; d.nb = j.nb; d.ldk = j.ldk; d.ldkt = j.ldkt;
        d.flags = j.flags;
        d.tiles_m = (int32_t)((j.ma + BM - 1) / BM);
        d.tiles_n = (int32_t)((j.nb + BN - 1) / BN);
        d.sup_m = (d.tiles_m + SUPER - 1) / SUPER;
        d.sup_n = (d.tiles_n + SUPER - 1) / SUPER;
        d.block_begin = blocks;
        const int64_t nsup = sym ? (int64_t)d.sup_n * (d.sup_n + 1) / 2 : (int64_t)d.sup_m * d.sup_n;
        blocks += nsup * 256;
    }
    if (nj == 0) return QK_OK;
    args.n_jobs = nj;
    if (blocks > 0x7fffffffLL) return fail(QK_ERR_UNSUPPORTED, "qk_gram_jobs_dev: too many tiles; split the call");
    const size_t lds = (size_t)2 * 2 * BK * LDA * sizeof(double2);
    static bool attr_set = false;
    if (!attr_set) {
        QK_HIP(hipFuncSetAttribute((const void*)gram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(gram_kernel, dim3((unsigned)blocks), dim3(GRAM_THREADS), lds, (hipStream_t)stream, args);
    QK_HIP(hipGetLastError());
    return QK_OK;
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
