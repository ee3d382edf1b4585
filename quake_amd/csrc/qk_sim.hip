// Batched statevector simulator for gfx950: one workgroup per (sample, chunk).
//
// The 2^L local amplitudes of a chunk live in LDS as interleaved complex128.  Each
// sample's gate angles are evaluated on the device from its feature row (a small
// postfix machine, one thread per angle), turned into (cos, sin) pairs and 2x2
// matrices in LDS, and then the pass's stages run back to back:
//   DIAG  one sweep multiplying every amplitude by the product of its phase factors
//   U1    2x2 butterfly on one local qubit
//   CX    conditional swap (control may be a chunk bit)
//   ECR   fixed 4x4 on two local qubits
// Psi is written to HBM in 16-byte elements, consecutive lanes -> consecutive amplitudes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "qk_common.h"
#include "qk_math.h"
#include "qk_rdm.h"
#include "qk_reg.h"
#include "qk_wave.h"

namespace qk {

struct SimArgs {
    const double2* cs;   // (cos, sin) of every gate angle of the pass, [n_expr][N] (angle_table_kernel)
    int64_t N;
    double2* psi;   // may be null in fused-RDM mode (the statevectors are never written)
    float2* psi32;  // FP32 output (qk_simulate_c64_dev): the state is evolved in FP64 and rounded once, at the store
    double* rdm;    // fused mode: N x n x 4 reduced density matrices, computed from LDS (single chunk per sample)
    int64_t D;  // amplitudes per sample
    const DevStage* stages;
    const DevTerm* terms;
    const DevExpr* exprs;
    const DevMat* mats;
    const DevComp* comps;    // composite matrices of the pass and their factor lists (pass-local matrix indices)
    const int32_t* facts;
    int n_comp, n_fact;
    // every qubit's first gate is a Hadamard (each hand-written map): the product state is the constant 2^(-n/2), its
    // factors over the low / high local bits and the chunk bits are these three numbers instead of n table look-ups
    int init_h;
    double init_lo, init_hi, init_chunk;
    int dep_bytes;               // size of the deposit tables in LDS (a multiple of 16)
    int h_load, h_store;         // the pass's first / last stage is a radix-16 Hadamard sweep that reads / writes HBM itself
    int n_stage, n_expr, n_mat, n_term;
    int staged;                                // the pass's DevExpr records (destinations of the angles) are copied to LDS
    int red_off;                               // fused-RDM scratch: offset in doubles from the end of the phase tables
    int L, h, lgT, n_qubits, first;
    uint32_t zero_in, zero_out;  // virtual-bit masks of still-|0> qubits (see Pass)
    int n_init;                  // first pass: leading ST_INIT stages (hoisted first gates)
    int direct;                  // the pass is product state + ONE diagonal stage: amplitudes go straight to HBM, no LDS image
    int last_direct;             // the last stage of the last pass is diagonal: its sweep stores to HBM instead of LDS
    int hoist;                   // fused first diagonal stage: 1 = tables built once per sample, 2 = ... as separate base tables (several chunks per sample)
    uint32_t live_chunks;        // chunk-index bits that are not known-zero after the pass
    int nlive;                   // popcount(live_chunks)
    int64_t n_items;             // N << nlive (sample, live chunk) pairs
    int nruns;
    uint32_t run_mask[kMaxRuns];
    int32_t run_shift[kMaxRuns];
    double* trace;               // phase-timing builds only (QK_SIM_TRACE): 16 doubles per workgroup
};

// Phase-timing builds (tools/sim_trace.sh compiles this file again with -DQK_SIM_TRACE into a separate library):
// thread 0 of every workgroup adds up the 100 MHz ticks between the marks below and leaves the sums in a table.
#ifdef QK_SIM_TRACE
#define QK_MARK(k) do { if (tid == 0) { const uint64_t t_ = wall_clock64(); tr_acc[k] += (double)(t_ - tr_last); tr_last = t_; } } while (0)
#define QK_MARK2(k) do { if (tid == 0) { const uint64_t t_ = wall_clock64(); tr_acc2[k] += (double)(t_ - tr_last); tr_last = t_; } } while (0)
#else
#define QK_MARK(k) do { } while (0)
#define QK_MARK2(k) do { } while (0)
#endif

// a diagonal term with its per-sample (cos, sin) resolved, 32 bytes in LDS
struct RTerm {
    double c, s;
    int32_t type, va, vb, pad;
};

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ double2 cfma(double2 a, double2 b, double2 c) {  // a*b + c
    return make_double2(fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y)));
}

__device__ __forceinline__ uint32_t deposit(uint32_t w, const SimArgs& a) {
    uint32_t g = 0;
    // not unrolled: the unrolled form's trip-count tests ("one run?", "odd number?") are hoisted out of the item
    // loop as booleans the register allocator then keeps in scratch memory (see the note at the item loop's head);
    // for the same reason the loop has no entry test (a pass always has at least one run)
    int r = 0;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    do {
        const uint32_t m = w & a.run_mask[r];
        const int sh = a.run_shift[r];
        g |= sh >= 0 ? (m << sh) : (m >> (-sh));
    } while (++r < a.nruns);
    return g;
}

// multiply f by the factor of one diagonal term for the amplitude with virtual index w
__device__ __forceinline__ void apply_term(const RTerm& t, uint32_t w, double& fr, double& fi) {
    const uint32_t ba = (w >> t.va) & 1u, bb = (w >> t.vb) & 1u;
    double s = t.s;
    bool on;
    if (t.type == TM_RZ) {            // diag(e^{-i th/2}, e^{+i th/2})
        on = true;
        s = ba ? s : -s;
    } else if (t.type == TM_P) {      // diag(1, e^{i lam})
        on = ba;
    } else if (t.type == TM_CZ) {     // (c, s) = (-1, 0) where both bits are set
        on = ba & bb;
    } else {                          // TM_ZZP: e^{i lam} where the two bits differ
        on = ba ^ bb;
    }
    if (on) {
        const double nr = fma(fr, t.c, -(fi * s));
        const double ni = fma(fr, s, fi * t.c);
        fr = nr;
        fi = ni;
    }
}

// The planner's tables (stages, matrix kinds) are written by the host before the launch and never by a kernel.  Read
// through the constant address space with wave-uniform indices they come in over the scalar data path (s_load,
// counted by lgkmcnt).  As ordinary global reads they were vector loads, and the s_waitcnt vmcnt(0) in front of
// their first use also sat out every store of the previous item still on its way to HBM -- the drain the item loop
// is arranged to compute under.
#define QK_CONST_AS __attribute__((address_space(4)))
typedef const int32_t QK_CONST_AS* ConstWords;
__device__ __forceinline__ ConstWords const_words(const void* p) { return (ConstWords)(uintptr_t)p; }

constexpr int kStageWords = (int)(sizeof(DevStage) / sizeof(int32_t));
constexpr int kStageXoWord = (int)(offsetof(DevStage, xo) / sizeof(int32_t));
__device__ __forceinline__ DevStage load_stage(ConstWords stages, int idx) {
    static_assert(sizeof(DevStage) == (6 + 2 * kMaxGroup + kMaxRowBits + 1 + 5) * sizeof(int32_t), "DevStage is read word by word");
    ConstWords w = stages + idx * kStageWords;
    constexpr int G = kMaxGroup, X = kStageXoWord;
    static_assert(X == 8 + 2 * G && offsetof(DevStage, t0) == (4 + 2 * G) * sizeof(int32_t), "field offsets below");
    DevStage st;      // (xo[] is read from memory where it is needed: indexing this copy puts the whole record in scratch)
    st.kind = w[0]; st.va = w[1]; st.vb = w[2]; st.ng = w[3];
#pragma unroll
    for (int k = 0; k < G; ++k) { st.gv[k] = w[4 + k]; st.gm[k] = w[4 + G + k]; }
    st.t0 = w[4 + 2 * G]; st.nLo = w[5 + 2 * G]; st.nHi = w[6 + 2 * G]; st.nX = w[7 + 2 * G];
#pragma unroll
    for (int k = 0; k <= kMaxRowBits; ++k) st.xo[k] = w[X + k];
    st.nLoD = w[X + kMaxRowBits + 1]; st.nHiD = w[X + kMaxRowBits + 2]; st.reuse = w[X + kMaxRowBits + 3];
    return st;
}

// The factor of one diagonal term as a complex number, (1, 0) where the term does not act: no branches.
__device__ __forceinline__ double2 term_factor(const RTerm& t, uint32_t w) {
    const uint32_t ba = (w >> t.va) & 1u, bb = (w >> t.vb) & 1u;
    const bool rz = t.type == TM_RZ;
    const bool on = rz || (t.type == TM_P ? ba != 0u : (t.type == TM_CZ ? (ba & bb) != 0u : (ba ^ bb) != 0u));
    const double s = (rz && !ba) ? -t.s : t.s;
    return make_double2(on ? t.c : 1.0, on ? s : 0.0);
}

// f *= the factors of terms t[0 .. n) at index w.  The phase tables are built by a few threads per sample while the
// rest of the workgroup waits, so what counts is the length of the dependent chain: the terms of four steps are read
// together (their LDS reads are independent of the products) and the product runs as two chains.
__device__ __forceinline__ void apply_terms(const RTerm* t, int n, uint32_t w, double& fr, double& fi) {
    double2 p = make_double2(fr, fi), q = make_double2(1.0, 0.0);
    int k = 0;
    for (; k + 3 < n; k += 4) {
        const RTerm t0 = t[k], t1 = t[k + 1], t2 = t[k + 2], t3 = t[k + 3];
        p = cmul(p, term_factor(t0, w));
        q = cmul(q, term_factor(t1, w));
        p = cmul(p, term_factor(t2, w));
        q = cmul(q, term_factor(t3, w));
    }
    for (; k < n; ++k) p = cmul(p, term_factor(t[k], w));
    p = cmul(p, q);
    fr = p.x;
    fi = p.y;
}

// f *= the product-state factors of the hoisted first gates whose bit lies in one half of the local bits
// (is_lo: virtual bits < h; else h <= bit < L), for the amplitude index w; reads batched as in apply_terms
__device__ __forceinline__ double2 init_factors(const int2* s_init, int n_init, const double2* mat, int L, int h, bool is_lo,
                                                uint32_t w, double2 f) {
    double2 q = make_double2(1.0, 0.0);
    const double2 one = make_double2(1.0, 0.0);
    auto use = [&](int2 g) { return g.x < L && (g.x < h) == is_lo; };
    auto at = [&](int2 g) { return 4 * g.y + 2 * (int)((w >> (g.x & 31)) & 1u); };
    int k = 0;
    for (; k + 3 < n_init; k += 4) {
        const int2 g0 = s_init[k], g1 = s_init[k + 1], g2 = s_init[k + 2], g3 = s_init[k + 3];
        const double2 m0 = mat[at(g0)], m1 = mat[at(g1)], m2 = mat[at(g2)], m3 = mat[at(g3)];
        f = cmul(f, use(g0) ? m0 : one);
        q = cmul(q, use(g1) ? m1 : one);
        f = cmul(f, use(g2) ? m2 : one);
        q = cmul(q, use(g3) ? m3 : one);
    }
    for (; k < n_init; ++k) {
        const int2 g = s_init[k];
        const double2 m = mat[at(g)];
        f = cmul(f, use(g) ? m : one);
    }
    return cmul(f, q);
}

// a wave-uniform double as a scalar-register pair (values read from LDS arrive in vector registers)
__device__ __forceinline__ double uniform_f64(double x) {
    const uint64_t u = (uint64_t)__double_as_longlong(x);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

template <int NG, int IPT>
__device__ __forceinline__ void sweep_u1(double2* state, const double2* mat, ConstWords mats, const DevStage& st,
                                         uint32_t nloc, int tid, int T) {
    // NG single-qubit gates on distinct local qubits in one sweep: each thread holds the 2^NG
    // amplitudes that differ in the target bits and applies the gates in registers.
    // form: 0 = real matrix (x, ry), 1 = real diagonal + imaginary off-diagonal (rx), 2 = general (sx),
    //       3 = hadamard (adds only; the factors 1/sqrt 2 of the group are applied once at the end)
    // The 2^NG amplitude offsets depend on the stage only (wave-uniform: scalar registers), so an item costs
    // NG bit insertions and one OR per amplitude.
    // The workgroup runs two waves per SIMD at most (LDS holds two states per CU), so nothing hides the latency of
    // an LDS read but the thread's own independent work: a trip handles IPT items and issues all of their reads
    // before the first butterfly.  The matrices are wave-uniform and sit in scalar registers (they come out of
    // LDS as vector registers: 48 of them in a radix-8 sweep, which is what made two items per trip spill).
    double2 m[NG][4];
    uint32_t lm[NG], off[1 << NG];
    int form[NG];
    int nh = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int gate = mats[2 * st.gm[g]];      // DevMat::gate
        form[g] = gate == QK_H ? 3 : ((gate == QK_X || gate == QK_RY) ? 0 : (gate == QK_RX ? 1 : 2));
        nh += gate == QK_H ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double2 e = mat[4 * st.gm[g] + k];
            m[g][k] = make_double2(uniform_f64(e.x), uniform_f64(e.y));
        }
        lm[g] = (1u << st.gv[g]) - 1u;
    }
#pragma unroll
    for (int k = 0; k < (1 << NG); ++k) {
        uint32_t o = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) o |= (((uint32_t)k >> g) & 1u) << st.gv[g];
        off[k] = o;
    }
    // 2^(-nh/2), chosen among constants (a running product of 1/sqrt 2 ends up as a vector register the allocator spills)
    const double hscale = nh == 1 ? 0.70710678118654752440 : (nh == 2 ? 0.5 : (nh == 3 ? 0.35355339059327376220 : 1.0));
    auto base_of = [&](uint32_t i) {
        uint32_t b = i;
#pragma unroll
        for (int g = 0; g < NG; ++g) b = ((b & ~lm[g]) << 1) | (b & lm[g]);   // target bits ascending
        return b;
    };
    auto butterflies = [&](double2 (&v)[1 << NG]) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (form[g] == 3) {
#pragma unroll
                for (int k = 0; k < (1 << NG); ++k)
                    if (!((k >> g) & 1)) {
                        const int k1 = k | (1 << g);
                        const double2 a0 = v[k], a1 = v[k1];
                        v[k] = make_double2(a0.x + a1.x, a0.y + a1.y);
                        v[k1] = make_double2(a0.x - a1.x, a0.y - a1.y);
                    }
            } else if (form[g] == 0) {
#pragma unroll
                for (int k = 0; k < (1 << NG); ++k)
                    if (!((k >> g) & 1)) {
                        const int k1 = k | (1 << g);
                        const double2 a0 = v[k], a1 = v[k1];
                        v[k] = make_double2(fma(m[g][1].x, a1.x, m[g][0].x * a0.x), fma(m[g][1].x, a1.y, m[g][0].x * a0.y));
                        v[k1] = make_double2(fma(m[g][3].x, a1.x, m[g][2].x * a0.x), fma(m[g][3].x, a1.y, m[g][2].x * a0.y));
                    }
            } else if (form[g] == 1) {
#pragma unroll
                for (int k = 0; k < (1 << NG); ++k)
                    if (!((k >> g) & 1)) {
                        const int k1 = k | (1 << g);
                        const double2 a0 = v[k], a1 = v[k1];
                        // (i y)(a.x + i a.y) = -y a.y + i y a.x
                        v[k] = make_double2(fma(-m[g][1].y, a1.y, m[g][0].x * a0.x), fma(m[g][1].y, a1.x, m[g][0].x * a0.y));
                        v[k1] = make_double2(fma(-m[g][2].y, a0.y, m[g][3].x * a1.x), fma(m[g][2].y, a0.x, m[g][3].x * a1.y));
                    }
            } else {
#pragma unroll
                for (int k = 0; k < (1 << NG); ++k)
                    if (!((k >> g) & 1)) {
                        const int k1 = k | (1 << g);
                        const double2 a0 = v[k], a1 = v[k1];
                        v[k] = cfma(m[g][1], a1, cmul(m[g][0], a0));
                        v[k1] = cfma(m[g][3], a1, cmul(m[g][2], a0));
                    }
            }
        }
        if (nh != 0) {
#pragma unroll
            for (int k = 0; k < (1 << NG); ++k) v[k] = make_double2(v[k].x * hscale, v[k].y * hscale);
        }
    };
    const uint32_t items = nloc >> NG;     // a power of two, like T: every trip of the loop is full or not taken
    for (uint32_t i = tid; i < items; i += IPT * T) {
        uint32_t b[IPT];
        double2 v[IPT][1 << NG];
#pragma unroll
        for (int p = 0; p < IPT; ++p) b[p] = base_of(i + p * T);
#pragma unroll
        for (int p = 0; p < IPT; ++p)
#pragma unroll
            for (int k = 0; k < (1 << NG); ++k) v[p][k] = state[b[p] | off[k]];
#pragma unroll
        for (int p = 0; p < IPT; ++p) butterflies(v[p]);
#pragma unroll
        for (int p = 0; p < IPT; ++p)
#pragma unroll
            for (int k = 0; k < (1 << NG); ++k) state[b[p] | off[k]] = v[p][k];
    }
}

// Four Hadamards on distinct local qubits in one sweep: a thread holds the 16 amplitudes that differ in the target
// bits (one item per trip: 64 registers of amplitudes) -- three sweeps for a layer of 12 Hadamards.
// When the targets are bits 0..3 a thread's 16 amplitudes are 256 consecutive bytes, i.e. every lane of a bank group
// would be on the same banks in each of the 16 reads and writes.  There lane t reads amplitude k ^ s (s = t & 15) into
// register k and writes register i to position i ^ s -- 16 lanes, 16 different slots -- and the transform is repaired
// with signs, no data movement:  WHT(x)[i ^ s] = (-1)^(i.s + s.s) * sum_k (-1)^(i.k) [(-1)^(s.k) x[k ^ s]]   (a.b = parity
// of a & b), i.e. flip the sign of input register k where parity(s & k), transform, flip output register i where
// parity(s & i) ^ parity(s): XORs on the high words with per-lane masks.
__device__ __forceinline__ void sweep_h16(double2* state, const DevStage& st, uint32_t nloc, int tid, int T) {
    uint32_t lm[4], off[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) lm[g] = (1u << st.gv[g]) - 1u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint32_t o = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) o |= (((uint32_t)k >> g) & 1u) << st.gv[g];
        off[k] = o;
    }
    const bool low = st.gv[3] == 3;     // ascending and distinct: the targets are bits 0, 1, 2, 3
    auto flip = [](double x, uint32_t m) { return __longlong_as_double(__double_as_longlong(x) ^ (long long)((uint64_t)m << 32)); };
    const uint32_t items = nloc >> 4;
    for (uint32_t i = tid; i < items; i += T) {
        uint32_t b = i;
#pragma unroll
        for (int g = 0; g < 4; ++g) b = ((b & ~lm[g]) << 1) | (b & lm[g]);
        const uint32_t sw = low ? (i & 15u) : 0u;
        double2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = state[b | (off[k] ^ sw)];
        uint32_t mask[16];
        if (low) {
            const uint32_t b0 = (sw & 1u) << 31, b1 = (sw & 2u) << 30, b2 = (sw & 4u) << 29, b3 = (sw & 8u) << 28;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                mask[k] = ((k & 1) ? b0 : 0u) ^ ((k & 2) ? b1 : 0u) ^ ((k & 4) ? b2 : 0u) ^ ((k & 8) ? b3 : 0u);
#pragma unroll
            for (int k = 1; k < 16; ++k) v[k] = make_double2(flip(v[k].x, mask[k]), flip(v[k].y, mask[k]));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (!((k >> g) & 1)) {
                    const int k1 = k | (1 << g);
                    const double2 a0 = v[k], a1 = v[k1];
                    v[k] = make_double2(a0.x + a1.x, a0.y + a1.y);
                    v[k1] = make_double2(a0.x - a1.x, a0.y - a1.y);
                }
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = make_double2(v[k].x * 0.25, v[k].y * 0.25);
        if (low) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t m = mask[k] ^ mask[15];
                v[k] = make_double2(flip(v[k].x, m), flip(v[k].y, m));
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) state[b | (off[k] ^ sw)] = v[k];
    }
}

// The same sweep at either end of a pass.  ld(l) / stf(l, v): where the amplitudes come from and go to -- the LDS image, or (a pass's first / last sweep, targets
// above bit 3 so that 16 lanes still cover 256 consecutive bytes) the statevector in HBM directly.
template <class Ld, class St>
__device__ __forceinline__ void sweep_h16_io(const DevStage& st, uint32_t nloc, int tid, int T, Ld ld, St stf) {
    uint32_t lm[4], off[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) lm[g] = (1u << st.gv[g]) - 1u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint32_t o = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) o |= (((uint32_t)k >> g) & 1u) << st.gv[g];
        off[k] = o;
    }
    const bool low = st.gv[3] == 3;     // ascending and distinct: the targets are bits 0, 1, 2, 3
    auto flip = [](double x, uint32_t m) { return __longlong_as_double(__double_as_longlong(x) ^ (long long)((uint64_t)m << 32)); };
    const uint32_t items = nloc >> 4;
    for (uint32_t i = tid; i < items; i += T) {
        uint32_t b = i;
#pragma unroll
        for (int g = 0; g < 4; ++g) b = ((b & ~lm[g]) << 1) | (b & lm[g]);
        const uint32_t sw = low ? (i & 15u) : 0u;
        double2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = ld(b | (off[k] ^ sw));
        uint32_t mask[16];
        if (low) {
            const uint32_t b0 = (sw & 1u) << 31, b1 = (sw & 2u) << 30, b2 = (sw & 4u) << 29, b3 = (sw & 8u) << 28;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                mask[k] = ((k & 1) ? b0 : 0u) ^ ((k & 2) ? b1 : 0u) ^ ((k & 4) ? b2 : 0u) ^ ((k & 8) ? b3 : 0u);
#pragma unroll
            for (int k = 1; k < 16; ++k) v[k] = make_double2(flip(v[k].x, mask[k]), flip(v[k].y, mask[k]));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (!((k >> g) & 1)) {
                    const int k1 = k | (1 << g);
                    const double2 a0 = v[k], a1 = v[k1];
                    v[k] = make_double2(a0.x + a1.x, a0.y + a1.y);
                    v[k1] = make_double2(a0.x - a1.x, a0.y - a1.y);
                }
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = make_double2(v[k].x * 0.25, v[k].y * 0.25);
        if (low) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t m = mask[k] ^ mask[15];
                v[k] = make_double2(flip(v[k].x, m), flip(v[k].y, m));
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) stf(b | (off[k] ^ sw), v[k]);
    }
}

template <int NG>
__device__ __forceinline__ void sweep_u1_any(double2* state, const double2* mat, ConstWords mats, const DevStage& st,
                                             uint32_t nloc, int tid, int T) {
    if ((nloc >> NG) >= 2u * (uint32_t)T) sweep_u1<NG, 2>(state, mat, mats, st, nloc, tid, T);
    else sweep_u1<NG, 1>(state, mat, mats, st, nloc, tid, T);
}

// The per-sample gate angles.  A pre-pass (angle_table_kernel) evaluates them for all samples, one LANE per sample:
// every lane of a wave runs the same postfix program (validated on the host: qk_circuit_create) on its own feature
// row, so the interpreter's control flow is uniform and 64 samples share each of its instructions -- inside the
// simulator the same work was one thread per angle of ONE sample, a single wave of lanes diverging over the
// interpreter's switch while the rest of the workgroup waited (3 us of a 31 us item for the ZZ map at 12 qubits, 9 us
// for a GA genome with its asin: tools/sim_trace.py).  The table holds (cos, sin) of angle * scale as [expr][sample].
__global__ void __launch_bounds__(256) angle_table_kernel(const DevExpr* exprs, int n_expr, const qk_instr* prog,
                                                          const double* consts, const double* X, int64_t ldx, int64_t N,
                                                          double2* cs) {
    // a wave = 64 consecutive samples x one angle (blockIdx.y * 4 + wave): small batches still fill the chip
    const int e = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * 4 + (threadIdx.x >> 6)));
    const int64_t smp = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    if (e >= n_expr || smp >= N) return;
    const int start = __builtin_amdgcn_readfirstlane(exprs[e].start), len = __builtin_amdgcn_readfirstlane(exprs[e].len);
    const double angle = run_postfix(prog, start, len, consts, X + smp * ldx);
    double sn, cn;
    sincos_cw(angle * exprs[e].scale, &sn, &cn);   // the routine the wave-resident kernel uses: one set of angles per circuit
    cs[(int64_t)e * N + smp] = make_double2(cn, sn);
}

// (cos, sin) of one angle into the diagonal term or the 2x2 matrix it feeds
__device__ __forceinline__ void scatter_angle(const DevExpr& ex, double cn, double sn, RTerm* rt, double2* mat) {
    if (ex.dst >= 0) {
        rt[ex.dst].c = cn;
        rt[ex.dst].s = sn;
    } else {  // rx / ry / rz from cos, sin of theta/2, p from those of lambda (rz, p: only as factors of a composite)
        const int m = ~ex.dst;
        if (ex.gate == QK_RX) {
            mat[4 * m + 0] = make_double2(cn, 0); mat[4 * m + 1] = make_double2(0, -sn);
            mat[4 * m + 2] = make_double2(0, -sn); mat[4 * m + 3] = make_double2(cn, 0);
        } else if (ex.gate == QK_RZ) {
            mat[4 * m + 0] = make_double2(cn, -sn); mat[4 * m + 1] = make_double2(0, 0);
            mat[4 * m + 2] = make_double2(0, 0); mat[4 * m + 3] = make_double2(cn, sn);
        } else if (ex.gate == QK_P) {
            mat[4 * m + 0] = make_double2(1, 0); mat[4 * m + 1] = make_double2(0, 0);
            mat[4 * m + 2] = make_double2(0, 0); mat[4 * m + 3] = make_double2(cn, sn);
        } else {
            mat[4 * m + 0] = make_double2(cn, 0); mat[4 * m + 1] = make_double2(-sn, 0);
            mat[4 * m + 2] = make_double2(sn, 0); mat[4 * m + 3] = make_double2(cn, 0);
        }
    }
}

// MAXG = 3: up to 256 VGPRs, two waves per SIMD (radix-8 sweeps, two items per trip); MAXG = 2: at most 128 VGPRs, four
// waves per SIMD (radix-4 sweeps; without the angle interpreter, which went into angle_table_kernel, it needs 102).
// RDM = true: the fused simulate + reduced-density-matrix variant (qk_simulate_rdm1_dev); the reduction
// holds 16 amplitudes per thread, so it gets the 256-VGPR budget (2 waves/SIMD) of its own instantiation.
// The workgroup's waves talk to each other through LDS only (statevectors in global memory are written by one
// pass and read by the next, a separate launch), so its barriers wait for LDS traffic alone.  __syncthreads()
// also waits for every outstanding global access of the wave (s_waitcnt vmcnt(0)): the first barrier of an item
// then sat out the drain of the previous item's stores -- about 10 us of a 17-37 us item at 12 qubits
// (tools/sim_trace.py) -- instead of computing under it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The batched loops below name their in-flight values one by one: as arrays (even fully unrolled ones) they stay in
// scratch memory in this kernel -- the backend's alloca promotion has no register budget left at 256 VGPRs.
#define QK_REP4(M) M(0) M(1) M(2) M(3)
#define QK_REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

template <int MAXG, bool RDM>
__global__ void __launch_bounds__(512, (MAXG == 3 || RDM) ? 2 : 4) sim_pass_kernel(SimArgs a) {
    extern __shared__ double2 lds[];
    const int L = a.L, h = a.h;
    const uint32_t nloc = 1u << L;
    double2* state = lds;
    RTerm* rt = (RTerm*)(lds + (a.direct ? 0u : nloc));     // n_term resolved diagonal terms (direct mode keeps no state in LDS)
    double2* mat = (double2*)(rt + a.n_term);               // n_mat * 4 complex entries
    double2* tlo = mat + 4 * a.n_mat;                       // phase table over local bits < h
    double2* thi = tlo + (1u << h);                         // phase table over local bits >= h
    double2* rat = thi + (1u << (L - h));                   // [row bit][column]: cross-term ratio of setting that bit
    double2* tables_end = rat + ((uint32_t)(L - h) << h);
    // hoist == 2: the chunk-independent part of tlo / thi has its own tables; hoist == 1 (one chunk per
    // sample): it IS tlo / thi
    double2* blo = a.hoist == 2 ? tables_end : tlo;
    double2* bhi = a.hoist == 2 ? blo + (1u << h) : thi;
    // deposit(l) for every local index, as two tables (the map is linear over disjoint bits): dep_lo[l & (2^h - 1)] |
    // dep_hi[l >> h].  Evaluated per amplitude, deposit() is a loop over the pass's bit runs whose masks and shifts come
    // from kernel-argument memory by scalar loads -- a chain of scalar-load latencies per stored amplitude.
    uint32_t* dep_lo = (uint32_t*)(a.hoist == 2 ? (bhi + (1u << (L - h))) : tables_end);
    uint32_t* dep_hi = dep_lo + (1u << h);
    double* after_tables = (double*)((char*)dep_lo + a.dep_bytes);
    DevExpr* s_expr = (DevExpr*)after_tables;              // staged DevExpr records (a.staged)
    int2* s_init = (int2*)(s_expr + (a.staged ? a.n_expr : 0));   // (virtual bit, matrix) of the hoisted first gates
    // composite matrices of the pass and their factor lists, read per thread at the start of a sample: in LDS when
    // staged (see const_words above for why), else where the host put them
    const DevComp* s_comp = a.staged ? (const DevComp*)(s_init + a.n_init) : a.comps;
    const int32_t* s_fact = a.staged ? (const int32_t*)((const DevComp*)(s_init + a.n_init) + a.n_comp) : a.facts;
    double* s_red = after_tables + a.red_off;              // fused-RDM partial sums
    const int tid = threadIdx.x;
    const int T = blockDim.x;
    static_assert(sizeof(DevMat) == 2 * sizeof(int32_t), "DevMat is read word by word");
    const ConstWords c_stages = const_words(a.stages), c_mats = const_words(a.mats);

    // ---- once per workgroup: everything that does not depend on the sample -----------------
    for (int t = tid; t < a.n_term; t += T) {
        const DevTerm d = a.terms[t];
        RTerm r;
        r.c = -1.0; r.s = 0.0;            // TM_CZ; parametric terms get their (cos, sin) per item
        r.type = d.type; r.va = d.va; r.vb = d.vb; r.pad = 0;
        rt[t] = r;
    }
    for (int m = tid; m < a.n_mat; m += T) {
        const int gate = a.mats[m].gate;
        const double r = 0.70710678118654752440;
        double2 m00 = make_double2(0, 0), m01 = m00, m10 = m00, m11 = m00;
        if (gate == QK_H) {
            m00 = make_double2(r, 0); m01 = make_double2(r, 0); m10 = make_double2(r, 0); m11 = make_double2(-r, 0);
        } else if (gate == QK_X) {
            m01 = make_double2(1, 0); m10 = make_double2(1, 0);
        } else if (gate == QK_SX) {
            m00 = make_double2(0.5, 0.5); m01 = make_double2(0.5, -0.5);
            m10 = make_double2(0.5, -0.5); m11 = make_double2(0.5, 0.5);
        }
        mat[4 * m + 0] = m00; mat[4 * m + 1] = m01; mat[4 * m + 2] = m10; mat[4 * m + 3] = m11;
    }
    for (int k = tid; k < a.n_init; k += T) s_init[k] = make_int2(a.stages[k].va, a.stages[k].gm[0]);
    for (uint32_t k = tid; k < (1u << h) + (1u << (L - h)); k += T)
        dep_lo[k] = k < (1u << h) ? deposit(k, a) : deposit((k - (1u << h)) << h, a);
    if (a.staged) {
        for (int e = tid; e < a.n_expr; e += T) s_expr[e] = a.exprs[e];
        for (int k = tid; k < a.n_comp; k += T) ((DevComp*)s_comp)[k] = a.comps[k];
        for (int k = tid; k < a.n_fact; k += T) ((int32_t*)s_fact)[k] = a.facts[k];
    }
    // Persistent workgroups: each owns a contiguous range of (sample, chunk) items, so the launch
    // cost is paid once per CU slot and the chunks of one sample follow each other: the angle
    // programs (the longest dependent chain of an item) run once per sample, not once per chunk.
    const int64_t per_wg = (a.n_items + gridDim.x - 1) / gridDim.x;
    const int64_t item_begin = (int64_t)blockIdx.x * per_wg;
    const int64_t item_end = item_begin + per_wg < a.n_items ? item_begin + per_wg : a.n_items;
    lds_barrier();
    // the angles of the workgroup's first sample; every later sample's are fetched while the sample before it is
    // computed and dealt out after its last stage
    if (item_begin < item_end) {
        for (int e = tid; e < a.n_expr; e += T) {
            const double2 v = a.cs[(int64_t)e * a.N + (item_begin >> a.nlive)];
            scatter_angle(a.staged ? s_expr[e] : a.exprs[e], v.x, v.y, rt, mat);
        }
    }
    lds_barrier();

    // Chunks whose amplitudes are all still zero after the pass (and a later pass knows) are not
    // items at all.  In steady state the only vector-memory traffic of a first pass is the store of
    // the chunk and the prefetch of the next sample's angles, so the stores of one item drain while the
    // next is computed.
    // A diagonal stage right after the hoisted first gates (every Z / ZZ map) takes the product
    // state as two more factors of its tables: the state is written once, phases included.
    // (readfirstlane: the value comes from a vector load, and a per-lane boolean that lives across the item loop
    // ends up in scratch memory)
    const bool fuse_init = a.first && a.n_init > 0 && a.n_init < a.n_stage &&
                           __builtin_amdgcn_readfirstlane(a.stages[a.n_init].kind) == ST_DIAG;
    // Several chunks per sample: what those tables owe to local bits only (the product state's factors,
    // the terms without a chunk bit, the cross-term ratios) is computed once per sample; a chunk
    // multiplies in its scalar and the few terms that read a chunk bit.
    const bool hoist = a.hoist != 0;    // = fuse_init, decided by the host with the LDS layout
    int64_t angles_of = -1;     // the sample whose (cos, sin) values sit in rt / mat
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef QK_SIM_TRACE
    double tr_acc2[4] = {0, 0, 0, 0};   // finer split of the diagonal stage's table phase: 0 stage record, 1 tables, 2 barrier
    double tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 0 load + angles + tables, 1 product state, 2 diag tables, 3 diag sweep, 4 u1, 5 cx / ecr, 6 store, 7 items
    uint64_t tr_last = wall_clock64();
    const uint64_t tr_begin = tr_last;
    uint64_t tr_angles = 0, tr_pre = 0;
#endif
    for (int64_t item = item_begin; item < item_end; ++item) {
        const int64_t next = item + 1;
        const bool fetch = next < item_end && (next >> a.nlive) != (item >> a.nlive);
        // The addresses of the angle prefetch are recomputed per item from an opaque copy of the thread index:
        // hoisted out of the loop they end up spilled, and a reload from scratch memory at the top of an item
        // waits (memory operations return in order) for every store of the previous item -- the drain this loop
        // is built to compute under (10 us of a 17-37 us item at 12 qubits: tools/sim_trace.py).
        // (the copy is rebuilt from the lane counter and the wave's number in a scalar register: held in a vector
        // register across the loop, the thread index itself is what gets spilled)
        uint32_t ones = ~0u;
        asm volatile("" : "+s"(ones));
        int tid_x = (wave_s << 6) | (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
        asm volatile("" : "+v"(tid_x));
        // ... and for the same reason every per-thread index of an item is derived from that copy: whatever the
        // compiler can prove invariant (column numbers, LDS addresses of this thread's table entries) it computes
        // before the loop, keeps across it, and -- at 168 or 256 registers -- spills
        const int tid = tid_x;
        const bool have_next = fetch && tid_x < a.n_expr;
        double2 cs_next = make_double2(1.0, 0.0);
        if (have_next) cs_next = a.cs[(int64_t)tid_x * a.N + (next >> a.nlive)];   // in flight until the end of the item
        uint32_t chunk = 0;
        {
            uint32_t ci = (uint32_t)(item & ((1u << a.nlive) - 1u)), m = a.live_chunks;
            for (int b = 0; m; ++b, m >>= 1)
                if (m & 1u) { chunk |= (ci & 1u) << b; ci >>= 1; }
        }
        const int64_t sample = item >> a.nlive;
        const uint32_t whi = chunk << L;

        // ---- chunk load -----------------------------------------------------------------
        // amplitudes with a bit of a still-|0> qubit set are zero by construction: never loaded
        const uint32_t chunk_phys = deposit(whi, a);
        auto dep_local = [&](uint32_t l) { return dep_lo[l & ((1u << h) - 1u)] | dep_hi[l >> h]; };
        double2* gpsi = a.psi + sample * a.D;
        auto store_direct = [&](uint32_t l, double2 v) {     // direct mode: the one place an amplitude is written
            const int64_t at = chunk_phys | dep_local(l);
            if (a.psi32) (a.psi32 + sample * a.D)[at] = make_float2((float)v.x, (float)v.y);
            else gpsi[at] = v;
        };
        if (!a.first && !a.h_load) {
            if (whi & a.zero_in) {
                for (uint32_t l = tid; l < nloc; l += T) state[l] = make_double2(0.0, 0.0);
            } else {
                // eight loads in flight per thread (the loop as written waits for each load before it issues the next
                // one: the LDS store may alias nothing the compiler can see, but it keeps the order)
                uint32_t l = tid;
                for (; l + 7u * T < nloc; l += 8u * T) {
#define QK_LD(k) const uint32_t l##k = l + k##u * T; \
                 const double2 v##k = (l##k & a.zero_in) ? make_double2(0.0, 0.0) : gpsi[chunk_phys | dep_local(l##k)];
#define QK_ST(k) state[l##k] = v##k;
                    QK_REP8(QK_LD)
                    QK_REP8(QK_ST)
#undef QK_LD
#undef QK_ST
                }
                for (; l < nloc; l += T)
                    state[l] = (l & a.zero_in) ? make_double2(0.0, 0.0) : gpsi[chunk_phys | dep_local(l)];
            }
        }
        // ---- per-sample gate angles -> (cos, sin) into their term or matrix ---------------
#ifdef QK_SIM_TRACE
        if (tid == 0) tr_pre += wall_clock64() - tr_last;
#endif
        if (sample != angles_of) {
            // (its angles are in place: dealt out before the loop or at the end of the previous sample's last item)
            angles_of = sample;
#ifdef QK_SIM_TRACE
            if (tid == 0) tr_angles += wall_clock64() - tr_last;
#endif
            if (a.n_comp) {     // runs of single-qubit gates on one qubit: one thread multiplies a run's factors
                for (int k = tid_x; k < a.n_comp; k += T) {
                    const DevComp cc = s_comp[k];
                    const int32_t* f = s_fact + cc.first;
                    double2 m0 = mat[4 * f[0]], m1 = mat[4 * f[0] + 1], m2 = mat[4 * f[0] + 2], m3 = mat[4 * f[0] + 3];
                    for (int j = 1; j < cc.count; ++j) {     // M <- G M
                        const double2* g = mat + 4 * f[j];
                        const double2 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
                        const double2 n0 = cfma(g1, m2, cmul(g0, m0)), n1 = cfma(g1, m3, cmul(g0, m1));
                        const double2 n2 = cfma(g3, m2, cmul(g2, m0)), n3 = cfma(g3, m3, cmul(g2, m1));
                        m0 = n0; m1 = n1; m2 = n2; m3 = n3;
                    }
                    mat[4 * cc.dst] = m0; mat[4 * cc.dst + 1] = m1; mat[4 * cc.dst + 2] = m2; mat[4 * cc.dst + 3] = m3;
                }
            }
            if (hoist) {
                lds_barrier();
                const DevStage st = load_stage(c_stages, a.n_init);
                const RTerm* tl = rt + st.t0;
                const RTerm* th = tl + st.nLo;
                const RTerm* tx = th + st.nHi;
                const uint32_t nlo = 1u << h, nhi = 1u << (L - h);
                const uint32_t n_rat = st.nX ? (nlo * (uint32_t)(L - h)) : 0u;
                // one chunk per sample (hoist == 1): these ARE the stage's tables, chunk-bit terms and
                // the chunk's scalar included
                const bool all = a.hoist == 1;
                const int n_tl = all ? st.nLo : st.nLo - st.nLoD, n_th = all ? st.nHi : st.nHi - st.nHiD;
                double2 gs = make_double2((all && (whi & a.zero_in)) ? 0.0 : 1.0, 0.0);
                if (all && a.init_h) gs.x = a.init_chunk;
                else if (all)
                    for (int q = 0; q < a.n_init; ++q) {
                        const int2 g = s_init[q];
                        if (g.x >= L) gs = cmul(gs, mat[4 * g.y + 2 * ((whi >> g.x) & 1u)]);
                    }
                for (uint32_t j = (uint32_t)tid_x; j < n_rat + nlo + nhi; j += T) {
                    double fr = 1.0, fi = 0.0;
                    if (j < n_rat) {
                        const uint32_t c = j & (nlo - 1u);
                        const uint32_t ra = __builtin_amdgcn_readfirstlane(j >> h);
                        const uint32_t w1 = c | (1u << (h + ra));
                        double gr = 1.0, gi = 0.0;
                        for (int k = (c_stages + a.n_init * kStageWords + kStageXoWord)[ra], k1 = (c_stages + a.n_init * kStageWords + kStageXoWord)[ra + 1]; k < k1; ++k) {
                            apply_term(tx[k], w1, fr, fi);
                            apply_term(tx[k], c, gr, gi);
                        }
                        rat[j] = make_double2(fma(fr, gr, fi * gi), fma(fi, gr, -(fr * gi)));
                    } else {
                        const bool is_lo = j < n_rat + nlo;
                        const uint32_t e = j - n_rat - (is_lo ? 0u : nlo);
                        const uint32_t wl = is_lo ? e : (e << h);
                        const uint32_t w = all ? (whi | wl) : wl;
                        double2 f = (wl & a.zero_in) ? make_double2(0.0, 0.0) : (is_lo ? gs : make_double2(1.0, 0.0));
                        if (a.init_h) { const double sc = is_lo ? a.init_lo : a.init_hi; f = make_double2(f.x * sc, f.y * sc); }
                        else f = init_factors(s_init, a.n_init, mat, L, h, is_lo, w, f);
                        fr = f.x; fi = f.y;
                        if (is_lo) {
                            apply_terms(tl, n_tl, w, fr, fi);
                            apply_terms(tx, st.nX, w, fr, fi);   // X(0, c)
                            blo[e] = make_double2(fr, fi);
                        } else {
                            apply_terms(th, n_th, w, fr, fi);
                            bhi[e] = make_double2(fr, fi);
                        }
                    }
                }
            }
        }
        lds_barrier();
        QK_MARK(0);

        // ---- first pass: the product state of the hoisted first gates -----------------------
        // amplitude(w) = prod_q U_q[w_q][0] over the initialised qubits, 0 where a bit of an
        // uninitialised qubit is set; factored over the column / row halves like a diagonal stage
        if (a.hoist == 2) {
            const DevStage st = load_stage(c_stages, a.n_init);
            const RTerm* tld = rt + st.t0 + (st.nLo - st.nLoD);
            const RTerm* thd = rt + st.t0 + st.nLo + (st.nHi - st.nHiD);
            const uint32_t nlo = 1u << h, nhi = 1u << (L - h);
            double2 gs = make_double2((whi & a.zero_in) ? 0.0 : 1.0, 0.0);
            if (a.init_h) gs.x = a.init_chunk;
            else
                for (int s = 0; s < a.n_init; ++s) {
                    const int2 g = s_init[s];
                    if (g.x >= L) gs = cmul(gs, mat[4 * g.y + 2 * ((whi >> g.x) & 1u)]);
                }
            for (uint32_t j = tid; j < nlo + nhi; j += T) {
                const bool is_lo = j < nlo;
                double2 f = is_lo ? cmul(blo[j], gs) : bhi[j - nlo];
                const uint32_t w = whi | (is_lo ? j : ((j - nlo) << h));
                if (is_lo) {
                    apply_terms(tld, st.nLoD, w, f.x, f.y);
                    tlo[j] = f;
                } else {
                    apply_terms(thd, st.nHiD, w, f.x, f.y);
                    thi[j - nlo] = f;
                }
            }
            lds_barrier();
        } else if (a.first && !hoist) {
            const uint32_t nlo = 1u << h, nhi = 1u << (L - h);
            double2 gs = make_double2((whi & a.zero_in) ? 0.0 : 1.0, 0.0);
            if (a.init_h) gs.x = a.init_chunk;
            else
                for (int s = 0; s < a.n_init; ++s) {
                    const int2 g = s_init[s];
                    if (g.x >= L) gs = cmul(gs, mat[4 * g.y + 2 * ((whi >> g.x) & 1u)]);
                }
            for (uint32_t j = tid; j < nlo + nhi; j += T) {
                const bool is_lo = j < nlo;
                const uint32_t w = is_lo ? j : ((j - nlo) << h);
                double2 f = (w & a.zero_in) ? make_double2(0.0, 0.0) : (is_lo ? gs : make_double2(1.0, 0.0));
                if (a.init_h) { const double sc = is_lo ? a.init_lo : a.init_hi; f = make_double2(f.x * sc, f.y * sc); }
                else f = init_factors(s_init, a.n_init, mat, L, h, is_lo, w, f);
                (is_lo ? tlo[j] : thi[j - nlo]) = f;
            }
            lds_barrier();
            if (!fuse_init) {
                uint32_t l = tid;
                if (T >= (int)nlo) {     // l & (nlo - 1) is the same in every trip: one read of the column factor,
                    const double2 fc = tlo[tid & (nlo - 1u)];
                    for (; l + 3u * T < nloc; l += 4u * T) {     // four row factors read, then four amplitudes written
#define QK_LD(k) const double2 r##k = thi[(l + k##u * T) >> h];
#define QK_ST(k) state[l + k##u * T] = cmul(fc, r##k);
                        QK_REP4(QK_LD)
                        QK_REP4(QK_ST)
#undef QK_LD
#undef QK_ST
                    }
                    for (; l < nloc; l += T) state[l] = cmul(fc, thi[l >> h]);
                }
                for (; l < nloc; l += T) state[l] = cmul(tlo[l & (nlo - 1u)], thi[l >> h]);
                lds_barrier();
            }
        }

        QK_MARK(1);
        // ---- stages -----------------------------------------------------------------------
        for (int s = a.first ? a.n_init : 0; s < a.n_stage; ++s) {
            const DevStage st = load_stage(c_stages, s);
            if (st.kind == ST_DIAG) {
                // Local bits < h are the column c of an amplitude, the rest its row r:
                //   phase(r, c) = tlo[c] * thi[r] * X(r, c),
                // X = the product of the terms with one bit in each half.  Each of those has ONE row
                // bit, so X(r, c) = X(0, c) * prod_{bits a set in r} rat[a][c]: the tables cost
                // (2^h + 2^(L-h)) * terms instead of 2^L * terms, and a thread that walks rows in
                // Gray-code order updates X with one complex multiplication per amplitude.
                const RTerm* tl = rt + st.t0;
                const RTerm* th = tl + st.nLo;
                const RTerm* tx = th + st.nHi;
                const int nrb = L - h;
                const uint32_t nlo = 1u << h, nhi = 1u << nrb;
                const uint32_t n_rat = st.nX ? (nlo * (uint32_t)nrb) : 0u;
                const bool fused = fuse_init && s == a.n_init;   // tlo / thi hold the product state's factors
                const bool to_hbm = a.direct || (a.last_direct && s == a.n_stage - 1);   // this sweep produces the final amplitudes
                // work items: the ratio table first (64 columns: a wavefront shares its row bit, so
                // the term ranges are scalar reads), then the column and the row tables
                QK_MARK2(0);
                // The same terms as the diagonal stage before (the layers of a map at r >= 2), one chunk per sample: its
                // tables are still there.  If that stage carried the product state, constants by SimArgs::init_h, they
                // are divided out once (128 multiplications instead of 128 chains of a dozen).
                const bool reuse = st.reuse != 0 && a.nlive == 0 && !fused && (st.reuse == 1 || (a.init_h && fuse_init && hoist));
                if (reuse) {
                    if (st.reuse == 2) {
                        const double klo = 1.0 / (a.init_lo * a.init_chunk), khi = 1.0 / a.init_hi;
                        for (uint32_t j = tid; j < nlo + nhi; j += T) {
                            double2& e = j < nlo ? tlo[j] : thi[j - nlo];
                            const double k = j < nlo ? klo : khi;
                            e = make_double2(e.x * k, e.y * k);
                        }
                    }
                } else if (!(fused && hoist)) for (uint32_t j = tid; j < n_rat + nlo + nhi; j += T) {
                    double fr = 1.0, fi = 0.0;
                    if (j < n_rat) {
                        const uint32_t c = j & (nlo - 1u);
                        const uint32_t ra = __builtin_amdgcn_readfirstlane(j >> h);
                        const uint32_t w0 = whi | c, w1 = w0 | (1u << (h + ra));
                        double gr = 1.0, gi = 0.0;
                        for (int k = (c_stages + s * kStageWords + kStageXoWord)[ra], k1 = (c_stages + s * kStageWords + kStageXoWord)[ra + 1]; k < k1; ++k) {
                            apply_term(tx[k], w1, fr, fi);
                            apply_term(tx[k], w0, gr, gi);
                        }
                        // unit-modulus factors: f1 / f0 = f1 * conj(f0)
                        rat[j] = make_double2(fma(fr, gr, fi * gi), fma(fi, gr, -(fr * gi)));
                    } else if (j < n_rat + nlo) {
                        const uint32_t c = j - n_rat, w = whi | c;
                        apply_terms(tl, st.nLo, w, fr, fi);
                        apply_terms(tx, st.nX, w, fr, fi);   // X(0, c)
                        tlo[c] = fused ? cmul(tlo[c], make_double2(fr, fi)) : make_double2(fr, fi);
                    } else {
                        const uint32_t r = j - n_rat - nlo, w = whi | (r << h);
                        apply_terms(th, st.nHi, w, fr, fi);
                        thi[r] = fused ? cmul(thi[r], make_double2(fr, fi)) : make_double2(fr, fi);
                    }
                }
                QK_MARK2(1);
                lds_barrier();
                QK_MARK2(2);
                // The fused stage followed by the Hadamards on the four row qubits a thread of the diagonal sweep walks (the
                // planner puts that group first in the layer): the thread's 16 amplitudes -- rows r0 .. r0 + 15 of its
                // column -- are exactly one item of that radix-16 sweep, so they go through the butterflies in registers
                // before they are written for the first time.  One LDS pass and one barrier less per layer.
                bool with_h = false;
                if constexpr (MAXG >= 3) {
                    if (fused && !to_hbm && h == 6 && nrb - (a.lgT - 6) == 4 && s + 1 < a.n_stage &&
                        !(a.h_store && s + 1 == a.n_stage - 1)) {
                        const DevStage nx = load_stage(c_stages, s + 1);
                        with_h = nx.kind == ST_U1 && nx.ng == 4 && nx.gv[0] == h && nx.gv[1] == h + 1 && nx.gv[2] == h + 2 &&
                                 nx.gv[3] == h + 3;
                    }
                }
                if (with_h) {
                    if constexpr (MAXG >= 3) {
                        const uint32_t c = tid & 63u, r0 = (uint32_t)(tid >> 6) << 4;
                        const bool cross = st.nX != 0;
                        double2 X = tlo[c];
                        double2 rr0 = make_double2(1.0, 0.0), rr1 = rr0, rr2 = rr0, rr3 = rr0;
                        if (cross) {
                            for (int b = 4; b < nrb; ++b)
                                if ((r0 >> b) & 1u) X = cmul(X, rat[((uint32_t)b << h) | c]);
                            rr0 = rat[c]; rr1 = rat[(1u << h) | c]; rr2 = rat[(2u << h) | c]; rr3 = rat[(3u << h) | c];
                        }
                        double2 vv[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int g = i ^ (i >> 1);
                            if (i > 0 && cross) {
                                const int b = __builtin_ctz(i);
                                double2 q = b == 0 ? rr0 : (b == 1 ? rr1 : (b == 2 ? rr2 : rr3));
                                if (!((g >> b) & 1)) q.y = -q.y;
                                X = cmul(X, q);
                            }
                            vv[g] = cmul(X, thi[r0 | (uint32_t)g]);
                        }
#pragma unroll
                        for (int gb = 0; gb < 4; ++gb)
#pragma unroll
                            for (int k = 0; k < 16; ++k)
                                if (!((k >> gb) & 1)) {
                                    const int k1 = k | (1 << gb);
                                    const double2 a0 = vv[k], a1 = vv[k1];
                                    vv[k] = make_double2(a0.x + a1.x, a0.y + a1.y);
                                    vv[k1] = make_double2(a0.x - a1.x, a0.y - a1.y);
                                }
#pragma unroll
                        for (int k = 0; k < 16; ++k)
                            state[((r0 | (uint32_t)k) << h) | c] = make_double2(vv[k].x * 0.25, vv[k].y * 0.25);
                        ++s;      // the sweep is done
                    }
                } else if (st.nX == 0) {
                    if (fused && to_hbm) {
                        for (uint32_t l = tid; l < nloc; l += T) store_direct(l, cmul(tlo[l & (nlo - 1u)], thi[l >> h]));
                    } else if (fused) {
                        for (uint32_t l = tid; l < nloc; l += T) state[l] = cmul(tlo[l & (nlo - 1u)], thi[l >> h]);
                    } else {
                        for (uint32_t l = tid; l < nloc; l += T) {
                            const double2 f0 = tlo[l & (nlo - 1u)], f1 = thi[l >> h];
                            const double2 v = cmul(state[l], cmul(f0, f1));
                            if (to_hbm) store_direct(l, v);
                            else state[l] = v;
                        }
                    }
                } else {
                    // h == 6: lane = column; the wave owns 2^kb consecutive rows
                    const uint32_t c = tid & 63u, wv = tid >> 6;
                    const int kb = nrb - (a.lgT - 6);
                    const uint32_t rpw = 1u << kb, r0 = wv << kb;
                    double2 X = tlo[c];
                    for (int b = kb; b < nrb; ++b)
                        if ((r0 >> b) & 1u) X = cmul(X, rat[((uint32_t)b << h) | c]);
                    double2 rr0 = rat[c], rr1 = rr0, rr2 = rr0, rr3 = rr0;   // ratios of the walked bits
                    if (kb > 1) rr1 = rat[(1u << h) | c];
                    if (kb > 2) rr2 = rat[(2u << h) | c];
                    if (kb > 3) rr3 = rat[(3u << h) | c];
                    if (kb >= 2 && !fused) {
                        // Four rows at a time: X(r0 + i0 + j, c) = X(r0 + i0, c) * {1, rr0, rr1, rr0 rr1}[j].  The four
                        // amplitudes, their row factors and the products are independent of each other, so the LDS reads
                        // of a block are in flight together and the multiplications overlap -- with two waves per SIMD a
                        // walk that updates X once per row (Gray code: one multiplication per amplitude) is a chain of
                        // LDS and FP64 latencies nothing hides.  (The fused form below, which reads no amplitudes, keeps the walk:
                        // measured, 4.9 vs 6.0 us per 12-qubit item.)
                        const double2 rr01 = cmul(rr0, rr1);
                        const uint32_t dep_c = to_hbm ? dep_lo[c] : 0u;
                        for (uint32_t i0 = 0; i0 < rpw; i0 += 4) {
                            const uint32_t r = r0 | i0;
                            const double2 t0 = thi[r], t1 = thi[r + 1], t2 = thi[r + 2], t3 = thi[r + 3];
                            const double2 s0 = state[(r << h) | c], s1 = state[((r + 1) << h) | c],
                                          s2 = state[((r + 2) << h) | c], s3 = state[((r + 3) << h) | c];
                            double2 Xb = X;
                            if (i0 & 4u) Xb = cmul(Xb, rr2);
                            if (i0 & 8u) Xb = cmul(Xb, rr3);
                            double2 f0 = cmul(Xb, t0), f1 = cmul(cmul(Xb, rr0), t1), f2 = cmul(cmul(Xb, rr1), t2),
                                    f3 = cmul(cmul(Xb, rr01), t3);
                            f0 = cmul(s0, f0); f1 = cmul(s1, f1); f2 = cmul(s2, f2); f3 = cmul(s3, f3);
                            if (to_hbm) {
                                // deposit() is linear over disjoint bits: the column's part once per item, the row's part
                                // wave-uniform
#define QK_ST(j) { const int64_t at = chunk_phys | dep_c | dep_hi[r + j##u]; \
                   if (a.psi32) (a.psi32 + sample * a.D)[at] = make_float2((float)f##j.x, (float)f##j.y); \
                   else gpsi[at] = f##j; }
                                QK_REP4(QK_ST)
#undef QK_ST
                            } else {
                                state[(r << h) | c] = f0; state[((r + 1) << h) | c] = f1;
                                state[((r + 2) << h) | c] = f2; state[((r + 3) << h) | c] = f3;
                            }
                        }
                    } else {
#pragma unroll 2
                        for (uint32_t i = 0; i < rpw; ++i) {
                            const uint32_t g = i ^ (i >> 1);
                            if (i > 0) {
                                const int b = __builtin_ctz(i);              // the bit that flips between gray(i-1) and gray(i)
                                double2 q = b == 0 ? rr0 : (b == 1 ? rr1 : (b == 2 ? rr2 : rr3));
                                if (!((g >> b) & 1u)) q.y = -q.y;            // bit cleared: divide = multiply by the conjugate
                                X = cmul(X, q);
                            }
                            const uint32_t r = r0 | g;
                            const double2 f = cmul(X, thi[r]);
                            const uint32_t l = (r << h) | c;
                            const double2 v = fused ? f : cmul(state[l], f);
                            if (to_hbm) store_direct(l, v);
                            else state[l] = v;
                        }
                    }
                }
            } else if (st.kind == ST_U1) {
                if (st.ng == 1) sweep_u1_any<1>(state, mat, c_mats, st, nloc, tid, T);
                else if (st.ng == 2) sweep_u1_any<2>(state, mat, c_mats, st, nloc, tid, T);
                else if constexpr (MAXG >= 3) {
                    if (st.ng == 3) sweep_u1_any<3>(state, mat, c_mats, st, nloc, tid, T);
                    else {      // four Hadamards (the planner groups nothing else by four)
                        if (a.h_load && s == 0)
                            sweep_h16_io(st, nloc, tid, T, [&](uint32_t l) { return gpsi[chunk_phys | dep_local(l)]; },
                                         [&](uint32_t l, double2 v) { state[l] = v; });
                        else if (a.h_store && s == a.n_stage - 1)
                            sweep_h16_io(st, nloc, tid, T, [&](uint32_t l) { return state[l]; },
                                         [&](uint32_t l, double2 v) { store_direct(l, v); });
                        else
                            sweep_h16(state, st, nloc, tid, T);
                    }
                }
            } else if (st.kind == ST_CX) {
                const int vc = st.va, vt = st.vb;
                if (vc >= L) {  // control is a chunk bit: X on the target, or nothing
                    if ((whi >> vc) & 1u) {
                        const uint32_t lowmask = (1u << vt) - 1u;
                        uint32_t i = tid;
                        for (; i + 3u * T < (nloc >> 1); i += 4u * T) {
#define QK_LD(k) const uint32_t b##k = (((i + k##u * T) & ~lowmask) << 1) | ((i + k##u * T) & lowmask); \
                 const double2 x##k = state[b##k], y##k = state[b##k | (1u << vt)];
#define QK_ST(k) state[b##k] = y##k; state[b##k | (1u << vt)] = x##k;
                            QK_REP4(QK_LD)
                            QK_REP4(QK_ST)
#undef QK_LD
#undef QK_ST
                        }
                        for (; i < (nloc >> 1); i += T) {
                            const uint32_t i0 = ((i & ~lowmask) << 1) | (i & lowmask);
                            const uint32_t i1 = i0 | (1u << vt);
                            const double2 a0 = state[i0];
                            state[i0] = state[i1];
                            state[i1] = a0;
                        }
                    }
                } else {
                    const int lo = vc < vt ? vc : vt, hi = vc < vt ? vt : vc;
                    const uint32_t m_lo = (1u << lo) - 1u, m_hi = (1u << hi) - 1u;
                    uint32_t i = tid;
                    for (; i + 3u * T < (nloc >> 2); i += 4u * T) {     // four swaps in flight
#define QK_LD(k) uint32_t b##k = (((i + k##u * T) & ~m_lo) << 1) | ((i + k##u * T) & m_lo); \
                 b##k = (((b##k & ~m_hi) << 1) | (b##k & m_hi)) | (1u << vc); \
                 const double2 x##k = state[b##k], y##k = state[b##k | (1u << vt)];
#define QK_ST(k) state[b##k] = y##k; state[b##k | (1u << vt)] = x##k;
                        QK_REP4(QK_LD)
                        QK_REP4(QK_ST)
#undef QK_LD
#undef QK_ST
                    }
                    for (; i < (nloc >> 2); i += T) {
                        uint32_t b = ((i & ~m_lo) << 1) | (i & m_lo);
                        b = ((b & ~m_hi) << 1) | (b & m_hi);
                        const uint32_t i0 = b | (1u << vc);
                        const uint32_t i1 = i0 | (1u << vt);
                        const double2 a0 = state[i0];
                        state[i0] = state[i1];
                        state[i1] = a0;
                    }
                }
            } else {  // ST_ECR on (q0 = va, q1 = vb); matrix index = 2*bit(q1) + bit(q0)
                const int v0 = st.va, v1 = st.vb;
                const int lo = v0 < v1 ? v0 : v1, hi = v0 < v1 ? v1 : v0;
                const uint32_t m_lo = (1u << lo) - 1u, m_hi = (1u << hi) - 1u;
                const double r = 0.70710678118654752440;
                // 1/sqrt2 * [[0,1,0,i],[1,0,-i,0],[0,i,0,1],[-i,0,1,0]]
#define QK_LD(k) uint32_t j##k = (((i + k##u * T) & ~m_lo) << 1) | ((i + k##u * T) & m_lo); \
                 j##k = ((j##k & ~m_hi) << 1) | (j##k & m_hi); \
                 const double2 p##k = state[j##k], q##k = state[j##k | (1u << v0)], u##k = state[j##k | (1u << v1)], \
                               w##k = state[j##k | (1u << v0) | (1u << v1)];
#define QK_ST(k) state[j##k] = make_double2(r * (q##k.x - w##k.y), r * (q##k.y + w##k.x)); \
                 state[j##k | (1u << v0)] = make_double2(r * (p##k.x + u##k.y), r * (p##k.y - u##k.x)); \
                 state[j##k | (1u << v1)] = make_double2(r * (w##k.x - q##k.y), r * (w##k.y + q##k.x)); \
                 state[j##k | (1u << v0) | (1u << v1)] = make_double2(r * (u##k.x + p##k.y), r * (u##k.y - p##k.x));
                uint32_t i = tid;
                for (; i + T < (nloc >> 2); i += 2u * T) {     // two items in flight
                    QK_LD(0) QK_LD(1)
                    QK_ST(0) QK_ST(1)
                }
                for (; i < (nloc >> 2); i += T) {
                    QK_LD(0)
                    QK_ST(0)
                }
#undef QK_LD
#undef QK_ST
            }
            lds_barrier();
            QK_MARK(st.kind == ST_DIAG ? 3 : (st.kind == ST_U1 ? 4 : 5));
        }
        // ---- write back (amplitudes that are still known-zero are left to the next pass) ------
        // the next sample's angles: every stage of this item is behind a barrier, its terms and matrices are free
        if (have_next) scatter_angle(a.staged ? s_expr[tid_x] : a.exprs[tid_x], cs_next.x, cs_next.y, rt, mat);
        if (fetch)   // more angles than threads: the rest without the register detour
            for (int e = tid_x + T; e < a.n_expr; e += T) {
                const double2 v = a.cs[(int64_t)e * a.N + (next >> a.nlive)];
                scatter_angle(a.staged ? s_expr[e] : a.exprs[e], v.x, v.y, rt, mat);
            }
        lds_barrier();
        if constexpr (RDM) rdm1_from_lds(state, L, s_red, a.rdm + sample * (L * 4), tid, T);
        if (a.direct || a.last_direct || a.h_store) {
            // already stored by the last stage (a diagonal sweep or a radix-16 Hadamard sweep)
        } else if (a.psi32) {      // only on the last pass: intermediate passes stay FP64
            float2* g32 = a.psi32 + sample * a.D;
            for (uint32_t l = tid; l < nloc; l += T) {
                const double2 v = state[l];
                g32[chunk_phys | dep_local(l)] = make_float2((float)v.x, (float)v.y);
            }
        } else if (!RDM || a.psi) {
            uint32_t l = tid;
            for (; l + 7u * T < nloc; l += 8u * T) {
#define QK_LD(k) const double2 v##k = state[l + k##u * T];
#define QK_ST(k) if (!((l + k##u * T) & a.zero_out)) gpsi[chunk_phys | dep_local(l + k##u * T)] = v##k;
                QK_REP8(QK_LD)
                QK_REP8(QK_ST)
#undef QK_LD
#undef QK_ST
            }
            for (; l < nloc; l += T)
                if (!(l & a.zero_out)) gpsi[chunk_phys | dep_local(l)] = state[l];
        }
        // LDS is reused by the next item.  A bare barrier: every wave's LDS reads have returned
        // before it issued its stores, and the stores themselves need not have landed.
        __builtin_amdgcn_s_barrier();
        QK_MARK(6);
#ifdef QK_SIM_TRACE
        if (tid == 0) tr_acc[7] += 1.0;
#endif
    }
#ifdef QK_SIM_TRACE
    if (tid == 0 && a.trace) {
        for (int k = 0; k < 8; ++k) a.trace[(int64_t)blockIdx.x * 16 + k] = tr_acc[k];
        a.trace[(int64_t)blockIdx.x * 16 + 8] = (double)tr_angles;                    // part of slot 0: up to the end of thread 0's angle program
        a.trace[(int64_t)blockIdx.x * 16 + 9] = (double)(wall_clock64() - tr_begin);  // the workgroup's whole item loop
        a.trace[(int64_t)blockIdx.x * 16 + 10] = (double)tr_begin;
        a.trace[(int64_t)blockIdx.x * 16 + 11] = (double)tr_pre;
        for (int k = 0; k < 4; ++k) a.trace[(int64_t)blockIdx.x * 16 + 12 + k] = tr_acc2[k];                      // part of slot 0: up to the start of the angle programs
    }
#endif
}

int upload_circuit(qk_circuit& c, DevTables& out) {
    int dev = 0;
    QK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(c.mu);
    auto it = c.tables.find(dev);
    if (it != c.tables.end()) {
        out = it->second;
        return QK_OK;
    }
    DevTables t;
    auto up = [&](auto*& dptr, const auto& vec) -> hipError_t {
        using T = typename std::remove_reference<decltype(vec)>::type::value_type;
        const size_t bytes = std::max<size_t>(vec.size(), 1) * sizeof(T);
        hipError_t e = hipMalloc((void**)&dptr, bytes);
        if (e != hipSuccess) return e;
        if (!vec.empty()) e = hipMemcpy(dptr, vec.data(), vec.size() * sizeof(T), hipMemcpyHostToDevice);
        return e;
    };
    auto all = [&]() -> hipError_t {
        hipError_t e;
        if ((e = up(t.stages, c.stages)) != hipSuccess) return e;
        if ((e = up(t.terms, c.terms)) != hipSuccess) return e;
        if ((e = up(t.exprs, c.exprs)) != hipSuccess) return e;
        if ((e = up(t.mats, c.mats)) != hipSuccess) return e;
        if ((e = up(t.comps, c.comps)) != hipSuccess) return e;
        if ((e = up(t.facts, c.facts)) != hipSuccess) return e;
        if ((e = up(t.prog, c.prog)) != hipSuccess) return e;
        if ((e = up(t.consts, c.consts)) != hipSuccess) return e;
        if (!c.wave_blob.empty()) {
            if ((e = hipMalloc(&t.wave, c.wave_blob.size() * 8)) != hipSuccess) return e;
            if ((e = hipMemcpy(t.wave, c.wave_blob.data(), c.wave_blob.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return e;
        }
        if (!c.reg_blob.empty()) {
            if ((e = hipMalloc(&t.reg, c.reg_blob.size() * 8)) != hipSuccess) return e;
            e = hipMemcpy(t.reg, c.reg_blob.data(), c.reg_blob.size() * 8, hipMemcpyHostToDevice);
        }
        return e;
    };
    const hipError_t e = all();
    if (e != hipSuccess) {      // nothing half-built stays behind
        (void)hipFree(t.stages); (void)hipFree(t.terms); (void)hipFree(t.exprs); (void)hipFree(t.mats);
        (void)hipFree(t.comps); (void)hipFree(t.facts); (void)hipFree(t.prog); (void)hipFree(t.consts);
        if (t.wave) (void)hipFree(t.wave);
        if (t.reg) (void)hipFree(t.reg);
        return fail(QK_ERR_HIP, std::string("uploading the circuit tables: ") + hipGetErrorString(e));
    }
    c.tables[dev] = t;
    out = t;
    return QK_OK;
}

void free_circuit_device(qk_circuit& c) {
    std::lock_guard<std::mutex> lock(c.mu);
    for (auto& kv : c.tables) {
        DevTables& t = kv.second;
        (void)hipFree(t.stages); (void)hipFree(t.terms); (void)hipFree(t.exprs); (void)hipFree(t.mats);
        (void)hipFree(t.comps); (void)hipFree(t.facts); (void)hipFree(t.prog); (void)hipFree(t.consts);
        if (t.wave) (void)hipFree(t.wave);
        if (t.reg) (void)hipFree(t.reg);
    }
    c.tables.clear();
}

}  // namespace qk

#ifdef QK_SIM_TRACE
static std::atomic<double*> g_sim_trace{nullptr};
extern "C" void qk_sim_set_trace(double* table_dev) { g_sim_trace = table_dev; }
#endif

namespace qk {
// psi and / or rdm: rdm != null asks for the fused mode (one pass, one chunk per sample)
// Stream-ordered scratch (gate angles, phase tables, matrices of a batch) comes from a memory pool of the library's own,
// one per device, that keeps up to 2 GiB cached between calls: the device's default pool hands everything back at the next
// synchronisation, and a call of a fraction of a millisecond then pays a fresh allocation of tens of MiB each time
// (0.25 ms of a 0.34 ms simulation of 32768 8-qubit samples, tools/sim_bench.py --gib 0.125).
static int scratch_alloc(void** out, size_t bytes, hipStream_t stream) {
    static std::mutex mu;
    static hipMemPool_t pools[64] = {};
    int dev = 0;
    QK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(QK_ERR_UNSUPPORTED, "device index out of range");
    hipMemPool_t pool = nullptr;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!pools[dev]) {
            hipMemPoolProps props;
            std::memset(&props, 0, sizeof(props));
            props.allocType = hipMemAllocationTypePinned;
            props.handleTypes = hipMemHandleTypeNone;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = dev;
            QK_HIP(hipMemPoolCreate(&pools[dev], &props));
            uint64_t keep = (uint64_t)2 << 30;
            QK_HIP(hipMemPoolSetAttribute(pools[dev], hipMemPoolAttrReleaseThreshold, &keep));
        }
        pool = pools[dev];
    }
    QK_HIP(hipMallocFromPoolAsync(out, bytes, pool, stream));
    return QK_OK;
}

// diagonal stages and layers of Hadamards only behind the first layer (the maps at r >= 2)
static bool shape_h_layers(const qk_circuit& c) {
    if (c.passes.size() != 1) return false;
    const Pass& ps = c.passes[0];
    bool layers = false;
    for (int k = ps.n_init; k < ps.n_stage; ++k) {
        const DevStage& st = c.stages[ps.stage_off + k];
        if (st.kind == ST_DIAG) continue;
        if (st.kind != ST_U1) return false;
        for (int g = 0; g < st.ng; ++g)
            if (c.mats[ps.mat_off + st.gm[g]].gate != QK_H) return false;
        layers = true;
    }
    return layers;
}

// The register-resident kernel in its reduced-density-matrix mode (the projected kernel; unpacked plans with 16 amplitudes
// per thread: 10 qubits and up) instead of the LDS kernel's fused mode, from 2^22 amplitudes.  Measured, LDS-fused ->
// register-fused (tools/rdm_route_ab.py): 12 qubits x 32768: ZZ r = 1 0.82 -> 0.66 ms, r = 2 1.17 -> 1.02, custom1 r = 2
// 2.19 -> 1.32, a genome 1.62 -> 0.93; 11 x 65536: ZZ r = 1 1.27 -> 0.67, r = 2 1.52 -> 1.03; 10 x 131072: ZZ r = 1 1.47 ->
// 0.76, custom1 r = 2 3.31 -> 1.69 -- but ZZ r = 2 1.63 -> 1.89: Hadamard-layer circuits at 10 qubits keep the LDS kernel.
static bool reg_rdm_route(const qk_circuit& c, int64_t N) {
    static const int env = [] { const char* e = std::getenv("QK_RDM_REG"); return e ? std::atoi(e) : -1; }();      // (A/B runs: 0 never, 1 always)
    static const bool force_reg = [] { const char* e = std::getenv("QK_SIM_PATH"); return e && std::strcmp(e, "reg") == 0; }();
    if (!reg_rdm_mode_ok(c) || c.passes.size() != 1) return false;
    if (c.passes[0].L != c.n_qubits) return true;      // (the LDS kernel cannot run this circuit fused: 13 qubits with a chunk bit)
    if (env == 0) return false;
    if (env == 1 || force_reg) return true;
    const int n = c.n_qubits;
    return (N << n) >= ((int64_t)1 << 22) && !(n == 10 && shape_h_layers(c));
}

// psi32 != null: the last pass writes complex64 there; with several passes psi_dev (complex128) is
// the workspace the earlier passes go through.
static int simulate_impl(const qk_circuit* cc, const double* X_dev, int64_t N, int64_t ldx, void* psi_dev, double* rdm_dev,
                         void* stream, void* psi32_dev = nullptr) {
    // rdm_dev: the plan without 512-thread passes, where the circuit has one
    qk_circuit& c = (rdm_dev && cc->rdm_plan) ? *cc->rdm_plan : *const_cast<qk_circuit*>(cc);
    if (c.n_params > 0 && (!X_dev || ldx < c.n_params))
        return fail(QK_ERR_INVALID, "qk_simulate_dev: X is null or its row stride is smaller than the parameter count");
    DevTables dt;
    int rc = upload_circuit(c, dt);
    if (rc) return rc;
    const int n = c.n_qubits;
    // up to 10 qubits: the wave-resident kernel (qk_wave.hip) -- the state never leaves the registers of one
    // wavefront.  QK_SIM_PATH=lds keeps such circuits on the LDS kernel below (A/B runs, and the fused
    // simulate + RDM mode, which reduces from the LDS image).
    static const bool force_lds = [] { const char* e = std::getenv("QK_SIM_PATH"); return e && std::strcmp(e, "lds") == 0; }();
    // A circuit that is the product state and ONE diagonal stage (the Z / ZZ / custom maps at r = 1) at 9 or 10 qubits
    // is faster on the LDS kernel's direct-store pass now that its angles come from the pre-pass (4 GiB of Psi, 10
    // qubits: ZZ 1.41 vs 1.89 ms, Z 1.03 vs 1.25; 9 qubits: ZZ 1.72 vs 2.54; equal at 8, the wave kernel ahead below
    // and on everything with butterflies).  QK_SIM_PATH=wave keeps such circuits on the wave kernel.
    static const bool force_wave = [] { const char* e = std::getenv("QK_SIM_PATH"); return e && std::strcmp(e, "wave") == 0; }();
    const bool direct_shape = c.passes.size() == 1 && c.passes[0].n_init > 0 && c.passes[0].n_stage == c.passes[0].n_init + 1 &&
                              c.stages[c.passes[0].stage_off + c.passes[0].n_init].kind == ST_DIAG;
    // ... and from 6 qubits where, in addition, every first gate is a Hadamard (the Z and ZZ maps: the product state is a
    // constant there): 8 qubits ZZ 1.16 vs 1.50 ms per 2 GiB, Z 0.78 vs 1.03; 7 qubits 1.87 vs 2.36; the custom maps
    // are level or behind at 8 qubits and below.
    auto all_h_init = [&](const Pass& ps) {
        bool all_h = ps.first && ps.n_init == n;
        for (int k = 0; all_h && k < ps.n_init; ++k) all_h = c.mats[ps.mat_off + c.stages[ps.stage_off + k].gm[0]].gate == QK_H;
        return all_h;
    };
    // ... and the maps at r >= 2 as well (diagonal stages and layers of Hadamards only), from 6 qubits: the layers'
    // repeated diagonal stage builds its phase tables once (DevStage::reuse), at 10 qubits the Hadamards go 16
    // amplitudes per lane.  ZZ r = 2, 4 GiB: 10 qubits 3.03 vs 4.3 ms, 9: 4.70 vs 5.91, 8: 5.33 vs 6.20, 6: 14.9 vs 16.2.
    auto h_layers_shape = [&]() {
        if (c.passes.size() != 1) return false;
        const Pass& ps = c.passes[0];
        bool layers = false;
        for (int k = ps.n_init; k < ps.n_stage; ++k) {
            const DevStage& st = c.stages[ps.stage_off + k];
            if (st.kind == ST_DIAG) continue;
            if (st.kind != ST_U1) return false;
            for (int g = 0; g < st.ng; ++g)
                if (c.mats[ps.mat_off + st.gm[g]].gate != QK_H) return false;
            layers = true;
        }
        return layers;
    };
    // (small batches stay where they were: the pre-pass is one more launch, 5-10 us of a 0.1-1 ms call)
    const bool big = (N >> 22 >> 0) != 0 || (N << n) >= ((int64_t)1 << 22);
    const bool prefer_lds = !force_wave && (big || n > kWaveMaxQubits) && ((direct_shape && (n >= 9 || (n >= 6 && all_h_init(c.passes[0])))) ||
                                            (n >= 6 && h_layers_shape()));
    // The register-resident kernel (qk_reg.hip), for circuits that have such a plan (4 to 13 qubits, one pass):
    //   11 to 13 qubits: whenever there are butterflies or swaps to do -- a circuit that is the product state and ONE
    //     diagonal stage stays on the direct-store pass below (nothing to keep in registers: it writes every amplitude once);
    //   up to 10 qubits: every circuit, once the batch fills the chip with groups of samples (tools/sim_bench.py, round 3:
    //     8 qubits ZZ r=1 2.30 -> 1.48 ms, r=2 5.23 -> 2.48 ms per 4 GiB); small batches stay on the wave-resident
    //     kernel, which is one launch instead of three or four.
    // QK_SIM_PATH=reg sends every circuit that has a plan there, QK_SIM_PATH=noreg none (A/B runs).
    static const int reg_mode = [] {
        const char* e = std::getenv("QK_SIM_PATH");
        return !e ? 0 : (std::strcmp(e, "reg") == 0 ? 1 : (std::strcmp(e, "noreg") == 0 || std::strcmp(e, "lds") == 0 || std::strcmp(e, "wave") == 0 ? -1 : 0));
    }();
    // (where the batch fills the chip for it: 2^20 amplitudes at 6 qubits and below, 2^22 at 7 to 9, 2^23 at 10 -- second
    // half of round 3, both kernels' latest builds: 2^22 amplitudes, ZZ r = 2: 10 qubits 0.072 ms wave-resident / 0.075
    // register-resident, 8 qubits 0.101 / 0.069, 6 qubits 0.260 / 0.095; 2^20: 8 qubits 0.043 / 0.046, 6 qubits 0.073 / 0.048)
    const int reg_from = n <= 6 ? 20 : (n >= 10 ? 23 : 22);
    // (10 qubits, the shapes the LDS kernel's one-wave workgroups keep: the product state of an all-Hadamard layer and one
    // diagonal stage -- ZZ r = 1 1.03 against 1.11 ms per 4 GiB -- and Hadamard layers between diagonal stages of
    // single-qubit terms only -- Z r = 2 0.84 against 1.03)
    auto single_qubit_terms_only = [&]() {
        for (const DevTerm& t : c.terms)
            if (t.type == TM_CZ || t.type == TM_ZZP) return false;
        return true;
    };
    // (circuits that are a product state altogether -- the maps without entanglement, whose gates the planner multiplies
    // into one matrix per qubit -- go either way by 15 %: Z r = 2 0.83 on the LDS kernel against 1.03, custom2 r = 2 1.21
    // against 1.07; they stay on the register path)
    const bool lds_at_10 = n == 10 && c.passes.size() == 1 &&
                           ((direct_shape && all_h_init(c.passes[0])) || (h_layers_shape() && single_qubit_terms_only()));
    const bool reg_pays = n >= 11 ? !direct_shape : (!lds_at_10 && ((N >> reg_from) != 0 || (N << n) >= ((int64_t)1 << reg_from)));
    const bool reg_rdm = rdm_dev && dt.reg && reg_rdm_route(c, N);      // (its reduced-density-matrix mode: see reg_rdm_route)
    // (a reduced-density-matrix call on a circuit whose LDS plan keeps a chunk bit has no other fused route: QK_SIM_PATH does
    // not apply to it)
    const bool reg_rdm_only = reg_rdm && c.passes.size() == 1 && c.passes[0].L != n;
    if (rdm_dev && !reg_rdm && !(c.passes.size() == 1 && c.passes[0].L == n && n <= RDM_MAXQ))
        return fail(QK_ERR_UNSUPPORTED, "qk_simulate_rdm1_dev: no fused route for this circuit (a workspace is needed)");
    if (dt.reg && (reg_mode >= 0 || reg_rdm_only) && (rdm_dev ? reg_rdm : (reg_mode > 0 || reg_pays))) {
        const RegHeader* H = (const RegHeader*)c.reg_blob.data();
        // The angles and the stage tables of the batch live in stream order.  Large batches go in chunks of whole sample
        // groups: the tables stay within 1 GiB whatever N is, and the pre-pass kernels of chunk k + 1 (a seventh of the time
        // of a call: tools/sim_bench.py under rocprofv3) run on a side stream while the simulator works on chunk k -- the
        // simulator leaves a third of a SIMD's registers and most of its wave slots free, the pre-pass kernels use no LDS.
        const int64_t per_sample = (int64_t)H->n_expr * 16 + (int64_t)(H->n_tab - H->n_itab) * 256 + (int64_t)H->n_mat * 64;
        const int64_t group = (int64_t)1 << (H->L - H->nq);
        int64_t chunk = per_sample > 0 ? std::max<int64_t>(((int64_t)1 << 30) / per_sample, 65536) : N;
        static const int overlap_env = [] { const char* e = std::getenv("QK_REG_OVERLAP"); return e ? std::atoi(e) : -1; }();      // (A/B runs: 0 off, k >= 2 chunks)
        // Four chunks, from 2^25 amplitudes (a call of half a millisecond), where a sample's tables are three eighths of its
        // statevector or more -- that is, narrow registers or many parametrised gates (measured at 4 GiB of Psi, one chunk ->
        // four: 4 qubits ZZ r = 2 12.6 -> 11.7 ms, genome 17.0 -> 13.0; 5 qubits custom1 19.5 -> 17.2; 6 qubits genome
        // 9.7 -> 8.0; 7 qubits custom1 5.9 -> 4.9; 8 qubits custom1 3.9 -> 3.4).  Below that share the four launches' tails
        // and the kernels sharing the CUs cost more than the pre-pass was (8 qubits ZZ r = 2, 0.19 of Psi: 1.66 -> 1.79; 12
        // qubits genome, 0.04: 1.73 -> 2.0 and more).
        int parts = 1;
        if (overlap_env >= 2) parts = overlap_env;
        else if (overlap_env != 0 && per_sample * 8 >= 3 * ((int64_t)16 << n) && (N << n) >= ((int64_t)1 << 25)) parts = 4;
        if (parts > 1) {      // (not under stream capture: the side stream is made on first use, and a captured call stays on one stream)
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) { (void)hipGetLastError(); parts = 1; }
            else if (cap != hipStreamCaptureStatusNone) parts = 1;
        }
        if (parts > 1) chunk = std::min(chunk, (N + parts - 1) / parts);
        chunk = std::max<int64_t>(group, std::min<int64_t>(N, chunk / group * group));
        const bool overlap = parts > 1 && chunk < N;
        static const bool dbg_reg = std::getenv("QK_REG_DEBUG") != nullptr;
        if (dbg_reg) std::fprintf(stderr, "reg call: n %d N %lld table bytes per sample %lld (%.2f of Psi) chunk %lld parts %d\n", n, (long long)N,
                                  (long long)per_sample, (double)per_sample / (double)((int64_t)16 << n), (long long)chunk, overlap ? parts : 1);
        hipStream_t side = nullptr;
        // Events: a ring per calling thread and device, created on first use and never destroyed (an event handed to
        // hipStreamWaitEvent and destroyed right away is one of the two things the first form of this code did before it lost
        // a wait now and then; a thread's calls come one after the other, a call waits for an event at most two chunks after
        // recording it, so no record can overtake a wait that is still to be enqueued; the few events of a thread that ends
        // stay with the process).
        constexpr int kEvents = 16;
        static std::mutex side_mu;
        static hipStream_t side_streams_[64] = {};
        thread_local hipEvent_t ev_ring[64][kEvents] = {};
        thread_local unsigned ev_next[64] = {};
        int dev = 0;
        auto next_event = [&](hipEvent_t* out) -> int {
            hipEvent_t& e = ev_ring[dev][ev_next[dev]++ % kEvents];
            if (!e) QK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            *out = e;
            return QK_OK;
        };
        if (overlap) {
            QK_HIP(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64) return fail(QK_ERR_UNSUPPORTED, "device index out of range");
            {
                std::lock_guard<std::mutex> lock(side_mu);
                if (!side_streams_[dev]) QK_HIP(hipStreamCreateWithFlags(&side_streams_[dev], hipStreamNonBlocking));
                side = side_streams_[dev];
            }
        }
        const hipStream_t main_s = (hipStream_t)stream;
        const hipStream_t pre = overlap ? side : main_s;
        // The buffers of a call -- one set, or two that alternate when the pre-pass runs ahead -- are allocated and freed in the
        // caller's stream order only: the side stream launches kernels and nothing else (memory that one stream frees and the
        // other allocates while the free is still pending came back before its last reader had finished).
        const int nbuf = overlap ? 2 : 1;
        double2* cs_buf[2] = {nullptr, nullptr};
        void* tab_buf[2] = {nullptr, nullptr};
        hipEvent_t sim_done[2] = {nullptr, nullptr};
        const size_t tb_max = reg_side_bytes(c, chunk);
        if (H->n_expr > 0 && (chunk + 63) / 64 > 0x7fffffff) return fail(QK_ERR_UNSUPPORTED, "qk_simulate_dev: N too large");
        auto release = [&](int code) {
            if (code != QK_OK && overlap) (void)hipStreamSynchronize(side);      // (kernels of the side stream may still be writing)
            for (int b = 0; b < nbuf; ++b) {
                if (cs_buf[b]) (void)hipFreeAsync(cs_buf[b], main_s);
                if (tab_buf[b]) (void)hipFreeAsync(tab_buf[b], main_s);
            }
            return code;
        };
        for (int b = 0; b < nbuf; ++b) {
            if (H->n_expr > 0 && (rc = scratch_alloc((void**)&cs_buf[b], (size_t)H->n_expr * (size_t)chunk * sizeof(double2), main_s))) return release(rc);
            if (tb_max > 0 && (rc = scratch_alloc(&tab_buf[b], tb_max, main_s))) return release(rc);
        }
        if (overlap) {
            // the side stream starts where the caller's stream is now (X is ready there, the buffers are allocated)
            hipEvent_t ev_in = nullptr;
            if ((rc = next_event(&ev_in))) return release(rc);
            if (hipEventRecord(ev_in, main_s) != hipSuccess || hipStreamWaitEvent(side, ev_in, 0) != hipSuccess)
                return release(fail(QK_ERR_HIP, "qk_simulate_dev: event between the caller's stream and the pre-pass stream"));
        }
        int64_t k = 0;
        for (int64_t s0 = 0; s0 < N; s0 += chunk, ++k) {
            const int64_t Nc = std::min(chunk, N - s0);
            const int b = (int)(k % nbuf);
            double2* cs = cs_buf[b];
            void* tabs = tab_buf[b];
            // (this set's previous reader, the simulator of chunk k - 2, has to be done before the pre-pass writes it again)
            if (overlap && sim_done[b] && hipStreamWaitEvent(side, sim_done[b], 0) != hipSuccess)
                return release(fail(QK_ERR_HIP, "qk_simulate_dev: hipStreamWaitEvent"));
            if (H->n_expr > 0)
                hipLaunchKernelGGL(angle_table_kernel, dim3((unsigned)((Nc + 63) / 64), (unsigned)((H->n_expr + 3) / 4)), dim3(256), 0,
                                   pre, (const DevExpr*)((const char*)dt.reg + H->off_expr), H->n_expr, dt.prog, dt.consts,
                                   X_dev + s0 * ldx, ldx, Nc, cs);
            rc = QK_OK;
            if (tb_max > 0) rc = launch_reg_side(c, dt.reg, cs, Nc, tabs, pre);
            if (rc == QK_OK && overlap) {      // the simulator's launch waits for this chunk's pre-pass
                hipEvent_t ev = nullptr;
                if ((rc = next_event(&ev))) return release(rc);
                if (hipEventRecord(ev, side) != hipSuccess || hipStreamWaitEvent(main_s, ev, 0) != hipSuccess)
                    return release(fail(QK_ERR_HIP, "qk_simulate_dev: event between the pre-pass stream and the caller's stream"));
            }
            if (rc == QK_OK && rdm_dev)
                rc = launch_reg_sim(c, dt.reg, cs, tabs, Nc, nullptr, nullptr, main_s, rdm_dev + s0 * (int64_t)(4 * n));
            else if (rc == QK_OK)
                rc = launch_reg_sim(c, dt.reg, cs, tabs, Nc, psi32_dev ? nullptr : (void*)((double2*)psi_dev + (s0 << n)),
                                    psi32_dev ? (void*)((float2*)psi32_dev + (s0 << n)) : nullptr, main_s);
            if (rc != QK_OK) return release(rc);
            if (overlap && s0 + 2 * chunk < N) {      // (only when a later chunk will write this set again)
                if ((rc = next_event(&sim_done[b]))) return release(rc);
                if (hipEventRecord(sim_done[b], main_s) != hipSuccess) return release(fail(QK_ERR_HIP, "qk_simulate_dev: hipEventRecord"));
            }
        }
        return release(QK_OK);
    }
    if (dt.wave && !rdm_dev && !force_lds && !prefer_lds) {
        WaveJob job{};
        job.prog = dt.wave;
        job.X = X_dev;
        job.ldx = ldx;
        job.N = N;
        job.psi = psi32_dev ? psi32_dev : psi_dev;
        job.f32 = psi32_dev ? 1 : 0;
        job.unit_begin = 0;
        job.prog_bytes = (int32_t)(c.wave_blob.size() * 8);
        const int64_t units = (N + c.wave_spw - 1) / c.wave_spw;
        return launch_wave_jobs(&job, 1, nullptr, wave_nreg_class(n), units, wave_lds_bytes(c), (hipStream_t)stream);
    }
    static DeviceOnce attr_set;
    if (attr_set.pending()) {
        QK_HIP(hipFuncSetAttribute((const void*)sim_pass_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        QK_HIP(hipFuncSetAttribute((const void*)sim_pass_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        QK_HIP(hipFuncSetAttribute((const void*)sim_pass_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        QK_HIP(hipFuncSetAttribute((const void*)sim_pass_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.done();
    }
    // the gate angles of all samples, every pass's at once (angle_table_kernel); the table lives in stream order
    double2* cs_dev = nullptr;
    const int n_expr_all = (int)c.exprs.size();
    if (n_expr_all > 0) {
        if ((N + 63) / 64 > 0x7fffffff) return fail(QK_ERR_UNSUPPORTED, "qk_simulate_dev: N too large");
        // the device's default pool hands freed memory back to the driver at the next synchronisation unless told to
        // keep it: without this every call pays a driver allocation (0.1 ms) for its table
        static DeviceOnce pool_set;
        if (pool_set.pending()) {
            int dev = 0;
            hipMemPool_t pool = nullptr;
            QK_HIP(hipGetDevice(&dev));
            QK_HIP(hipDeviceGetDefaultMemPool(&pool, dev));
            uint64_t keep = ~0ull;
            QK_HIP(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
            pool_set.done();
        }
        if ((rc = scratch_alloc((void**)&cs_dev, (size_t)n_expr_all * (size_t)N * sizeof(double2), (hipStream_t)stream))) return rc;
        hipLaunchKernelGGL(angle_table_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((n_expr_all + 3) / 4)), dim3(256), 0,
                           (hipStream_t)stream, dt.exprs, n_expr_all, dt.prog, dt.consts, X_dev, ldx, N, cs_dev);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            (void)hipFreeAsync(cs_dev, (hipStream_t)stream);
            return fail(QK_ERR_HIP, std::string("angle_table_kernel: ") + hipGetErrorString(e));
        }
    }
    struct FreeCs {      // released in stream order once the passes below have been enqueued (or on an early return)
        double2* p; hipStream_t s;
        ~FreeCs() { if (p) (void)hipFreeAsync(p, s); }
    } free_cs{cs_dev, (hipStream_t)stream};
    for (size_t pi = 0; pi < c.passes.size(); ++pi) {
        const Pass& ps = c.passes[pi];
        SimArgs a{};
        a.psi32 = (psi32_dev && pi + 1 == c.passes.size()) ? (float2*)psi32_dev : nullptr;
        a.cs = cs_dev ? cs_dev + (size_t)ps.expr_off * (size_t)N : nullptr;
        a.N = N;
        a.psi = (double2*)psi_dev;
        a.rdm = rdm_dev;
        a.D = (int64_t)1 << n;
        a.stages = dt.stages + ps.stage_off;
        a.terms = dt.terms + ps.term_off;
        a.exprs = dt.exprs + ps.expr_off;
        a.mats = dt.mats + ps.mat_off;
#ifdef QK_SIM_TRACE
        a.trace = g_sim_trace.load();
#endif
        a.comps = dt.comps + ps.comp_off;
        a.facts = dt.facts + ps.fact_off;
        a.n_comp = ps.n_comp;
        {
            a.init_h = all_h_init(ps) ? 1 : 0;
            const double r = 0.70710678118654752440;
            auto rpow = [&](int k) { return std::ldexp((k & 1) ? r : 1.0, -(k / 2)); };     // 2^(-k/2)
            a.init_lo = rpow(ps.h);
            a.init_hi = rpow(ps.L - ps.h);
            a.init_chunk = rpow(n - ps.L);
        }
        a.n_fact = (int)((pi + 1 < c.passes.size() ? c.passes[pi + 1].fact_off : (int)c.facts.size()) - ps.fact_off);
        a.n_stage = ps.n_stage;
        a.n_expr = ps.n_expr;
        a.n_mat = ps.n_mat;
        a.n_term = ps.n_term;
        a.L = ps.L;
        a.h = ps.h;
        a.lgT = ps.lgT;
        a.n_init = ps.n_init;
        a.n_qubits = n;
        a.first = ps.first ? 1 : 0;
        a.zero_in = ps.zero_in;
        a.zero_out = ps.zero_out;
        a.nruns = ps.nruns;
        for (int r = 0; r < ps.nruns; ++r) { a.run_mask[r] = ps.run_mask[r]; a.run_shift[r] = ps.run_shift[r]; }
        const int nchunk_bits = n - ps.L;
        a.live_chunks = ((1u << nchunk_bits) - 1u) & ~(ps.zero_out >> ps.L);
        a.nlive = __builtin_popcount(a.live_chunks);
        if ((N >> (62 - a.nlive)) != 0) return fail(QK_ERR_UNSUPPORTED, "qk_simulate_dev: N too large");
        a.n_items = N << a.nlive;
        // first diagonal stage directly after the product state: its tables are built once per sample
        const bool fuse = ps.first && ps.n_init > 0 && ps.n_init < ps.n_stage &&
                          c.stages[ps.stage_off + ps.n_init].kind == ST_DIAG;
        const bool hoist = fuse && a.nlive > 0;     // several chunks per sample: separate base tables
        a.hoist = fuse ? (hoist ? 2 : 1) : 0;
        // nothing but the product state and one diagonal stage (Z / ZZ maps at r = 1): no LDS image of the state
        const bool direct = fuse && ps.n_stage == ps.n_init + 1 && !rdm_dev && ps.zero_out == 0 && pi + 1 == c.passes.size();
        a.direct = direct ? 1 : 0;
        a.last_direct = (!direct && !rdm_dev && ps.zero_out == 0 && pi + 1 == c.passes.size() && ps.n_stage > 0 &&
                         c.stages[ps.stage_off + ps.n_stage - 1].kind == ST_DIAG) ? 1 : 0;
        {
            // a radix-16 Hadamard sweep at either end of the pass goes to HBM itself (no LDS image in between) when its
            // targets lie above bit 3 and every amplitude of the chunk is live
            auto h16 = [&](int k) {
                const DevStage& st = c.stages[ps.stage_off + k];
                return st.kind == ST_U1 && st.ng == 4 && st.gv[0] >= 4;
            };
            const bool reg3 = ps.maxg >= 3 && !ps.narrow;
            const int first_stage = ps.first ? ps.n_init : 0;
            a.h_store = (reg3 && !rdm_dev && ps.zero_out == 0 && ps.n_stage > first_stage && h16(ps.n_stage - 1)) ? 1 : 0;
            a.h_load = (reg3 && !ps.first && ps.zero_in == 0 && ps.n_stage > 0 && h16(0) && !(a.h_store && ps.n_stage == 1)) ? 1 : 0;
        }
        a.dep_bytes = (int)((4 * ((1u << ps.h) + (1u << (ps.L - ps.h))) + 15u) & ~15u);
        size_t lds = (direct ? (size_t)0 : ((size_t)16 << ps.L)) + (size_t)32 * ps.n_term + (size_t)64 * ps.n_mat + (size_t)a.dep_bytes +
                     (size_t)(hoist ? 32 : 16) * ((1u << ps.h) + (1u << (ps.L - ps.h))) +
                     (size_t)16 * ((size_t)(ps.L - ps.h) << ps.h) + (size_t)8 * ps.n_init;
        const size_t stage_bytes = sizeof(DevExpr) * ps.n_expr + (size_t)16 * ps.n_comp + (size_t)8 * ((a.n_fact + 1) / 2);
        // stage the angle records and the composite tables in LDS unless that costs a resident workgroup
        // (counted with everything else the workgroup holds: the fused mode's reduction scratch comes on top)
        const size_t budget = (size_t)160 * 1024;
        const size_t rdm_bytes = rdm_dev ? (size_t)8 * RDM_SLOTS * ((1 << a.lgT) / 64) : 0;
        const size_t need0 = lds + rdm_bytes, need1 = need0 + stage_bytes;
        a.staged = (need1 <= budget && budget / need1 >= std::min<size_t>(budget / need0, 2)) ? 1 : 0;
        if (a.staged) lds += stage_bytes;
        if (rdm_dev) {
            a.red_off = (int)((lds - (((size_t)16 << ps.L) + (size_t)32 * ps.n_term + (size_t)64 * ps.n_mat + (size_t)a.dep_bytes +
                                      (size_t)(hoist ? 32 : 16) * ((1u << ps.h) + (1u << (ps.L - ps.h))) +
                                      (size_t)16 * ((size_t)(ps.L - ps.h) << ps.h))) / 8);
            lds += rdm_bytes;
        }
        if (lds > (size_t)160 * 1024) return fail(QK_ERR_UNSUPPORTED, "qk_simulate_dev: pass tables exceed LDS");
        const int threads = 1 << a.lgT;
        // resident workgroups per CU: LDS, and registers (MAXG = 3: 2 waves/SIMD; MAXG = 2: 3 waves/SIMD)
        // a direct pass has no butterflies: the 256-thread instantiation (3 waves/SIMD) serves it whatever L is
        const int maxg = ((direct && threads <= 256) || ps.narrow) ? 2 : ps.maxg;
        const int by_regs = ((maxg >= 3 || rdm_dev) ? 512 : 1024) / threads;
        int per_cu = (int)std::min<size_t>((160 * 1024) / lds, (size_t)by_regs);
        per_cu = std::max(1, std::min(per_cu, 16));
        // one-wave workgroups with sweeps to do: past 12 (8 qubits) or 8 (9 qubits) of them a CU gets slower, not faster
        // (ZZ r = 2, 2 GiB: 8 qubits 3.33 ms at 12 against 3.98 at 16; 9 qubits 2.84 at 8 against 3.59 at 12); a pass that
        // only stores keeps all it can get
        if (!direct && threads == 64) per_cu = std::min(per_cu, ps.L >= 9 ? 8 : 12);
        const int64_t nblocks = std::min<int64_t>(a.n_items, (int64_t)256 * per_cu);
        const dim3 grid((unsigned)nblocks), block(threads);
        if (rdm_dev && maxg >= 3) hipLaunchKernelGGL((sim_pass_kernel<3, true>), grid, block, lds, (hipStream_t)stream, a);
        else if (rdm_dev) hipLaunchKernelGGL((sim_pass_kernel<2, true>), grid, block, lds, (hipStream_t)stream, a);
        else if (maxg >= 3) hipLaunchKernelGGL((sim_pass_kernel<3, false>), grid, block, lds, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((sim_pass_kernel<2, false>), grid, block, lds, (hipStream_t)stream, a);
        QK_HIP(hipGetLastError());
    }
    return QK_OK;
}
}  // namespace qk

extern "C" int qk_simulate_dev(const qk_circuit* cc, const double* X_dev, int64_t N, int64_t ldx,
                               void* psi_dev, void* stream) {
    using namespace qk;
    if (!cc || !psi_dev) return fail(QK_ERR_INVALID, "qk_simulate_dev: null circuit or output");
    if (N < 0) return fail(QK_ERR_INVALID, "qk_simulate_dev: negative N");
    if (N == 0) return QK_OK;
    return simulate_impl(cc, X_dev, N, ldx, psi_dev, nullptr, stream);
}

extern "C" int qk_simulate_c64_dev(const qk_circuit* cc, const double* X_dev, int64_t N, int64_t ldx, void* psi32_dev,
                                   void* psi_ws_dev, void* stream) {
    using namespace qk;
    if (!cc || !psi32_dev) return fail(QK_ERR_INVALID, "qk_simulate_c64_dev: null circuit or output");
    if (N < 0) return fail(QK_ERR_INVALID, "qk_simulate_c64_dev: negative N");
    if (N == 0) return QK_OK;
    if (cc->passes.size() > 1 && !psi_ws_dev)
        return fail(QK_ERR_INVALID, "qk_simulate_c64_dev: a multi-pass circuit needs the complex128 workspace");
    return simulate_impl(cc, X_dev, N, ldx, psi_ws_dev, nullptr, stream, psi32_dev);
}

// The fused mode reduces from the LDS kernel's image; circuits that the register-resident kernel runs much faster than
// the LDS kernel are quicker through qk_simulate_dev + qk_rdm1_dev even though Psi is then written and read once
// (N x 2^n complex128 of workspace).  Measured, fused -> two steps (tools/rdm_route_ab.py): GA genomes 12 qubits x 32768
// 2.20 -> 1.68 ms, 11 x 65536 3.34 -> 2.04, 10 x 131072 2.61 -> 1.84; custom1 r = 2 at 11 qubits 3.12 -> 2.14; ZZ r = 2 with
// full entanglement at 12 qubits 2.47 -> 1.96; the ZZ map at r = 1 stays fused at every width (0.87 against 1.25 at 12
// qubits), at r = 2 from 10 qubits (1.29 against 1.56 at 12; 8 qubits: 2.17 -> 1.81).
static bool rdm_two_steps_pay(const qk_circuit* cc, int64_t N) {
    using namespace qk;
    const qk_circuit& c = *cc;
    const int n = c.n_qubits;
    if (c.reg_blob.empty() || c.passes.size() != 1) return false;
    if (qk_simulate_rdm1_is_fused(cc) && reg_rdm_route(cc->rdm_plan ? *cc->rdm_plan : c, N)) return false;      // fused in the register kernel: faster still
    const int reg_from = n <= 6 ? 20 : (n >= 10 ? 23 : 22);      // (the batch sizes from which simulate_impl takes the register path)
    if (n < 11 && (N << n) < ((int64_t)1 << reg_from)) return false;
    if (n >= 11 && (N << n) < ((int64_t)1 << 22)) return false;
    if (((N << n) >> 29) != 0) return false;      // more than 8 GiB of workspace: stay fused
    const Pass& ps = c.passes[0];
    const bool direct_shape = ps.n_init > 0 && ps.n_stage == ps.n_init + 1 && c.stages[ps.stage_off + ps.n_init].kind == ST_DIAG;
    if (direct_shape || ps.n_stage == ps.n_init) return false;      // product state (+ one diagonal stage): nothing to gain
    bool h_layers = true;      // diagonal stages and layers of Hadamards only (the maps at r >= 2)
    for (int k = ps.n_init; k < ps.n_stage && h_layers; ++k) {
        const DevStage& st = c.stages[ps.stage_off + k];
        if (st.kind == ST_DIAG) continue;
        h_layers = st.kind == ST_U1;
        for (int g = 0; h_layers && g < st.ng; ++g) h_layers = c.mats[ps.mat_off + st.gm[g]].gate == QK_H;
    }
    if (!h_layers) return true;
    const RegHeader* H = (const RegHeader*)c.reg_blob.data();
    const RegOp* ops = (const RegOp*)((const char*)c.reg_blob.data() + H->off_op);
    bool ratio = false;
    for (int k = 0; k < H->n_op; ++k) ratio = ratio || (ops[k].kind == RG_DIAG && (ops[k].flags & 512));
    return n < 10 || (ratio && n <= 12);      // (13 qubits, ZZ r = 2 with full entanglement: 1.78 fused against 2.73)
}

extern "C" int qk_simulate_rdm1_wants_workspace(const qk_circuit* c, int64_t N) {
    if (!c) return 0;
    return (!qk_simulate_rdm1_is_fused(c) || rdm_two_steps_pay(c, N)) ? 1 : 0;
}

extern "C" int qk_simulate_rdm1_dev(const qk_circuit* cc, const double* X_dev, int64_t N, int64_t ldx, double* rdm_dev,
                                    void* psi_ws_dev, void* stream) {
    using namespace qk;
    if (!cc || !rdm_dev) return fail(QK_ERR_INVALID, "qk_simulate_rdm1_dev: null circuit or output");
    if (N < 0) return fail(QK_ERR_INVALID, "qk_simulate_rdm1_dev: negative N");
    if (N == 0) return QK_OK;
    if (qk_simulate_rdm1_is_fused(cc) && !(psi_ws_dev && rdm_two_steps_pay(cc, N))) return simulate_impl(cc, X_dev, N, ldx, nullptr, rdm_dev, stream);
    if (!psi_ws_dev)
        return fail(QK_ERR_INVALID, "qk_simulate_rdm1_dev: this circuit needs a statevector workspace (not fused)");
    const int rc = simulate_impl(cc, X_dev, N, ldx, psi_ws_dev, nullptr, stream);
    if (rc) return rc;
    return qk_rdm1_dev(psi_ws_dev, N, cc->n_qubits, rdm_dev, stream);
}

extern "C" int qk_simulate_rdm1_is_fused(const qk_circuit* c) {
    if (!c) return 0;
    // the whole statevector of a sample is one LDS-resident chunk of a single pass ...
    if (c->passes.size() == 1 && c->passes[0].L == c->n_qubits && c->n_qubits <= qk::RDM_MAXQ) return 1;
    // ... or the register-resident kernel holds it (13 qubits where the LDS plan keeps a chunk bit: GA genomes)
    const qk_circuit& r = c->rdm_plan ? *c->rdm_plan : *c;
    return r.passes.size() == 1 && qk::reg_rdm_mode_ok(r) ? 1 : 0;
}
