// Register-resident statevector simulator for gfx950: circuits of 4 to 13 qubits (format and idea: qk_reg.h).
//
// One workgroup of 2^(L-4) threads per sample, 16 amplitudes per thread, held in registers across ALL stages of the
// circuit.  In a layout four qubits are "register" qubits -- a thread's 16 amplitudes differ in exactly those bits --
// and every gate is applied where its target is a register qubit: a butterfly between registers of the lane, with
// the matrix in scalar registers.  Diagonal stages are applied in whatever layout the state is in: the phase of
// amplitude (k, t) is  A[k] * Blo[t & 15] * Bhi[t >> 4] * (the few terms that straddle those three bit groups),
// three 16-entry tables per stage instead of the LDS kernel's 64 + 64 + 6 x 64 entries with chains of a dozen
// factors, and the straddling terms cost a thread one ratio per register bit that it walks in Gray-code order.
// Between two layouts the workgroup transposes through LDS: every thread writes its 16 amplitudes to its OWN slots
// of the layout it is in, one barrier, and reads the 16 of the next layout -- slots only it will write again, so one
// barrier per transpose is enough.  The LDS image is XOR-swizzled (slot = i ^ (i >> 4 & 15) ^ (i >> 8 & 15) ^ (i >> 12)):
// linear over the index bits, so a thread's slot is (its own part) ^ (a scalar per register), and the 8 (writes) or 16
// (reads) lanes that the LDS serves together hit distinct 16-byte slots whichever aligned group of four is in registers.
//
// Circuits of 4 to 9 qubits (PACK): 2^(L - nq) samples share a workgroup, side by side in a virtual register of L = 10 .. 12
// index bits whose top bits number the sample.  Those bits are thread bits in every layout and no gate touches them:
// segments, ops, transposes and stores are the same instruction stream as for one wide sample, and the per-sample
// cost of that stream (record loads, dispatch, barriers, loop control -- most of what a narrow circuit pays in the
// wave-resident and LDS kernels) is shared by all samples of the group.  What differs per sample -- angles, matrices,
// phase tables -- lives in LDS once per sample and is read at a per-lane offset; matrices stay in vector registers.
//
// Nothing that depends on the sample is computed here: reg_table_kernel and reg_matrix_kernel (below; one thread per
// sample and table / matrix, uniform control flow per wave) and angle_table_kernel (qk_sim.hip) prepare the phase
// tables, the matrices and the (cos, sin) of the angles for the whole batch; the simulator fetches a group's block
// with LDS-DMA into one of two buffers while the previous group computes, behind one barrier and a counted s_waitcnt.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

#include "qk_common.h"
#include "qk_reg.h"
#include "qk_rdm.h"

namespace qk {

typedef double v2d __attribute__((ext_vector_type(2)));   // native vector: stays in registers

struct RegArgs {
    const char* blob;      // RegHeader + tables (device)
    const double2* cs;     // (cos, sin) of every angle program, [expr][sample]
    const double2* side;   // per group of samples: the phase tables of its diagonal stages [sample][table][16], then its matrices
                           // [sample][matrix][4] (reg_table_kernel, reg_matrix_kernel; the batch padded to whole groups)
    int64_t N;
    double2* psi;
    float2* psi32;         // FP32 output: the state is evolved in FP64 and rounded once, at the store
    int64_t D;
    double* trace;         // QK_REG_TRACE builds: 8 doubles per workgroup (100 MHz ticks per phase, summed over its samples)
    double* rdm;           // not null: instead of Psi, the per-qubit reduced density matrices [sample][qubit][4] (one sample per workgroup)
    int red_off;           // rdm: byte offset in LDS of the reduction scratch (RDM_SLOTS doubles per wave)
};

namespace {

#define QK_CONST_AS __attribute__((address_space(4)))
typedef const int32_t QK_CONST_AS* CWords;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ CWords cwords(const void* p) { return (CWords)(uintptr_t)p; }

__device__ __forceinline__ v2d cmul(v2d a, v2d b) { return (v2d){fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)}; }
__device__ __forceinline__ v2d cmulc(v2d a, v2d b) {      // a * conj(b)
    return (v2d){fma(a.x, b.x, a.y * b.y), fma(a.y, b.x, -(a.x * b.y))};
}
__device__ __forceinline__ double2 cmul2(double2 a, double2 b) {
    return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ double2 cfma2(double2 a, double2 b, double2 c) {
    return make_double2(fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y)));
}
__device__ __forceinline__ double uniform_f64(double x) {
    const uint64_t u = (uint64_t)__double_as_longlong(x);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS accesses of the sample loop go through 32-bit LDS addresses turned into pointers on the spot.  The angles arrive
// by LDS-DMA, and behind a DMA instruction the compiler puts s_waitcnt vmcnt(0) in front of every LDS access whose
// pointer it can trace to the workgroup's LDS block -- the very wait (memory operations complete in order: it would
// also sit out the previous sample's stores) that the counted wait at the top of the sample loop replaces.
template <class T> __device__ __forceinline__ T ldsr(uint32_t addr) { return *(const __attribute__((address_space(3))) T*)(size_t)addr; }
template <class T> __device__ __forceinline__ void ldsw(uint32_t addr, T v) { *(__attribute__((address_space(3))) T*)(size_t)addr = v; }
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int2v __attribute__((ext_vector_type(2)));
// A term a thread evaluates itself (RegTerm, 16 B): the record comes from the plan by scalar loads, one term ahead, so
// what kind of term it is and which bits it looks at are wave-uniform scalars; per lane: the bits and the (cos, sin) of
// the lane's sample (cs_l: the lane's base in the angle buffer, sb = log2 of the samples per group).
struct TermS { int type, va, vb, rpos, expr; };
__device__ __forceinline__ TermS term_rec(CWords c_term, int t) {
    const int x = c_term[4 * t], va = c_term[4 * t + 1], vb = c_term[4 * t + 2], e = c_term[4 * t + 3];
    return TermS{x & 255, va, vb, x >> 8, e};
}
__device__ __forceinline__ v2d term_cs(const TermS& t, uint32_t cs_l, int sb) {
    const v2d cs = ldsr<v2d>(cs_l + ((uint32_t)(t.expr < 0 ? 0 : t.expr) << (4 + sb)));
    return t.expr < 0 ? (v2d){-1.0, 0.0} : cs;
}
__device__ __forceinline__ v2d term_factor(const TermS& t, v2d cs, uint32_t w) {
    const uint32_t ba = (w >> t.va) & 1u, bb = (w >> t.vb) & 1u;
    const bool rz = t.type == TM_RZ;
    const bool on = rz || (t.type == TM_P ? ba != 0u : (t.type == TM_CZ ? (ba & bb) != 0u : (ba ^ bb) != 0u));
    const double sn = (rz && !ba) ? -cs.y : cs.y;
    return (v2d){on ? cs.x : 1.0, on ? sn : 0.0};
}

__device__ __forceinline__ uint32_t swz(uint32_t i) { return i ^ ((i >> 4) & 15u) ^ ((i >> 8) & 15u) ^ ((i >> 12) & 15u); }

template <int V> struct IC { static constexpr int value = V; };

// The thread's amplitudes as 2 NA separate doubles, not NA two-element vectors: a vector-typed value that is live across
// the op dispatch becomes a 128-bit register tuple whose halves the allocator will not update in place, and every join
// behind a butterfly then copies all of them (877 v_mov_b64 of 5618 instructions with v2d v[NA]).
template <int NA> struct Amps {
    double re[NA], im[NA];
    struct Ref {
        double& x;
        double& y;
        __device__ __forceinline__ operator v2d() const { return (v2d){x, y}; }
        __device__ __forceinline__ Ref& operator=(v2d a) { x = a.x; y = a.y; return *this; }
        __device__ __forceinline__ Ref& operator=(const Ref& r) { const double rx = r.x, ry = r.y; x = rx; y = ry; return *this; }
    };
    __device__ __forceinline__ Ref operator[](int k) { return Ref{re[k], im[k]}; }
};
template <class F>
__device__ __forceinline__ void with_pos(int p, F&& f) {      // p is wave-uniform: ONE scalar dispatch, straight-line code behind it
    if (p == 0) f(IC<0>());
    else if (p == 1) f(IC<1>());
    else if (p == 2) f(IC<2>());
    else f(IC<3>());
}

// In-place arithmetic on the amplitude registers.  Every result is written onto one of its own inputs (a tied asm
// operand), so a value that is live across the op dispatch keeps its register: written as plain expressions, the
// butterflies of an arm go to a second set of 64 registers and every join copies it back (32 v_mov_b64 per
// register position and op -- a third of a single-qubit gate's instructions).  `dep` operands are not read: they
// order an instruction behind the last reader of the value it overwrites.  VM: multiplier in a vector register
// (packed samples: every lane has its own matrix), else in scalar registers.
template <bool VM> __device__ __forceinline__ void ip_mul(double& p, double m, double dep) {       // p = m p
    if constexpr (VM) asm("v_mul_f64 %0, %1, %0" : "+v"(p) : "v"(m), "v"(dep));
    else asm("v_mul_f64 %0, %1, %0" : "+v"(p) : "s"(m), "v"(dep));
}
template <bool VM, bool NEG> __device__ __forceinline__ void ip_fma_acc(double& p, double m, double q) {      // p = (+-m) q + p
    if constexpr (VM) {
        if constexpr (NEG) asm("v_fma_f64 %0, -%1, %2, %0" : "+v"(p) : "v"(m), "v"(q));
        else asm("v_fma_f64 %0, %1, %2, %0" : "+v"(p) : "v"(m), "v"(q));
    } else {
        if constexpr (NEG) asm("v_fma_f64 %0, -%1, %2, %0" : "+v"(p) : "s"(m), "v"(q));
        else asm("v_fma_f64 %0, %1, %2, %0" : "+v"(p) : "s"(m), "v"(q));
    }
}
template <bool VM> __device__ __forceinline__ void ip_fma_self(double& q, double m, double t, double d0, double d1, double d2) {      // q = m q + t
    if constexpr (VM) asm("v_fma_f64 %0, %1, %0, %2" : "+v"(q) : "v"(m), "v"(t), "v"(d0), "v"(d1), "v"(d2));
    else asm("v_fma_f64 %0, %1, %0, %2" : "+v"(q) : "s"(m), "v"(t), "v"(d0), "v"(d1), "v"(d2));
}
// v <- v * ph (complex), in place
__device__ __forceinline__ void ip_cmul(double& x, double& y, v2d ph) {
    const double t = x * ph.y;
    ip_mul<true>(x, ph.x, t);
    ip_fma_acc<true, true>(x, ph.y, y);
    ip_fma_self<true>(y, ph.x, t, x, x, x);
}

// The butterflies of one register position go stage by stage over all eight pairs, so that an instruction and the one
// that needs its result are sixteen instructions apart (a 64-bit operation occupies the SIMD for four cycles and its
// result takes longer than that: with one or two waves on a SIMD a dependent pair back to back stalls).
// pair j of position P: registers lo(j) and lo(j) | (1 << P)
template <int P> __device__ __forceinline__ constexpr int pair_lo(int j) { return ((j >> P) << (P + 1)) | (j & ((1 << P) - 1)); }

// (a0, a1) <- (a0 + a1, a0 - a1): the difference as (a0 + a1) - 2 a1, onto a1
template <int P, int NA> __device__ __forceinline__ void hadamard_pos(Amps<NA>& v) {
    if constexpr ((1 << P) < NA) {
#pragma unroll
        for (int j = 0; j < NA / 2; ++j) {
            const int k = pair_lo<P>(j), h = k | (1 << P);
            asm("v_add_f64 %0, %0, %1" : "+v"(v.re[k]) : "v"(v.re[h]));
            asm("v_add_f64 %0, %0, %1" : "+v"(v.im[k]) : "v"(v.im[h]));
        }
#pragma unroll
        for (int j = 0; j < NA / 2; ++j) {
            const int k = pair_lo<P>(j), h = k | (1 << P);
            asm("v_fma_f64 %0, -2.0, %0, %1" : "+v"(v.re[h]) : "v"(v.re[k]));
            asm("v_fma_f64 %0, -2.0, %0, %1" : "+v"(v.im[h]) : "v"(v.im[k]));
        }
    }
}

// NA real systems [p; q] <- [A, +-B; +-C, D] [p; q]: system 2 j is (re[lo], CROSS ? im[hi] : re[hi]) with signs (NB0, NC0),
// system 2 j + 1 is (im[lo], CROSS ? re[hi] : im[hi]) with (NB1, NC1); four instructions and one temporary each
template <int P, int NA, bool VM, bool CROSS, bool NB0, bool NC0, bool NB1, bool NC1>
__device__ __forceinline__ void sys2_pos(Amps<NA>& v, double A, double B, double C, double D) {
    double t[NA];
#pragma unroll
    for (int j = 0; j < NA / 2; ++j) {
        const int k = pair_lo<P>(j);
        t[2 * j] = NC0 ? -(C * v.re[k]) : C * v.re[k];
        t[2 * j + 1] = NC1 ? -(C * v.im[k]) : C * v.im[k];
    }
#pragma unroll
    for (int j = 0; j < NA / 2; ++j) {
        const int k = pair_lo<P>(j);
        ip_mul<VM>(v.re[k], A, t[2 * j]);
        ip_mul<VM>(v.im[k], A, t[2 * j + 1]);
    }
#pragma unroll
    for (int j = 0; j < NA / 2; ++j) {
        const int k = pair_lo<P>(j), h = k | (1 << P);
        ip_fma_acc<VM, NB0>(v.re[k], B, CROSS ? v.im[h] : v.re[h]);
        ip_fma_acc<VM, NB1>(v.im[k], B, CROSS ? v.re[h] : v.im[h]);
    }
#pragma unroll
    for (int j = 0; j < NA / 2; ++j) {
        const int k = pair_lo<P>(j), h = k | (1 << P);
        ip_fma_self<VM>(CROSS ? v.im[h] : v.re[h], D, t[2 * j], v.re[k], v.re[k], v.re[k]);
        ip_fma_self<VM>(CROSS ? v.re[h] : v.im[h], D, t[2 * j + 1], v.im[k], v.im[k], v.im[k]);
    }
}

// a general complex matrix: a0 <- m0 a0 + m1 a1, a1 <- m2 a0 + m3 a1; sixteen instructions and four temporaries a pair,
// four pairs at a time
template <int P, int NA, bool VM>
__device__ __forceinline__ void gen2_pos(Amps<NA>& v, double m0x, double m0y, double m1x, double m1y, double m2x, double m2y, double m3x, double m3y) {
    constexpr int CH = NA / 2 < 4 ? NA / 2 : 4;
#pragma unroll
    for (int j0 = 0; j0 < NA / 2; j0 += CH) {
        double tx[CH], ty[CH], ux[CH], uy[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {      // what each amplitude adds to the other one, from the old values
            const int k = pair_lo<P>(j0 + j), h = k | (1 << P);
            tx[j] = fma(m2x, v.re[k], -(m2y * v.im[k])); ty[j] = fma(m2x, v.im[k], m2y * v.re[k]);      // m2 a0 -> a1
            ux[j] = fma(m1x, v.re[h], -(m1y * v.im[h])); uy[j] = fma(m1x, v.im[h], m1y * v.re[h]);      // m1 a1 -> a0
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int k = pair_lo<P>(j0 + j), h = k | (1 << P);
            ip_fma_acc<VM, false>(uy[j], m0y, v.re[k]);
            ip_fma_acc<VM, false>(ty[j], m3y, v.re[h]);
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {      // (behind every reader of the old real parts)
            const int k = pair_lo<P>(j0 + j), h = k | (1 << P);
            ip_fma_self<VM>(v.re[k], m0x, ux[j], tx[j], ty[j], uy[j]);
            ip_fma_self<VM>(v.re[h], m3x, tx[j], ux[j], uy[j], ty[j]);
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int k = pair_lo<P>(j0 + j), h = k | (1 << P);
            ip_fma_acc<VM, true>(v.re[k], m0y, v.im[k]);
            ip_fma_acc<VM, true>(v.re[h], m3y, v.im[h]);
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {      // (behind every reader of the old imaginary parts)
            const int k = pair_lo<P>(j0 + j), h = k | (1 << P);
            ip_fma_self<VM>(v.im[k], m0x, uy[j], v.re[k], v.re[h], v.re[k]);
            ip_fma_self<VM>(v.im[h], m3x, ty[j], v.re[h], v.re[k], v.re[h]);
        }
    }
}

// a 2x2 matrix on register position P: eight butterflies between registers k and k | (1 << P).  form: 0 real matrix
// (x, ry), 1 real diagonal + imaginary off-diagonal (rx), 2 general (sx, composites), 3 hadamard (adds only; its
// 1 / sqrt 2 goes to the running scale).  e0 .. e3: the matrix as it came out of LDS (not read for a hadamard).
template <int P, int NA, bool PACK>
__device__ __forceinline__ void u1_reg(Amps<NA>& v, v2d e0, v2d e1, v2d e2, v2d e3, int form, double& scale) {
    if constexpr ((1 << P) >= NA) return;
    if (form == 3) {
        hadamard_pos<P, NA>(v);
        scale *= 0.70710678118654752440;
        return;
    }
    // one sample per workgroup: the matrix is wave-uniform and goes to scalar registers; PACK: every lane has its sample's
    auto un = [](double x) __attribute__((always_inline)) { if constexpr (PACK) return x; else return uniform_f64(x); };
    const double m0x = un(e0.x), m0y = un(e0.y), m1x = un(e1.x), m1y = un(e1.y);
    const double m2x = un(e2.x), m2y = un(e2.y), m3x = un(e3.x), m3y = un(e3.y);
    if (form == 0) {
        sys2_pos<P, NA, PACK, false, false, false, false, false>(v, m0x, m1x, m2x, m3x);
    } else if (form == 1) {
        // a0.x' = m0x a0.x - m1y a1.y, a1.y' = m2y a0.x + m3x a1.y;  a0.y' = m0x a0.y + m1y a1.x, a1.x' = -m2y a0.y + m3x a1.x
        sys2_pos<P, NA, PACK, true, true, false, false, true>(v, m0x, m1y, m2y, m3x);
    } else {
        gen2_pos<P, NA, PACK>(v, m0x, m0y, m1x, m1y, m2x, m2y, m3x, m3y);
    }
}

template <int C, int T, int NA>
__device__ __forceinline__ void cx_reg(Amps<NA>& v) {
    if constexpr (C != T && (1 << C) < NA && (1 << T) < NA) {
#pragma unroll
        for (int k = 0; k < NA; ++k)
            if (((k >> C) & 1) && !((k >> T) & 1)) {
                const v2d a = v[k];
                v[k] = v[k | (1 << T)];
                v[k | (1 << T)] = a;
            }
    }
}

template <int T, int NA>
__device__ __forceinline__ void cx_thread(Amps<NA>& v, bool on) {
    if constexpr ((1 << T) >= NA) return;
#pragma unroll
    for (int k = 0; k < NA; ++k)
        if (!((k >> T) & 1)) {
            const v2d a0 = v[k], a1 = v[k | (1 << T)];
            v[k] = on ? a1 : a0;
            v[k | (1 << T)] = on ? a0 : a1;
        }
}

// ecr(q0, q1) on register positions P0, P1 (matrix index 2 bit(q1) + bit(q0)); its 1 / sqrt 2 goes to the scale
template <int P0, int P1, int NA>
__device__ __forceinline__ void ecr_reg(Amps<NA>& v) {
    if constexpr (P0 != P1 && (1 << P0) < NA && (1 << P1) < NA) {
#pragma unroll
        for (int k = 0; k < NA; ++k)
            if (!((k >> P0) & 1) && !((k >> P1) & 1)) {
                const v2d p = v[k], q = v[k | (1 << P0)], u = v[k | (1 << P1)], w = v[k | (1 << P0) | (1 << P1)];
                v[k] = (v2d){q.x - w.y, q.y + w.x};
                v[k | (1 << P0)] = (v2d){p.x + u.y, p.y - u.x};
                v[k | (1 << P1)] = (v2d){w.x - q.y, w.y + q.x};
                v[k | (1 << P0) | (1 << P1)] = (v2d){u.x + p.y, u.y - p.x};
            }
    }
}

}  // namespace

// TBN = log2(threads), RB = register qubits per layout (NA = 2^RB amplitudes per thread), L = TBN + RB.
// RB = 4: two waves per SIMD (256 VGPRs); RB = 3: four (128 VGPRs) -- LDS holds the same number of samples either way.
// PACK: 2^(L - nq) samples per workgroup (see the top of the file); nq = L otherwise.
// Waves per SIMD, lower and upper bound alike: LDS, not registers, decides how many workgroups a CU holds (two states of 12
// qubits, seven of 10), so the scheduler is told not to trade instruction-level parallelism for an occupancy the kernel
// cannot have.
// RDM: the reduced-density-matrix mode (a.rdm; its reduction is kept out of the plain kernels' register allocation).
template <int TBN, int RB, bool PACK, bool RDM = false>
__global__ void __launch_bounds__(1 << TBN) __attribute__((amdgpu_waves_per_eu(RB == 3 ? 4 : 2, RB == 3 ? 4 : 2))) reg_sim_kernel(RegArgs a) {
    extern __shared__ double2 lds[];
    constexpr int L = TBN + RB, T = 1 << TBN, NA = 1 << RB;
    const CWords hw = cwords(a.blob);
    const int nq = PACK ? hw[21] : L;                  // qubits of the circuit
    const int sbits = L - nq, S = 1 << sbits;           // samples per workgroup
    static_assert(offsetof(RegHeader, nq) == 84, "RegHeader is read word by word");
    const int n_seg = hw[1], n_mat = hw[6], init_h = hw[9];
    const int n_itab = hw[24], n_dtab = hw[3] - hw[24], n_expr = hw[27];      // n_expr: the angle programs the simulator reads itself
    const int off_seg = hw[10], off_op = hw[11], off_tab = hw[12], off_term = hw[13], off_init = hw[18];
    const double init_c = ((const double*)(a.blob + 88))[0];
    static_assert(sizeof(RegHeader) == 112 && offsetof(RegHeader, init_c) == 88 && offsetof(RegHeader, n_itab) == 96 && offsetof(RegHeader, n_expr_k) == 108 &&
                      sizeof(RegSeg) == 64 && sizeof(RegTerm) == 16, "RegHeader, RegSeg and RegTerm are read word by word");
    // LDS map (32-bit LDS addresses: see ldsr / ldsw): the transpose image; the terms evaluated per thread; per sample the
    // tables of the product state; one or two buffers of a group's stage tables and matrices (reg_table_kernel,
    // reg_matrix_kernel: fetched by LDS-DMA); the staged records; two buffers of angles (those of the per-thread terms)
    const int n_ib = (init_h & 1) ? 0 : n_itab;      // product-state tables are built here, unless every first gate is a Hadamard
    const bool side_single = (init_h & 2) != 0;      // one buffer of tables and matrices: fetched at the top of the group's own turn
    const uint32_t itab_s = 256u * (uint32_t)n_ib, dtab_s = 256u * (uint32_t)n_dtab, mat_s = 64u * (uint32_t)n_mat;      // per sample
    const int t_piece = (S * n_dtab) << 4, side_piece = t_piece + ((S * n_mat) << 2);      // 16-byte pieces of a group's tables (+ matrices)
    const uint32_t side_bytes = (uint32_t)((side_piece + 63) & ~63) << 4;               // one buffer: whole fetch instructions of 64 pieces
    const uint32_t state_a = (uint32_t)(size_t)(lptr_t)lds;
    const uint32_t itab_a = state_a + (16u << L);                   // 256 B per table
    const uint32_t side_a = itab_a + itab_s * (uint32_t)S;
    const uint32_t stab_a = side_a + (side_single ? 1u : 2u) * side_bytes;      // RegTab records of the product-state tables, 32 B
    const uint32_t init_a = stab_a + 32u * (uint32_t)n_itab;
    const int n_slot = (n_expr * S + 63) & ~63;                     // one slot of 64 pairs per fetch instruction; pair e * S + s
    const uint32_t cs_a = (init_a + 4u * (uint32_t)nq + 15u) & ~15u;   // two buffers of n_slot (cos, sin) pairs
    const CWords c_seg = cwords(a.blob + off_seg), c_op = cwords(a.blob + off_op), c_term = cwords(a.blob + off_term);
    const int tid0 = threadIdx.x;
    if (state_a & ((16u << L) - 1u)) __builtin_trap();      // the swizzled image must start where its address bits are clear (dynamic LDS starts at 0)

    // ---- once per workgroup: what does not depend on the sample -----------------------------------------------
    for (int t = tid0; t < 8 * n_itab; t += T) ldsw<int>(stab_a + 4u * t, ((const int32_t*)(a.blob + off_tab))[t]);
    for (int q = tid0; q < nq; q += T) ldsw<int>(init_a + 4u * q, ((const int32_t*)(a.blob + off_init))[q]);

    // a workgroup walks groups of S consecutive samples (the last group of the batch may be short: its missing samples
    // repeat sample N - 1 -- same angles, same amplitudes, stored to the same place)
    const int64_t n_group = (a.N + S - 1) >> sbits;
    const int64_t per_wg = (n_group + gridDim.x - 1) / gridDim.x;
    const int64_t s_begin = (int64_t)blockIdx.x * per_wg;
    const int64_t s_end = s_begin + per_wg < n_group ? s_begin + per_wg : n_group;
    const int wave_s = uni(tid0 >> 6);
    // What a group needs arrives by LDS-DMA, 64 pieces of 16 bytes per instruction, wave w issuing the instructions
    // w, w + T / 64, ...: the angles of its per-thread terms (lanes past the last one re-read it) and one contiguous
    // block of stage tables and matrices (the pre-pass kernels pad the batch to whole groups).
    auto share = [&](int slots) { const int k = (slots - wave_s + T / 64 - 1) / (T / 64); return k > 0 ? k : 0; };
    const int my_na = share(n_slot / 64), my_ns = share((int)(side_bytes >> 10));
    auto fetch_angles = [&](int64_t smp, int buf, int lane) __attribute__((always_inline)) {      // smp: the group
        for (int j = 0; j < my_na; ++j) {
            const int slot = wave_s + j * (T / 64);
            const int i = min(slot * 64 + lane, n_expr * S - 1);
            const int e = i >> sbits;
            const int64_t one = (smp << sbits) + (i & (S - 1));
            const double2* src = a.cs + ((int64_t)e * a.N + (one < a.N ? one : a.N - 1));
            const uint32_t dst = cs_a + 16u * (uint32_t)(buf * n_slot + slot * 64);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(size_t)dst, 16, 0, 0);
        }
    };
    auto fetch_side = [&](int64_t smp, int buf, int lane) __attribute__((always_inline)) {
        for (int j = 0; j < my_ns; ++j) {
            const int slot = wave_s + j * (T / 64);
            const double2* src = a.side + (smp * side_piece + min(slot * 64 + lane, side_piece - 1));
            const uint32_t dst = side_a + (uint32_t)buf * side_bytes + 1024u * (uint32_t)slot;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(size_t)dst, 16, 0, 0);
        }
    };
    if (s_begin < s_end) { fetch_angles(s_begin, 0, tid0 & 63); fetch_side(s_begin, 0, tid0 & 63); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

#ifdef QK_REG_TRACE
    double tr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tr_last = wall_clock64();
#define QK_RMARK(k) do { if (tid0 == 0) { const uint64_t t_ = wall_clock64(); tr[k] += (double)(t_ - tr_last); tr_last = t_; } } while (0)
#else
#define QK_RMARK(k) do { } while (0)
#endif
    int it = 0;
    for (int64_t sample = s_begin; sample < s_end; ++sample, it ^= 1) {
        // the thread index, rebuilt per sample from an opaque copy (as in sim_pass_kernel: what the compiler can prove
        // loop-invariant it computes before the loop, keeps across it and spills)
        uint32_t ones = ~0u;
        asm volatile("" : "+s"(ones));
        int tid = (wave_s << 6) | (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
        asm volatile("" : "+v"(tid));
        // This group's angles, tables and matrices were fetched one turn ago; younger than those fetches are only that
        // turn's NA stores of Psi, and memory operations complete in order -- a counted wait, so the stores go on
        // draining under this group's arithmetic (a plain vmcnt(0), which is what a prefetch into registers gets from the
        // compiler, holds the workgroup until the whole statevector of the previous sample has left the chip).
        if constexpr (RDM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (no stores of Psi behind the fetch in this mode: wait for everything)
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");
        QK_RMARK(0);
        lds_barrier();      // every wave's share of the fetch has landed; every wave is done with the previous group
        QK_RMARK(8);
        // The next group's data travel while this one is computed -- issued only now: the buffers they land in were read
        // by the previous group up to its last segment.
        const int64_t nxt = sample + 1 < s_end ? sample + 1 : sample;
        fetch_angles(nxt, it ^ 1, tid & 63);
        if (!side_single) {
            fetch_side(nxt, it ^ 1, tid & 63);
        } else if (sample > s_begin) {      // one buffer: this group's own tables and matrices, now (the first group's came before the loop)
            fetch_side(sample, 0, tid & 63);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
        }
        const uint32_t side_i = side_a + (side_single ? 0u : (uint32_t)it * side_bytes);      // this group's tables, then its matrices
        const uint32_t mats_i = side_i + ((uint32_t)t_piece << 4);
        QK_RMARK(9);
        // ---- the tables of the product state, 16 entries each: products of the first columns of the first gates' matrices
        // (not needed when every first gate is a Hadamard; the tables of the diagonal stages arrive ready-made)
        // (PACK: a wave builds the tables of the samples wave, wave + T / 64, ... of the group, its lanes over the entries)
        if (n_ib) {
            for (int so = PACK ? wave_s : 0; so < S; so += PACK ? T / 64 : 1) {
                const uint32_t mat_i = mats_i + mat_s * (uint32_t)so, tab_i = itab_a + itab_s * (uint32_t)so;
                for (int e = PACK ? (tid & 63) : tid; e < 16 * n_ib; e += PACK ? 64 : T) {
                    const uint32_t tb = stab_a + 32u * (uint32_t)(e >> 4);      // bits[4], nbits, fixed, t0 (mask of extra qubits), nt < 0
                    const int4v bits = ldsr<int4v>(tb), rest = ldsr<int4v>(tb + 16u);
                    const int nbits = rest.x, t0 = rest.z;
                    const uint32_t x = (uint32_t)e & 15u;
                    v2d f = (v2d){1.0, 0.0};
                    auto column = [&](int q, uint32_t bit) {
                        const int m = ldsr<int>(init_a + 4u * q);
                        const v2d u = ldsr<v2d>(mat_i + 64u * (uint32_t)(m < 0 ? 0 : m) + 32u * bit);
                        return m < 0 ? (v2d){bit ? 0.0 : 1.0, 0.0} : u;
                    };
                    if (nbits > 0) f = cmul(f, column(bits.x, x & 1u));
                    if (nbits > 1) f = cmul(f, column(bits.y, (x >> 1) & 1u));
                    if (nbits > 2) f = cmul(f, column(bits.z, (x >> 2) & 1u));
                    if (nbits > 3) f = cmul(f, column(bits.w, (x >> 3) & 1u));
                    for (uint32_t m = (uint32_t)t0; m; m &= m - 1u) {      // the extra high thread qubits of this slot (their values: fixed)
                        const int q = __builtin_ctz(m);
                        f = cmul(f, column(q, ((uint32_t)rest.y >> q) & 1u));
                    }
                    if ((int)x >= (1 << nbits)) f = (v2d){1.0, 0.0};
                    ldsw<v2d>(tab_i + 16u * e, f);
                }
            }
            lds_barrier();
        }
        QK_RMARK(11);

        Amps<NA> v;
        double scale = 1.0;
        // this lane's sample of the group and where its terms, matrices and tables are
        const uint32_t sid = PACK ? (uint32_t)tid >> (nq - RB) : 0u;
        const uint32_t mat_l = PACK ? mats_i + mat_s * sid : mats_i, dtab_l = PACK ? side_i + dtab_s * sid : side_i;
        const uint32_t itab_l = PACK ? itab_a + itab_s * sid : itab_a;
        const uint32_t cs_l = cs_a + 16u * ((uint32_t)(it * n_slot) + sid);      // angle e of this lane's sample: + (e << sbits) pairs
        const int n_lo = nq - RB < 4 ? nq - RB : 4, n_hi = nq - RB - n_lo;      // thread qubits behind the two thread tables (n_hi > 4: second slot)
        for (int sg = 0; sg < n_seg; ++sg) {
            // the segment record: 16 words, one scalar load
            const CWords sr = c_seg + 16 * sg;
            const int g0 = sr[0], g1 = sr[1], g2 = sr[2], g3 = RB > 3 ? sr[3] : 0;
            const int op0 = sr[4], op1 = sr[5], nrun = sr[6];
            // the thread's part of the amplitude index (runs of consecutive bits, set up by the planner)
            auto run = [&](int r) { const uint32_t w = (uint32_t)sr[7 + r]; return ((uint32_t)tid & (w & 0xffffu)) << (w >> 16); };
            uint32_t wt = run(0);
            if (nrun > 1) wt |= run(1);
            if (nrun > 2) wt |= run(2);
            if (nrun > 3) wt |= run(3);
            if (nrun > 4) wt |= run(4);
            const uint32_t st16 = state_a | (swz(wt) << 4);      // (state_a has none of the image's address bits: checked at the top)
            // register k's part of the index and of the swizzled LDS address: both are XOR-linear over the register
            // bits, so four scalars per segment (not a swizzle per register and access) make all sixteen
            const uint32_t gb0 = 1u << g0, gb1 = 1u << g1, gb2 = 1u << g2, gb3 = RB > 3 ? 1u << g3 : 0u;
            const uint32_t sb0 = swz(gb0) << 4, sb1 = swz(gb1) << 4, sb2 = swz(gb2) << 4, sb3 = RB > 3 ? swz(gb3) << 4 : 0u;
            auto kdep = [&](int k) { return ((k & 1) ? gb0 : 0u) | ((k & 2) ? gb1 : 0u) | ((k & 4) ? gb2 : 0u) | ((k & 8) ? gb3 : 0u); };
            auto kswz = [&](int k) { return ((k & 1) ? sb0 : 0u) ^ ((k & 2) ? sb1 : 0u) ^ ((k & 4) ? sb2 : 0u) ^ ((k & 8) ? sb3 : 0u); };
            QK_RMARK(1);
            if (sg > 0) {
                lds_barrier();      // every thread has written its amplitudes in the previous layout
#pragma unroll
                for (int k = 0; k < NA; ++k) v[k] = ldsr<v2d>(st16 ^ kswz(k));
            }
            QK_RMARK(2);
            // An op record is 8 words: one scalar load, issued one op ahead -- read field by field where it is needed,
            // every op paid two or three scalar-load latencies in a row (each behind an lgkmcnt(0) that also waits
            // for the LDS traffic in flight).
            int r_kind = c_op[8 * op0], r_a = c_op[8 * op0 + 1], r_b = c_op[8 * op0 + 2], r_tab = c_op[8 * op0 + 3],
                r_t0 = c_op[8 * op0 + 4], r_ntx = c_op[8 * op0 + 5], r_nrt = c_op[8 * op0 + 6], r_flags = c_op[8 * op0 + 7];
            for (int oi = op0; oi < op1; ++oi) {
                const int kind = r_kind, oa = r_a, ob = r_b, o_tab = r_tab, o_t0 = r_t0, o_ntx = r_ntx, o_nrt = r_nrt, o_flags = r_flags;
                {
                    const int nx = 8 * (oi + 1 < op1 ? oi + 1 : oi);
                    r_kind = c_op[nx]; r_a = c_op[nx + 1]; r_b = c_op[nx + 2]; r_tab = c_op[nx + 3];
                    r_t0 = c_op[nx + 4]; r_ntx = c_op[nx + 5]; r_nrt = c_op[nx + 6]; r_flags = c_op[nx + 7];
                }
                if (kind == RG_U1) {
                    if (oa == NA - 1 && o_flags == (RB == 4 ? 0xff : 0x3f)) {      // a Hadamard on every register qubit: no matrices
hadamard_pos<0, NA>(v);
                        hadamard_pos<1, NA>(v);
                        hadamard_pos<2, NA>(v);
                        hadamard_pos<3, NA>(v);
                        scale *= RB == 4 ? 0.25 : 0.35355339059327376220;
                    } else {
                        // the matrix of the next position is read while the butterflies of the current one run
                        auto mload = [&](int p, v2d (&m)[4]) __attribute__((always_inline)) {
                            const uint32_t ma = mat_l + 64u * (uint32_t)((ob >> (8 * p)) & 255);
                            const bool need = ((oa >> p) & 1) && ((o_flags >> (2 * p)) & 3) != 3;      // wave-uniform
                            if (need) { m[0] = ldsr<v2d>(ma); m[1] = ldsr<v2d>(ma + 16u); m[2] = ldsr<v2d>(ma + 32u); m[3] = ldsr<v2d>(ma + 48u); }
                            else { m[0] = m[1] = m[2] = m[3] = (v2d){0.0, 0.0}; }
                        };
                        if constexpr (RB > 3) {
                            v2d ma[4], mb[4];
                            mload(0, ma);
                            mload(1, mb);
                            if (oa & 1) u1_reg<0, NA, PACK>(v, ma[0], ma[1], ma[2], ma[3], o_flags & 3, scale);
                            mload(2, ma);
                            if (oa & 2) u1_reg<1, NA, PACK>(v, mb[0], mb[1], mb[2], mb[3], (o_flags >> 2) & 3, scale);
                            mload(3, mb);
                            if (oa & 4) u1_reg<2, NA, PACK>(v, ma[0], ma[1], ma[2], ma[3], (o_flags >> 4) & 3, scale);
                            if (oa & 8) u1_reg<3, NA, PACK>(v, mb[0], mb[1], mb[2], mb[3], (o_flags >> 6) & 3, scale);
                        } else {      // 128 registers: one matrix at a time
                            v2d ma[4];
                            if (oa & 1) { mload(0, ma); u1_reg<0, NA, PACK>(v, ma[0], ma[1], ma[2], ma[3], o_flags & 3, scale); }
                            if (oa & 2) { mload(1, ma); u1_reg<1, NA, PACK>(v, ma[0], ma[1], ma[2], ma[3], (o_flags >> 2) & 3, scale); }
                            if (oa & 4) { mload(2, ma); u1_reg<2, NA, PACK>(v, ma[0], ma[1], ma[2], ma[3], (o_flags >> 4) & 3, scale); }
                        }
                    }
                } else if (kind == RG_DIAG || kind == RG_INIT) {
                    // table A (the register bits), then the tables of the low and of the high thread bits: consecutive for the
                    // product state; a diagonal stage names each of them (stages that repeat share their tables)
                    const uint32_t ta = (kind == RG_INIT ? itab_l : dtab_l) + 256u * ((uint32_t)o_tab & 0xffffu);
                    const uint32_t tl = kind == RG_INIT ? ta + 256u : dtab_l + 256u * (uint32_t)oa;
                    const uint32_t th = kind == RG_INIT ? ta + 512u : dtab_l + 256u * (uint32_t)ob;
                    // the sixteen entries of the register table: on their way before anything else of the stage, so that the
                    // waits below (thread tables, terms) cover them as well
                    v2d ph[NA];
                    if (kind == RG_DIAG || !(o_flags & 1)) {
#pragma unroll
                        for (int k = 0; k < NA; ++k) ph[k] = ldsr<v2d>(ta + 16u * k);
                    } else {
#pragma unroll
                        for (int k = 0; k < NA; ++k) ph[k] = (v2d){1.0, 0.0};
                    }
                    v2d bt;
                    if constexpr (PACK) {      // the thread's bits above the circuit's qubits number the sample: not part of the table index
                        bt = ldsr<v2d>(tl + 16u * ((uint32_t)tid & ((1u << n_lo) - 1u)));
                        if (n_hi > 0) bt = cmul(bt, ldsr<v2d>(th + 16u * (((uint32_t)tid >> 4) & ((1u << n_hi) - 1u))));
                    } else {
                        const uint32_t tlo = (uint32_t)tid & 15u, thi = (uint32_t)tid >> 4;
                        bt = cmul(ldsr<v2d>(tl + 16u * tlo), ldsr<v2d>(th + 16u * thi));   // (thi >= 16: the second Bhi slot follows the first)
                    }
                    if (kind == RG_INIT) {
                        const int all_h = o_flags & 1;
                        if (all_h) {
#pragma unroll
                            for (int k = 0; k < NA; ++k) v[k] = (v2d){1.0, 0.0};
                            scale = init_c;
                        } else {
#pragma unroll
                            for (int k = 0; k < NA; ++k) v[k] = cmul(bt, ph[k]);
                        }
                    } else {
                        const int t0 = o_t0, nTX = o_ntx, nRT = o_nrt, rtmask = o_flags & 15;
                        const bool fresh = (o_flags & 256) != 0;      // the state is the identity times the running scale: write, do not multiply
                        bt = (v2d){bt.x * scale, bt.y * scale};
                        scale = 1.0;
                        // terms between thread qubits of different tables, then terms between a register and a thread qubit (unless
                        // those came as ratio tables): record and angle of the next term are on their way while one is computed
                        const int nT = nTX + nRT, t_last = t0 + (nT > 0 ? nT - 1 : 0);
                        TermS tr1 = TermS{0, 0, 0, 0, -1}, tr2 = tr1;
                        v2d tc1 = (v2d){1.0, 0.0};
                        if (nT > 0) {
                            tr1 = term_rec(c_term, t0);
                            tr2 = term_rec(c_term, t0 + 1 < t_last ? t0 + 1 : t_last);
                            tc1 = term_cs(tr1, cs_l, sbits);
                        }
                        auto next_term = [&](int k, TermS& t, v2d& cs) __attribute__((always_inline)) {
                            t = tr1; cs = tc1;
                            tr1 = tr2;
                            tr2 = term_rec(c_term, t0 + k + 2 < t_last ? t0 + k + 2 : t_last);
                            tc1 = term_cs(tr1, cs_l, sbits);
                        };
                        for (int k = 0; k < nTX; ++k) {
                            TermS t; v2d cs;
                            next_term(k, t, cs);
                            bt = cmul(bt, term_factor(t, cs, wt));
                        }
                        v2d r0 = (v2d){1.0, 0.0}, r1 = r0, r2 = r0, r3 = r0;
                        if (o_flags & 512) {
                            // The terms between a register qubit and a thread qubit come ready-made: per register position
                            // what flipping that bit multiplies the phase by, as a product of a table over the low and one
                            // over the high thread bits (behind the stage's own three tables, in order of position).
                            const uint32_t hs = n_hi > 4 ? 1u << (n_hi - 4) : (n_hi > 0 ? 1u : 0u);      // slots of the high-bits table
                            uint32_t qa = dtab_l + 256u * ((uint32_t)o_tab >> 16);
                            auto ratio = [&](uint32_t at) __attribute__((always_inline)) {
                                if constexpr (PACK) {
                                    v2d q = ldsr<v2d>(at + 16u * ((uint32_t)tid & ((1u << n_lo) - 1u)));
                                    if (n_hi > 0) q = cmul(q, ldsr<v2d>(at + 256u + 16u * (((uint32_t)tid >> 4) & ((1u << n_hi) - 1u))));
                                    return q;
                                } else {
                                    return cmul(ldsr<v2d>(at + 16u * ((uint32_t)tid & 15u)), ldsr<v2d>(at + 256u + 16u * ((uint32_t)tid >> 4)));
                                }
                            };
                            if (rtmask & 1) { r0 = ratio(qa); qa += 256u * (1u + hs); }
                            if (rtmask & 2) { r1 = ratio(qa); qa += 256u * (1u + hs); }
                            if (rtmask & 4) { r2 = ratio(qa); qa += 256u * (1u + hs); }
                            if (rtmask & 8) { r3 = ratio(qa); }
                        }
                        for (int k = 0; k < nRT; ++k) {
                            TermS t; v2d cs;
                            next_term(nTX + k, t, cs);
                            const int rp = t.rpos;
                            const uint32_t gb = rp == 0 ? 1u << g0 : (rp == 1 ? 1u << g1 : (rp == 2 ? 1u << g2 : 1u << g3));
                            const v2d f0 = term_factor(t, cs, wt), f1 = term_factor(t, cs, wt | gb);
                            bt = cmul(bt, f0);
                            const v2d q = cmulc(f1, f0);      // unit modulus: f1 / f0
                            if (rp == 0) r0 = cmul(r0, q);
                            else if (rp == 1) r1 = cmul(r1, q);
                            else if (rp == 2) r2 = cmul(r2, q);
                            else r3 = cmul(r3, q);
                        }
                        QK_RMARK(10);
                        // The phase of every register state by a Gray-code walk over the register bits -- one multiplication by a
                        // position's ratio (or its conjugate: unit modulus) where that bit flips and the stage has terms on
                        // it (wave-uniform: a branch per step, not a select between two products) -- and the amplitudes
                        // multiplied in place, stage by stage over all of them (see the butterflies above).
                        {
                            double Xx = bt.x, Xy = bt.y;
#pragma unroll
                            for (int i = 0; i < NA; ++i) {
                                const int g = i ^ (i >> 1);
                                if (i > 0) {
                                    const int b = __builtin_ctz(i);
                                    if ((rtmask >> b) & 1) {
                                        asm volatile("" : "+v"(Xx), "+v"(Xy));      // (keeps the branch: nothing here may be computed speculatively)
                                        const v2d q = b == 0 ? r0 : (b == 1 ? r1 : (b == 2 ? r2 : r3));
                                        const v2d X2 = ((g >> b) & 1) ? cmul((v2d){Xx, Xy}, q) : cmulc((v2d){Xx, Xy}, q);
                                        Xx = X2.x; Xy = X2.y;
                                    }
                                }
                                ph[g] = cmul((v2d){Xx, Xy}, ph[g]);
                            }
                        }
                        if (fresh) {
#pragma unroll
                            for (int k = 0; k < NA; ++k) v[k] = ph[k];
                        } else {
                            double t[NA];
#pragma unroll
                            for (int k = 0; k < NA; ++k) t[k] = v.re[k] * ph[k].y;
#pragma unroll
                            for (int k = 0; k < NA; ++k) ip_mul<true>(v.re[k], ph[k].x, t[k]);
#pragma unroll
                            for (int k = 0; k < NA; ++k) ip_fma_acc<true, true>(v.re[k], ph[k].y, v.im[k]);
#pragma unroll
                            for (int k = 0; k < NA; ++k) ip_fma_self<true>(v.im[k], ph[k].x, t[k], v.re[k], v.re[k], v.re[k]);
                        }
                    }
                } else if (kind == RG_CX) {
                    with_pos(oa, [&](auto CC) __attribute__((always_inline)) {
                        with_pos(ob, [&](auto TT) __attribute__((always_inline)) { cx_reg<decltype(CC)::value, decltype(TT)::value, NA>(v); });
                    });
                } else if (kind == RG_CXT) {
                    const bool on = (((uint32_t)tid >> oa) & 1u) != 0u;
                    with_pos(ob, [&](auto TT) __attribute__((always_inline)) { cx_thread<decltype(TT)::value, NA>(v, on); });
                } else {      // RG_ECR
                    with_pos(oa, [&](auto P0) __attribute__((always_inline)) {
                        with_pos(ob, [&](auto P1) __attribute__((always_inline)) { ecr_reg<decltype(P0)::value, decltype(P1)::value, NA>(v); });
                    });
                    scale *= 0.70710678118654752440;
                }
#ifdef QK_REG_TRACE
                if (kind == RG_DIAG || kind == RG_INIT) QK_RMARK(3); else if (kind == RG_U1) QK_RMARK(4); else QK_RMARK(5);
#endif
            }
            if (sg + 1 < n_seg) {
                // to the thread's own slots of this layout (the next layout's reads follow the barrier at the top)
#pragma unroll
                for (int k = 0; k < NA; ++k) ldsw<v2d>(st16 ^ kswz(k), v[k]);
                QK_RMARK(6);
            } else {
                if (scale != 1.0) {
#pragma unroll
                    for (int k = 0; k < NA; ++k) v[k] = (v2d){v[k].x * scale, v[k].y * scale};
                }
                // a wave-uniform base per register (scalar registers) plus ONE 32-bit lane offset.  (PACK: `sample` is the
                // group, a.D its 2^L amplitudes; a sample past the end of the batch is a copy of sample N - 1 and goes there.)
                uint32_t wo = wt;
                if constexpr (PACK) {
                    const int64_t last = a.N - 1 - (sample << sbits);      // >= 0
                    if ((int64_t)sid > last) wo = (wt & ((1u << nq) - 1u)) | ((uint32_t)last << nq);
                }
                if constexpr (RDM) {
                    // the projected kernel's mode: the state goes to the image once more, in index order, and the per-qubit
                    // reduced density matrices are reduced from there (qk_rdm.h, as in the LDS simulator's fused mode);
                    // Psi is never written.  One sample per workgroup (the host does not take this path for packed plans).
#pragma unroll
                    for (int k = 0; k < NA; ++k) ldsw<v2d>(state_a + ((wt | kdep(k)) << 4), v[k]);
                    __syncthreads();
                    rdm1_from_lds((const double2*)lds + (state_a >> 4), nq, (double*)((char*)lds + a.red_off), a.rdm + sample * (int64_t)(4 * nq), tid, T);
                } else if (a.psi32) {
                    char* gp = (char*)(a.psi32 + sample * a.D);
                    uint32_t off = wo << 3;
#pragma unroll
                    for (int k = 0; k < NA; ++k) {
                        asm volatile("" : "+v"(off));      // keeps zext(off) at the store: selects the SGPR-base form
                        *(float2*)(gp + ((uint64_t)kdep(k) << 3) + (uint64_t)off) = make_float2((float)v[k].x, (float)v[k].y);
                    }
                } else {
                    char* gp = (char*)(a.psi + sample * a.D);
                    uint32_t off = wo << 4;
#pragma unroll
                    for (int k = 0; k < NA; ++k) {
                        asm volatile("" : "+v"(off));
                        *(v2d*)(gp + ((uint64_t)kdep(k) << 4) + (uint64_t)off) = v[k];
                    }
                }
                QK_RMARK(6);
            }
        }
    }
#ifdef QK_REG_TRACE
    if (tid0 == 0 && a.trace)
        for (int k = 0; k < 12; ++k) a.trace[(int64_t)blockIdx.x * 12 + k] = k != 7 ? tr[k] : (double)(s_end - s_begin);
#endif
}

// ---- pre-pass: what a sample needs beside its amplitudes, for every sample of the batch ---------------------------
// Layout of the result (RegArgs::side): per group of S = 2^(L - nq) samples one contiguous block -- S x n_dtab tables of 16
// entries, then S x n_mat matrices of 4 entries -- which the simulator fetches with LDS-DMA as it is.  The batch is
// padded to whole groups with copies of the last sample.
struct SideShape { int sb, n_dtab, n_mat; int64_t piece; };      // piece: 16-byte entries per group
__device__ __forceinline__ SideShape side_shape(CWords hw) {
    SideShape s;
    s.sb = hw[0] - hw[21];
    s.n_dtab = hw[3] - hw[24];
    s.n_mat = hw[6];
    s.piece = ((int64_t)(s.n_dtab * 16 + s.n_mat * 4)) << s.sb;
    return s;
}

// The phase tables of the diagonal stages: one thread per (sample, table), the samples of a wave consecutive.  Which
// entries of a table take a term's factor (or its conjugate) is the same for every sample -- two 16-bit masks per
// (table, term) from the planner -- so the threads of a wave run one instruction stream of coalesced angle loads and
// complex multiplications, without a select: inside the simulator the same products were one lane per ENTRY of ONE
// sample, each a chain of LDS reads and selects (a fifth of a 12-qubit sample's time, most of an 8-qubit one's).
__global__ void __launch_bounds__(256) reg_table_kernel(const char* blob, const double2* cs, int64_t N, int64_t n_pad, double2* out) {
    const CWords hw = cwords(blob);
    const SideShape sh = side_shape(hw);
    const int n_itab = hw[24];
    const int table = blockIdx.y;
    const int64_t sample = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (sample - (threadIdx.x & 63) >= n_pad) return;      // (whole waves only: every lane of a wave carries entries out below)
    const int64_t src = sample < N ? sample : N - 1;
    const CWords rec = cwords(blob + hw[12] + 32 * (n_itab + table));
    const int t0 = rec[6], nt = rec[7];
    const CWords tt = cwords(blob + hw[26]) + 2 * t0;
    v2d f[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) f[x] = (v2d){1.0, 0.0};
    for (int k = 0; k < nt; ++k) {
        const int expr = tt[2 * k];
        const uint32_t masks = (uint32_t)tt[2 * k + 1];
        v2d u = (v2d){-1.0, 0.0};      // cz
        if (expr >= 0) {
            const double2 g = cs[(int64_t)expr * N + src];
            u = (v2d){g.x, g.y};
        }
        const v2d uc = (v2d){u.x, -u.y};
#pragma unroll
        for (int x = 0; x < 16; ++x)
            if ((masks >> x) & 1u) f[x] = cmul(f[x], ((masks >> (16 + x)) & 1u) ? uc : u);      // wave-uniform tests
    }
    // A lane's table is 256 contiguous bytes, but the lanes of a wave are n_dtab tables apart: stored straight from the
    // registers every store instruction touches 64 different lines, 16 bytes each.  The wave turns its 64 x 16 entries
    // through LDS (two halves of 8 entries, 144 bytes per lane: no bank conflicts either way) so that 8 consecutive lanes
    // write one whole 128-byte line.
    __shared__ v2d turn[256 * 9];
    const int lane = threadIdx.x & 63;
    v2d* mine = turn + (threadIdx.x & ~63) * 9;      // the wave's part; waves do not exchange anything
    const int64_t wave_first = sample - lane;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int x = 0; x < 8; ++x) mine[lane * 9 + x] = f[8 * h + x];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the wave's own LDS writes have landed
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int l2 = 8 * i + (lane >> 3);      // whose entries this lane carries out
            const int64_t smp = wave_first + l2;
            const v2d val = mine[l2 * 9 + (lane & 7)];
            if (smp < n_pad) {
                const int64_t s_in = smp & (((int64_t)1 << sh.sb) - 1);
                v2d* o = (v2d*)out + (smp >> sh.sb) * sh.piece + ((s_in * sh.n_dtab + table) << 4);
                o[8 * h + (lane & 7)] = val;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The 2x2 matrices the simulator applies (single gates and runs of gates on one qubit, multiplied out): one thread per
// (sample, matrix), uniform control flow per wave as above.  A gate multiplies the two rows of the product so far by
// its own shape -- a rotation about z scales them, about x or y mixes them with real weights, X swaps them -- instead of
// going through a general complex 2x2 product (8 complex multiplications per gate; a run of a GA genome has up to
// gates-per-qubit factors), and the angles of up to four factors are loaded before the first of them is used.
__device__ __forceinline__ void matrix_factor(int gate, double cn, double sn, v2d& a, v2d& b) {      // (a, b) <- G (a, b)
    const double r = 0.70710678118654752440;
    const v2d a0 = a, b0 = b;
    if (gate == QK_H) { a = (v2d){r * (a0.x + b0.x), r * (a0.y + b0.y)}; b = (v2d){r * (a0.x - b0.x), r * (a0.y - b0.y)}; }
    else if (gate == QK_X) { a = b0; b = a0; }
    else if (gate == QK_SX) {      // 0.5 [[1 + i, 1 - i], [1 - i, 1 + i]]
        a = (v2d){0.5 * ((a0.x - a0.y) + (b0.x + b0.y)), 0.5 * ((a0.x + a0.y) + (b0.y - b0.x))};
        b = (v2d){0.5 * ((a0.x + a0.y) + (b0.x - b0.y)), 0.5 * ((a0.y - a0.x) + (b0.x + b0.y))};
    }
    else if (gate == QK_RX) {      // [[c, -i s], [-i s, c]]
        a = (v2d){fma(cn, a0.x, sn * b0.y), fma(cn, a0.y, -(sn * b0.x))};
        b = (v2d){fma(cn, b0.x, sn * a0.y), fma(cn, b0.y, -(sn * a0.x))};
    }
    else if (gate == QK_RZ) {      // diag(c - i s, c + i s)
        a = (v2d){fma(cn, a0.x, sn * a0.y), fma(cn, a0.y, -(sn * a0.x))};
        b = (v2d){fma(cn, b0.x, -(sn * b0.y)), fma(cn, b0.y, sn * b0.x)};
    }
    else if (gate == QK_P) { b = (v2d){fma(cn, b0.x, -(sn * b0.y)), fma(cn, b0.y, sn * b0.x)}; }      // diag(1, c + i s)
    else {      // ry: [[c, -s], [s, c]]
        a = (v2d){fma(cn, a0.x, -(sn * b0.x)), fma(cn, a0.y, -(sn * b0.y))};
        b = (v2d){fma(cn, b0.x, sn * a0.x), fma(cn, b0.y, sn * a0.y)};
    }
}

__global__ void __launch_bounds__(256) reg_matrix_kernel(const char* blob, const double2* cs, int64_t N, int64_t n_pad, double2* out) {
    const CWords hw = cwords(blob);
    const SideShape sh = side_shape(hw);
    const int lane = threadIdx.x & 63;
    const int64_t sample = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t wave_first = sample - lane;
    if (wave_first >= n_pad) return;      // (whole waves only: every lane of a wave carries entries out below)
    const int64_t src = sample < N ? sample : N - 1;
    // A sample's matrices are contiguous, 64 bytes each; the lanes of a wave are n_mat matrices apart.  A block does its
    // share of the matrices two at a time and the wave turns each pair through LDS as reg_table_kernel does its halves:
    // 8 consecutive lanes write 128 contiguous bytes, and the pairs of one sample follow each other from the same CU.
    __shared__ v2d turn[256 * 9];
    v2d* mine = turn + (threadIdx.x & ~63) * 9;
    const int n_pairs = (sh.n_mat + 1) >> 1;
    const int per_block = (n_pairs + (int)gridDim.y - 1) / (int)gridDim.y;
    const int p_end = min(n_pairs, ((int)blockIdx.y + 1) * per_block);
    for (int p = (int)blockIdx.y * per_block; p < p_end; ++p) {
        v2d m[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int u_i = 2 * p + half;
            v2d m0 = (v2d){1, 0}, m1 = (v2d){0, 0}, m2 = (v2d){0, 0}, m3 = (v2d){1, 0};
            if (u_i < sh.n_mat) {
                const CWords rec = cwords(blob + hw[15]) + 2 * u_i;
                const int first = rec[0], count = rec[1];
                const CWords fc = cwords(blob + hw[17]) + 2 * first;
                for (int k0 = 0; k0 < count; k0 += 4) {
                    double2 g[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        g[j] = make_double2(1.0, 0.0);
                        if (k0 + j < count) {
                            const int expr = fc[2 * (k0 + j) + 1];
                            if (expr >= 0) g[j] = cs[(int64_t)expr * N + src];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k0 + j < count) {
                            const int gate = fc[2 * (k0 + j)];
                            matrix_factor(gate, g[j].x, g[j].y, m0, m2);      // first column
                            matrix_factor(gate, g[j].x, g[j].y, m1, m3);      // second column
                        }
                }
            }
            m[4 * half + 0] = m0; m[4 * half + 1] = m1; m[4 * half + 2] = m2; m[4 * half + 3] = m3;
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) mine[lane * 9 + x] = m[x];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        const int e = lane & 7;
        const bool exists = 2 * p + (e >> 2) < sh.n_mat;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int l2 = 8 * i + (lane >> 3);
            const int64_t smp = wave_first + l2;
            const v2d val = mine[l2 * 9 + e];
            if (exists && smp < n_pad) {
                const int64_t s_in = smp & (((int64_t)1 << sh.sb) - 1);
                v2d* o = (v2d*)out + (smp >> sh.sb) * sh.piece + (((int64_t)sh.n_dtab << sh.sb) << 4) + ((s_in * sh.n_mat + 2 * p) << 2);
                o[e] = val;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

bool reg_rdm_mode_ok(const qk_circuit& c) {
    if (c.reg_blob.empty()) return false;
    const RegHeader* H = (const RegHeader*)c.reg_blob.data();
    // (one sample per workgroup; 16 amplitudes per thread: with 8 the workgroup has twice the waves at 128 registers each, and
    // the reduction -- 16 amplitudes and 16 partial sums per thread -- spills there)
    if (H->nq != H->L || H->nq > RDM_MAXQ || H->rb != 4) return false;
    const size_t threads = (size_t)1 << (H->L - H->rb);
    return (c.reg_lds_bytes + 15) / 16 * 16 + sizeof(double) * RDM_SLOTS * (threads / 64) <= (size_t)160 * 1024;
}

// bytes of the tables and matrices of a batch of N samples (0: the circuit has neither)
size_t reg_side_bytes(const qk_circuit& c, int64_t N) {
    const RegHeader* H = (const RegHeader*)c.reg_blob.data();
    const int sb = H->L - H->nq;
    const int64_t n_group = (N + ((int64_t)1 << sb) - 1) >> sb;
    return (size_t)n_group * ((size_t)((H->n_tab - H->n_itab) * 16 + H->n_mat * 4) << sb) * 16;
}

int launch_reg_side(const qk_circuit& c, const void* blob_dev, const double2* cs, int64_t N, void* tabs, hipStream_t stream) {
    const RegHeader* H = (const RegHeader*)c.reg_blob.data();
    const int n_dtab = H->n_tab - H->n_itab, sb = H->L - H->nq;
    if (N <= 0) return QK_OK;
    const int64_t n_pad = ((N + ((int64_t)1 << sb) - 1) >> sb) << sb;
    if ((n_pad + 255) / 256 > 0x7fffffff) return fail(QK_ERR_UNSUPPORTED, "register-resident simulator: batch too large");
    if (n_dtab > 0)
        hipLaunchKernelGGL(reg_table_kernel, dim3((unsigned)((n_pad + 255) / 256), (unsigned)n_dtab), dim3(256), 0, stream,
                           (const char*)blob_dev, cs, N, n_pad, (double2*)tabs);
    if (H->n_mat > 0) {
        // pairs of matrices; one block takes all pairs of its 256 samples unless the batch is too small to fill the chip that way
        const int64_t bx = (n_pad + 255) / 256;
        const int n_pairs = (H->n_mat + 1) / 2;
        const int by = (int)std::max<int64_t>(1, std::min<int64_t>(n_pairs, (1024 + bx - 1) / bx));
        hipLaunchKernelGGL(reg_matrix_kernel, dim3((unsigned)bx, (unsigned)by), dim3(256), 0, stream,
                           (const char*)blob_dev, cs, N, n_pad, (double2*)tabs);
    }
    QK_HIP(hipGetLastError());
    return QK_OK;
}

int launch_reg_sim(const qk_circuit& c, const void* blob_dev, const double2* cs, const void* tabs, int64_t N, void* psi, void* psi32,
                   hipStream_t stream, double* rdm) {
    const RegHeader* H = (const RegHeader*)c.reg_blob.data();
    const int L = H->L, RB = H->rb;      // L > n_qubits: 2^(L - n_qubits) samples per workgroup
    const bool pack = H->nq < L;
    // [8 amplitudes per thread | 16 | 16, packed samples | 8, packed samples][L - 10]
    const void* fns[4][4] = {{(const void*)reg_sim_kernel<7, 3, false>, (const void*)reg_sim_kernel<8, 3, false>, (const void*)reg_sim_kernel<9, 3, false>, (const void*)reg_sim_kernel<10, 3, false>},
                             {(const void*)reg_sim_kernel<6, 4, false>, (const void*)reg_sim_kernel<7, 4, false>, (const void*)reg_sim_kernel<8, 4, false>, (const void*)reg_sim_kernel<9, 4, false>},
                             {(const void*)reg_sim_kernel<6, 4, true>, (const void*)reg_sim_kernel<7, 4, true>, (const void*)reg_sim_kernel<8, 4, true>, nullptr},
                             {(const void*)reg_sim_kernel<7, 3, true>, (const void*)reg_sim_kernel<8, 3, true>, (const void*)reg_sim_kernel<9, 3, true>, nullptr}};
    const void* fns_rdm[4] = {(const void*)reg_sim_kernel<6, 4, false, true>, (const void*)reg_sim_kernel<7, 4, false, true>,
                              (const void*)reg_sim_kernel<8, 4, false, true>, (const void*)reg_sim_kernel<9, 4, false, true>};
    if (H->nq != c.n_qubits || L < kRegMinQubits || L > kRegMaxQubits || (RB != 3 && RB != 4) || (pack && L > kRegPackMaxWidth))
        return fail(QK_ERR_INVALID, "register-resident simulator: malformed plan");
    static DeviceOnce attr_set;
    if (attr_set.pending()) {
        for (int r = 0; r < 4; ++r)
            for (int k = 0; k < 4; ++k)
                if (fns[r][k]) QK_HIP(hipFuncSetAttribute(fns[r][k], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int k = 0; k < 4; ++k) QK_HIP(hipFuncSetAttribute(fns_rdm[k], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.done();
    }
    RegArgs a{};
    a.blob = (const char*)blob_dev;
    a.cs = cs;
    a.side = (const double2*)tabs;
    a.N = N;
    a.psi = (double2*)psi;
    a.psi32 = (float2*)psi32;
    a.D = (int64_t)1 << L;
#ifdef QK_REG_TRACE
    static double* trace_buf = nullptr;
    if (!trace_buf) QK_HIP(hipMalloc((void**)&trace_buf, 12 * sizeof(double) * 256 * 16));
    QK_HIP(hipMemsetAsync(trace_buf, 0, 12 * sizeof(double) * 256 * 16, stream));
    a.trace = trace_buf;
#endif
    const int threads = 1 << (L - RB);
    size_t lds = c.reg_lds_bytes;
    if (rdm) {
        if (!reg_rdm_mode_ok(c)) return fail(QK_ERR_INVALID, "register-resident simulator: no reduced-density-matrix mode for this plan");
        a.rdm = rdm;
        a.red_off = (int)((lds + 15) / 16 * 16);
        lds = (size_t)a.red_off + sizeof(double) * RDM_SLOTS * (size_t)(threads / 64);
        if (lds > (size_t)160 * 1024) return fail(QK_ERR_UNSUPPORTED, "register-resident simulator: no LDS left for the reduction");
    }
    // Resident workgroups per CU, as the runtime counts them for this kernel's registers and this plan's LDS: the grid is
    // exactly what the chip holds at once, every workgroup walking its share of the sample groups.  (Asked, not assumed:
    // a grid sized for fewer workgroups than a CU takes lets the dispatcher fill some CUs and leave others short.)
    const void* fn = rdm ? fns_rdm[L - 10] : fns[pack ? (RB == 3 ? 3 : 2) : (RB == 3 ? 0 : 1)][L - 10];
    int per_cu = 0;
    {
        static std::mutex mu;
        static std::map<std::tuple<int, const void*, size_t>, int> cache;
        int dev = 0;
        QK_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        const auto key = std::make_tuple(dev, fn, lds);
        const auto it = cache.find(key);
        if (it != cache.end()) per_cu = it->second;
        else {
            QK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds));
            cache[key] = per_cu;
        }
    }
    if (const char* e = std::getenv("QK_REG_PER_CU")) per_cu = std::atoi(e);      // A/B runs
    per_cu = std::max(1, per_cu);
    static const bool debug = std::getenv("QK_REG_DEBUG") != nullptr;
    if (debug) std::fprintf(stderr, "[reg] L=%d nq=%d RB=%d threads=%d lds=%zu workgroups per CU=%d\n", L, H->nq, RB, threads, lds, per_cu);
    const int64_t n_group = (N + ((int64_t)1 << (L - H->nq)) - 1) >> (L - H->nq);
    const int64_t nblocks = std::min<int64_t>(n_group, (int64_t)256 * per_cu);
    const dim3 grid((unsigned)nblocks), block(threads);
    void* args[] = {&a};
    QK_HIP(hipLaunchKernel(fn, grid, block, args, lds, stream));
    QK_HIP(hipGetLastError());
#ifdef QK_REG_TRACE
    {
        QK_HIP(hipStreamSynchronize(stream));
        std::vector<double> h(12 * (size_t)nblocks);
        QK_HIP(hipMemcpy(h.data(), trace_buf, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        double sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t b = 0; b < nblocks; ++b)
            for (int k = 0; k < 12; ++k) sum[k] += h[12 * b + k];
        std::fprintf(stderr, "[reg trace] L=%d nq=%d RB=%d blocks=%lld groups=%.0f us per sample group and workgroup:", L, H->nq, RB, (long long)nblocks, sum[7]);
        const char* names[12] = {"wait for angles", "segment setup", "transpose read", "init/diag ops", "u1 ops", "other ops", "transpose write / stores", "",
                                 "barrier", "issue next fetch", "diag: terms", "product-state tables"};
        for (int k = 0; k < 12; ++k)
            if (k != 7) std::fprintf(stderr, " %s %.2f", names[k], sum[k] / sum[7] * 0.01);
        std::fprintf(stderr, "\n");
    }
#endif
    return QK_OK;
}

}  // namespace qk
