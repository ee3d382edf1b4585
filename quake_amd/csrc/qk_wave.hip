// Wave-resident statevector simulator for gfx950: circuits of up to 10 qubits.
//
// The whole statevector of a sample lives in the registers of ONE wavefront: amplitude index
// b = (r << 6) | lane, i.e. qubits 0..5 select the lane and qubits 6..9 one of up to 16 register pairs.
// No LDS image of the state, no barriers, no index arithmetic per amplitude:
//   * a gate on a register qubit is a butterfly between two registers of the same lane;
//   * a gate on a lane qubit fetches the partner amplitude from lane ^ (1 << q) with a cross-lane move
//     (DPP quad_perm / row rotations for qubits 0..3, ds_bpermute -- the LDS crossbar, no LDS memory --
//     for qubits 4 and 5) and combines it with the lane's own;
//   * a run of diagonal gates multiplies the amplitudes by phase factors selected from the lane's and the
//     register's own bits;
//   * Psi leaves as one 1 KiB store per register (64 lanes x 16 B, consecutive amplitudes).
// The per-sample gate angles are evaluated lane-parallel (one lane per angle, several samples per pass when
// the circuit has fewer than 33 angles) into a small LDS table of (cos, sin) pairs that the gate code reads
// as wave-uniform broadcasts.  One wavefront = one workgroup, so every synchronisation is wave-local.
// The circuit arrives as a position-independent blob (qk_wave.h); a launch can serve many circuits at once
// through a job table in device memory (the GA's heterogeneous batch).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "qk_common.h"
#include "qk_math.h"
#include "qk_wave.h"

namespace qk {

typedef double v2d __attribute__((ext_vector_type(2)));   // native vector: stays in registers

struct WaveArgs {
    WaveJob one;            // n_jobs == 1: the job by value (no device table to upload)
    const WaveJob* table;   // n_jobs > 1: device table, sorted by unit_begin
    int32_t n_jobs;
    int32_t cs_entries;     // LDS: (cos, sin) pairs of one unit, then the staged program
    int32_t stage_bytes;    // room for a circuit blob in LDS (0: programs are read from global memory)
    int32_t pad;
    int64_t total_units;
};


namespace {

// The compiler does not see that values loaded from the circuit blob are the same in every lane; told so, it
// keeps them in scalar registers and turns the gate dispatch into scalar branches.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ v2d cmul(v2d a, v2d b) {
    return (v2d){fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)};
}

// value of lane ^ (1 << q): the partner amplitude of a gate on lane qubit q.  KIND 0..3 = qubits 0..3 (DPP:
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E; row_ror:n = 0x120 + n, lane i of a 16-lane row takes lane
// (i - n) mod 16), KIND 4 = qubits 4 and 5 through the LDS crossbar (ds_bpermute, no LDS memory).
template <int KIND>
struct LaneXor {
    int addr;     // byte address of the partner lane for ds_bpermute
    __device__ __forceinline__ LaneXor(int q, int lane) : addr((lane ^ (1 << q)) << 2) {}
    __device__ __forceinline__ double operator()(double v) const {
        // The empty asm statements keep the double <-> integer reinterpretation of the cross-lane builtins in
        // registers: folded into the loads and stores of the amplitude array it leaves that array in scratch.
        double out;
        // every lane has a valid source under these controls, so the "old" operand never shows: bound_ctrl with
        // an old value of 0 lets the instruction stand alone (no register to clear first)
        if constexpr (KIND == 0) out = __builtin_amdgcn_update_dpp(0.0, v, 0xB1, 0xF, 0xF, true);
        else if constexpr (KIND == 1) out = __builtin_amdgcn_update_dpp(0.0, v, 0x4E, 0xF, 0xF, true);
        else if constexpr (KIND == 2) {   // lanes with bit 2 clear (banks 0, 2) take lane + 4, the others lane - 4
            const double t = __builtin_amdgcn_update_dpp(0.0, v, 0x12C, 0xF, 0x5, true);
            out = __builtin_amdgcn_update_dpp(t, v, 0x124, 0xF, 0xA, false);
        } else if constexpr (KIND == 3) out = __builtin_amdgcn_update_dpp(0.0, v, 0x128, 0xF, 0xF, true);
        else {
            const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
            const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
            out = __hiloint2double(hi, lo);
        }
        asm volatile("" : "+v"(out));
        return out;
    }
    __device__ __forceinline__ v2d operator()(v2d v) const { return (v2d){(*this)(v.x), (*this)(v.y)}; }
};

template <int V> struct IC { static constexpr int value = V; };

// f(LaneXor<KIND>) for lane qubit q, f(IC<T>) for register qubit 6 + T: ONE scalar dispatch per gate, the
// body behind it is straight-line code over the registers
template <class F>
__device__ __forceinline__ void with_lane_qubit(int q, int lane, F&& f) {
    switch (q) {
        case 0: f(LaneXor<0>(q, lane)); break;
        case 1: f(LaneXor<1>(q, lane)); break;
        case 2: f(LaneXor<2>(q, lane)); break;
        case 3: f(LaneXor<3>(q, lane)); break;
        default: f(LaneXor<4>(q, lane)); break;
    }
}
template <int NREG, class F>
__device__ __forceinline__ void with_reg_qubit(int q, F&& f) {
    const int t = q - 6;
    if (t == 0) { if constexpr (NREG > 1) f(IC<0>()); }
    else if (t == 1) { if constexpr (NREG > 2) f(IC<1>()); }
    else if (t == 2) { if constexpr (NREG > 4) f(IC<2>()); }
    else { if constexpr (NREG > 8) f(IC<3>()); }
}

// (cos, sin) of angle e of the current sample: wave-uniform LDS broadcast
__device__ __forceinline__ v2d angle_cs(const double2* cs, int e) {
    const double2 v = cs[e];
    return (v2d){v.x, v.y};
}

// column 0 of a single-qubit gate: what it makes of |0>  (u0 = <0|U|0>, u1 = <1|U|0>)
__device__ __forceinline__ void first_column(int gate, v2d cs, v2d& u0, v2d& u1) {
    const double r = 0.70710678118654752440;
    u0 = (v2d){1.0, 0.0};
    u1 = (v2d){0.0, 0.0};
    if (gate == QK_UC) return;     // composite: the caller takes the column from the sample's matrix table
    if (gate == QK_H) { u0 = (v2d){r, 0.0}; u1 = (v2d){r, 0.0}; }
    else if (gate == QK_X) { u0 = (v2d){0.0, 0.0}; u1 = (v2d){1.0, 0.0}; }
    else if (gate == QK_SX) { u0 = (v2d){0.5, 0.5}; u1 = (v2d){0.5, -0.5}; }
    else if (gate == QK_RX) { u0 = (v2d){cs.x, 0.0}; u1 = (v2d){0.0, -cs.y}; }
    else if (gate == QK_RY) { u0 = (v2d){cs.x, 0.0}; u1 = (v2d){cs.y, 0.0}; }
}

// ---- single-qubit gates --------------------------------------------------------------------------
// Lane qubit: own amplitude a, partner p = amplitude of lane ^ (1 << q); hi = this lane has the qubit set.
// Register qubit T: a butterfly between registers r and r | (1 << T).  h and sx leave their factor
// (1/sqrt 2, 1/2) to the caller's running scale.  GATE is a template argument so that every gate is its own
// straight-line block behind a scalar branch.
template <int NREG, int GATE>
__device__ __forceinline__ void u1_gate(v2d (&amp)[NREG], int q, v2d cs, int lane) {
    const double c = cs.x, s = cs.y;
    if (q < 6) {
        const bool hi = (lane >> q) & 1;
        const double sgn = hi ? -1.0 : 1.0;
        const double sp = hi ? s : -s;    // ry: row 0 = (c, -s), row 1 = (s, c)
        with_lane_qubit(q, lane, [&](auto X) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);   // at most 4 partner fetches in flight: bounds the live registers
                const v2d a = amp[r];
                const v2d p = X(a);
                if constexpr (GATE == QK_H) {            // (a + p) for bit 0, (p - a) for bit 1
                    amp[r] = (v2d){fma(a.x, sgn, p.x), fma(a.y, sgn, p.y)};
                } else if constexpr (GATE == QK_X) {
                    amp[r] = p;
                } else if constexpr (GATE == QK_SX) {    // ((a + p) + i (a - p)) / 2 for either bit
                    amp[r] = (v2d){(a.x + p.x) - (a.y - p.y), (a.y + p.y) + (a.x - p.x)};
                } else if constexpr (GATE == QK_RX) {    // c a - i s p
                    amp[r] = (v2d){fma(s, p.y, c * a.x), fma(-s, p.x, c * a.y)};
                } else {                                 // QK_RY
                    amp[r] = (v2d){fma(sp, p.x, c * a.x), fma(sp, p.y, c * a.y)};
                }
            }
        });
    } else {
        with_reg_qubit<NREG>(q, [&](auto TT) __attribute__((always_inline)) {
            constexpr int T = decltype(TT)::value;
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                if (r & (1 << T)) continue;
                const v2d a0 = amp[r], a1 = amp[r | (1 << T)];
                v2d n0, n1;
                if constexpr (GATE == QK_H) {
                    n0 = a0 + a1;
                    n1 = a0 - a1;
                } else if constexpr (GATE == QK_X) {
                    n0 = a1;
                    n1 = a0;
                } else if constexpr (GATE == QK_SX) {
                    const v2d sum = a0 + a1, dif = a0 - a1;
                    n0 = (v2d){sum.x - dif.y, sum.y + dif.x};
                    n1 = (v2d){sum.x + dif.y, sum.y - dif.x};
                } else if constexpr (GATE == QK_RX) {
                    n0 = (v2d){fma(s, a1.y, c * a0.x), fma(-s, a1.x, c * a0.y)};
                    n1 = (v2d){fma(s, a0.y, c * a1.x), fma(-s, a0.x, c * a1.y)};
                } else {   // QK_RY
                    n0 = (v2d){fma(-s, a1.x, c * a0.x), fma(-s, a1.y, c * a0.y)};
                    n1 = (v2d){fma(s, a0.x, c * a1.x), fma(s, a0.y, c * a1.y)};
                }
                amp[r] = n0;
                amp[r | (1 << T)] = n1;
            }
        });
    }
}

// a general 2x2 matrix (the product of a run of single-qubit gates: QK_UC), entries m[0..3] = m00 m01 m10 m11
template <int NREG>
__device__ __forceinline__ void u1_matrix(v2d (&amp)[NREG], int q, const double2* m, int lane) {
    const v2d m00 = (v2d){m[0].x, m[0].y}, m01 = (v2d){m[1].x, m[1].y};
    const v2d m10 = (v2d){m[2].x, m[2].y}, m11 = (v2d){m[3].x, m[3].y};
    if (q < 6) {
        const bool hi = (lane >> q) & 1;
        const v2d ca = hi ? m11 : m00, cp = hi ? m10 : m01;    // new = ca * own + cp * partner
        with_lane_qubit(q, lane, [&](auto X) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                const v2d a = amp[r];
                const v2d p = X(a);
                const v2d t = cmul(ca, a);
                amp[r] = (v2d){fma(cp.x, p.x, fma(-cp.y, p.y, t.x)), fma(cp.x, p.y, fma(cp.y, p.x, t.y))};
            }
        });
    } else {
        with_reg_qubit<NREG>(q, [&](auto TT) __attribute__((always_inline)) {
            constexpr int T = decltype(TT)::value;
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                if (r & (1 << T)) continue;
                const v2d a0 = amp[r], a1 = amp[r | (1 << T)];
                const v2d t0 = cmul(m00, a0), t1 = cmul(m10, a0);
                amp[r] = (v2d){fma(m01.x, a1.x, fma(-m01.y, a1.y, t0.x)), fma(m01.x, a1.y, fma(m01.y, a1.x, t0.y))};
                amp[r | (1 << T)] = (v2d){fma(m11.x, a1.x, fma(-m11.y, a1.y, t1.x)), fma(m11.x, a1.y, fma(m11.y, a1.x, t1.y))};
            }
        });
    }
}

// the matrix of one factor of a composite: h x sx constant, rx ry rz from (cos, sin) of theta / 2, p from those of lambda
__device__ __forceinline__ void factor_matrix(int gate, double c, double s, double2& g0, double2& g1, double2& g2, double2& g3) {
    const double r = 0.70710678118654752440;
    const double2 z = make_double2(0.0, 0.0);
    g0 = make_double2(1.0, 0.0); g1 = z; g2 = z; g3 = make_double2(1.0, 0.0);
    if (gate == QK_H) { g0 = make_double2(r, 0); g1 = make_double2(r, 0); g2 = make_double2(r, 0); g3 = make_double2(-r, 0); }
    else if (gate == QK_X) { g0 = z; g1 = make_double2(1, 0); g2 = make_double2(1, 0); g3 = z; }
    else if (gate == QK_SX) { g0 = make_double2(0.5, 0.5); g1 = make_double2(0.5, -0.5); g2 = make_double2(0.5, -0.5); g3 = make_double2(0.5, 0.5); }
    else if (gate == QK_RX) { g0 = make_double2(c, 0); g1 = make_double2(0, -s); g2 = make_double2(0, -s); g3 = make_double2(c, 0); }
    else if (gate == QK_RY) { g0 = make_double2(c, 0); g1 = make_double2(-s, 0); g2 = make_double2(s, 0); g3 = make_double2(c, 0); }
    else if (gate == QK_RZ) { g0 = make_double2(c, -s); g3 = make_double2(c, s); }
    else if (gate == QK_P) { g3 = make_double2(c, s); }
}
__device__ __forceinline__ double2 cmul2(double2 a, double2 b) {
    return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ double2 cfma2(double2 a, double2 b, double2 c) {
    return make_double2(fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y)));
}

template <int NREG>
__device__ __forceinline__ void u1_any(v2d (&amp)[NREG], int q, int gate, v2d cs, int lane) {
    switch (gate) {
        case QK_H: u1_gate<NREG, QK_H>(amp, q, cs, lane); break;
        case QK_X: u1_gate<NREG, QK_X>(amp, q, cs, lane); break;
        case QK_SX: u1_gate<NREG, QK_SX>(amp, q, cs, lane); break;
        case QK_RX: u1_gate<NREG, QK_RX>(amp, q, cs, lane); break;
        default: u1_gate<NREG, QK_RY>(amp, q, cs, lane); break;
    }
}

// ---- cx -----------------------------------------------------------------------------------------
template <int NREG>
__device__ __forceinline__ void cx_any(v2d (&amp)[NREG], int qc, int qt, int lane) {
    const bool lane_ctl = qc < 6;
    const bool on_lane = lane_ctl && ((lane >> qc) & 1);
    if (qt < 6) {
        with_lane_qubit(qt, lane, [&](auto X) __attribute__((always_inline)) {
            if (lane_ctl) {
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                    const v2d p = X(amp[r]);             // every lane takes part in the exchange
                    amp[r] = on_lane ? p : amp[r];
                }
            } else {
                with_reg_qubit<NREG>(qc, [&](auto CC) __attribute__((always_inline)) {
                    constexpr int C = decltype(CC)::value;
#pragma unroll
                    for (int r = 0; r < NREG; ++r)
                        if (r & (1 << C)) amp[r] = X(amp[r]);
                });
            }
        });
    } else {
        with_reg_qubit<NREG>(qt, [&](auto TT) __attribute__((always_inline)) {
            constexpr int T = decltype(TT)::value;
            if (lane_ctl) {
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    if (r & (1 << T)) continue;
                    const v2d a0 = amp[r], a1 = amp[r | (1 << T)];
                    amp[r] = on_lane ? a1 : a0;
                    amp[r | (1 << T)] = on_lane ? a0 : a1;
                }
            } else {
                with_reg_qubit<NREG>(qc, [&](auto CC) __attribute__((always_inline)) {
                    constexpr int C = decltype(CC)::value;
                    if constexpr (C != T) {
#pragma unroll
                        for (int r = 0; r < NREG; ++r) {
                            if ((r & (1 << T)) || !(r & (1 << C))) continue;
                            const v2d a0 = amp[r];
                            amp[r] = amp[r | (1 << T)];
                            amp[r | (1 << T)] = a0;
                        }
                    }
                });
            }
        });
    }
}

// ---- ecr(q0, q1) = [ (1 + i Z'_{q0} X_{q1}) / sqrt 2 ] X_{q0} ---------------------------------------
// with c = X_{q0} psi:  out(b) = ( c(b) + i sigma(b_{q0}) c(b ^ q1) ) / sqrt 2,  sigma(0) = +1, sigma(1) = -1
// (the 1 / sqrt 2 goes to the running scale).
template <int NREG>
__device__ __forceinline__ void ecr_any(v2d (&amp)[NREG], int q0, int q1, int lane) {
    u1_gate<NREG, QK_X>(amp, q0, (v2d){0.0, 0.0}, lane);
    // sigma: from the lane's bit (q0 a lane qubit) or the register's bit (wave-uniform)
    const bool lane_q0 = q0 < 6;
    const double sg_lane = (lane_q0 && ((lane >> q0) & 1)) ? -1.0 : 1.0;
    const int tq = lane_q0 ? 0 : q0 - 6;
    auto sigma = [&](int r) __attribute__((always_inline)) { return lane_q0 ? sg_lane : (((r >> tq) & 1) ? -1.0 : 1.0); };
    if (q1 < 6) {
        with_lane_qubit(q1, lane, [&](auto X) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                const v2d c = amp[r];
                const v2d p = X(c);
                const double sg = sigma(r);
                amp[r] = (v2d){fma(-sg, p.y, c.x), fma(sg, p.x, c.y)};
            }
        });
    } else {
        with_reg_qubit<NREG>(q1, [&](auto TT) __attribute__((always_inline)) {
            constexpr int T = decltype(TT)::value;
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                if (r & (1 << T)) continue;
                const v2d c0 = amp[r], c1 = amp[r | (1 << T)];   // q0 != q1: sigma is the same for both
                const double sg = sigma(r);
                amp[r] = (v2d){fma(-sg, c1.y, c0.x), fma(sg, c1.x, c0.y)};
                amp[r | (1 << T)] = (v2d){fma(-sg, c0.y, c1.x), fma(sg, c0.x, c1.y)};
            }
        });
    }
}

// ---- a run of diagonal gates ------------------------------------------------------------------------
// Factors that depend on lane qubits only are gathered in one per-lane factor F, applied once at the end of
// the run; a factor with a register qubit goes to the registers it selects (a compile-time set behind one
// scalar dispatch), the lane part of a mixed term being a select between the factor and 1.
template <int NREG>
__device__ __forceinline__ void diag_run(v2d (&amp)[NREG], const WaveTerm* terms, int first, int count,
                                         const double2* cs, int lane) {
    v2d F = (v2d){1.0, 0.0};
    const v2d one = (v2d){1.0, 0.0};
    WaveTerm t_next = terms[first];
    for (int k = first; k < first + count; ++k) {
        WaveTerm t = t_next;
        if (k + 1 < first + count) t_next = terms[k + 1];    // in flight while this term is applied
        t.type = uni(t.type); t.a = uni(t.a); t.b = uni(t.b); t.expr = uni(t.expr);
        v2d f = (v2d){-1.0, 0.0};                    // TM_CZ
        if (t.expr >= 0) f = angle_cs(cs, t.expr);
        const bool two = t.type == TM_CZ || t.type == TM_ZZP;
        const bool zz = t.type == TM_ZZP;
        const bool la = t.a < 6, lb = !two || t.b < 6;
        const bool bit_a = la && ((lane >> t.a) & 1), bit_b = two && lb && ((lane >> t.b) & 1);
        if (la && lb) {                               // lane qubits only: into F
            v2d g;
            if (t.type == TM_RZ) g = (v2d){f.x, bit_a ? f.y : -f.y};          // (c, -s) for bit 0, (c, +s) for bit 1
            else if (t.type == TM_P) g = bit_a ? f : one;                      // e^{i lam} where the bit is set
            else g = (zz ? (bit_a != bit_b) : (bit_a && bit_b)) ? f : one;     // zz: bits differ; cz: both set
            F = cmul(F, g);
        } else if (!two) {                            // p / rz on a register qubit
            const v2d f_lo = (v2d){f.x, -f.y};
            const bool rz = t.type == TM_RZ;
            with_reg_qubit<NREG>(t.a, [&](auto TT) __attribute__((always_inline)) {
                constexpr int T = decltype(TT)::value;
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    if (r & (1 << T)) amp[r] = cmul(amp[r], f);
                    else if (rz) amp[r] = cmul(amp[r], f_lo);
                }
            });
        } else if (la != lb) {                        // one lane qubit, one register qubit
            const bool bl = la ? bit_a : bit_b;       // the lane's bit
            // register bit 0: zz fires where the lane bit is set, cz never; register bit 1: zz where clear, cz where set
            const v2d f0 = (zz && bl) ? f : one;
            const v2d f1 = (zz ? !bl : bl) ? f : one;
            with_reg_qubit<NREG>(la ? t.b : t.a, [&](auto TT) __attribute__((always_inline)) {
                constexpr int T = decltype(TT)::value;
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    if (r & (1 << T)) amp[r] = cmul(amp[r], f1);
                    else if (zz) amp[r] = cmul(amp[r], f0);
                }
            });
        } else {                                      // two register qubits
            with_reg_qubit<NREG>(t.a, [&](auto AA) __attribute__((always_inline)) {
                constexpr int A = decltype(AA)::value;
                with_reg_qubit<NREG>(t.b, [&](auto BB) __attribute__((always_inline)) {
                    constexpr int B = decltype(BB)::value;
                    if constexpr (A != B) {
#pragma unroll
                        for (int r = 0; r < NREG; ++r) {
                            const bool ra = (r >> A) & 1, rb = (r >> B) & 1;
                            if (ra != rb) { if (zz) amp[r] = cmul(amp[r], f); }
                            else if (ra && !zz) amp[r] = -amp[r];
                        }
                    }
                });
            });
        }
    }
#pragma unroll
    for (int r = 0; r < NREG; ++r) amp[r] = cmul(amp[r], F);
}

__device__ __forceinline__ double2 eval_angle(const WaveExpr ex, const qk_instr* prog, const double* consts,
                                              const double* __restrict__ xrow) {
    double sn, cn;
    sincos_cw(run_postfix(prog, ex.start, ex.len, consts, xrow) * ex.scale, &sn, &cn);
    return make_double2(cn, sn);
}

}  // namespace

namespace {
// product state over the register qubits, one qubit per template step (static register indices)
template <int NREG, int T>
__device__ __forceinline__ void init_reg_qubits(v2d (&amp)[NREG], const WaveInit* init, const double2* cs, const double2* um, int n) {
    if constexpr ((1 << T) < NREG) {
        v2d u0 = (v2d){1.0, 0.0}, u1 = (v2d){0.0, 0.0};
        if (6 + T < n) {
            const int gate = uni(init[6 + T].gate), expr = uni(init[6 + T].expr);
            first_column(gate, (expr >= 0 && gate != QK_UC) ? angle_cs(cs, expr) : (v2d){1.0, 0.0}, u0, u1);
            if (gate == QK_UC) { u0 = (v2d){um[4 * expr].x, um[4 * expr].y}; u1 = (v2d){um[4 * expr + 2].x, um[4 * expr + 2].y}; }
        }
#pragma unroll
        for (int r = 0; r < (1 << T); ++r) {
            amp[r | (1 << T)] = cmul(amp[r], u1);
            amp[r] = cmul(amp[r], u0);
        }
        init_reg_qubits<NREG, T + 1>(amp, init, cs, um, n);
    }
}
}  // namespace

// NREG = 2^(n - 6) register pairs per lane (1 for n <= 6).  __launch_bounds__(64, 4): four waves per SIMD
// (128 VGPRs) -- the amplitudes take 4 * NREG of them.
template <int NREG>
__global__ void __launch_bounds__(64, NREG >= 16 ? 2 : (NREG >= 8 ? 3 : 4)) wave_sim_kernel(WaveArgs args) {
    extern __shared__ double2 cs_lds[];   // [samples of the pass][angles]: (cos, sin); then the staged circuit blob
    const int lane = threadIdx.x;
    char* const lds_prog = (char*)(cs_lds + args.cs_entries);
    const void* staged = nullptr;         // the blob that sits in lds_prog
    for (int64_t u = blockIdx.x; u < args.total_units; u += gridDim.x) {
        // ---- which job (circuit, data, output) this unit belongs to: wave-uniform --------------------
        WaveJob J = args.one;
        if (args.n_jobs > 1) {
            int lo = 0, hi = args.n_jobs;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (args.table[mid].unit_begin <= u) lo = mid; else hi = mid;
            }
            J = args.table[lo];
        }
        __syncthreads();    // the previous unit's readers are done with the LDS tables (one wave: a fence)
        // The gate loop reads the circuit one dependent load after the other; from LDS those cost ~100 cycles
        // instead of an L2 round trip each.  Copied once per change of circuit (16 bytes a lane and step).
        const char* blob = (const char*)J.prog;
        if (J.prog_bytes <= args.stage_bytes) {
            if (J.prog != staged) {
                for (int i = lane; i < (J.prog_bytes >> 4); i += 64) ((uint4*)lds_prog)[i] = ((const uint4*)J.prog)[i];
                staged = J.prog;
                __syncthreads();
            }
            blob = lds_prog;
        }
        const WaveHeader H = *(const WaveHeader*)blob;
        const WaveOp* ops = (const WaveOp*)(blob + H.off_ops);
        const WaveTerm* terms = (const WaveTerm*)(blob + H.off_terms);
        const WaveInit* init = (const WaveInit*)(blob + H.off_init);
        const WaveExpr* exprs = (const WaveExpr*)(blob + H.off_exprs);
        const qk_instr* prog = (const qk_instr*)(blob + H.off_prog);
        const double* consts = (const double*)(blob + H.off_consts);
        const WaveComp* comps = (const WaveComp*)(blob + H.off_comps);
        const WaveFactor* facts = (const WaveFactor*)(blob + H.off_facts);
        const int n = uni(H.n_qubits), ne = uni(H.n_exprs), n_ops = uni(H.n_ops), spw = uni(H.spw), nc = uni(H.n_comps);
        double2* const um_lds = cs_lds + spw * ne;     // [sample of the unit][composite][4]: after the (cos, sin) table
        const int64_t s0 = (u - J.unit_begin) * spw;
        const int cnt = uni((int)(J.N - s0 < spw ? J.N - s0 : spw));

        // ---- the angles of `cnt` samples, one lane per (sample, angle) -----------------------------------
        for (int idx = lane; idx < cnt * ne; idx += 64) {
            const int sub = idx / ne, e = idx - sub * ne;
            cs_lds[idx] = eval_angle(exprs[e], prog, consts, J.X + (s0 + sub) * J.ldx);
        }
        __syncthreads();
        // ---- composite matrices: one lane per (sample, run of single-qubit gates) multiplies the run's factors ----
        if (nc > 0) {
            for (int idx = lane; idx < cnt * nc; idx += 64) {
                const int sub = idx / nc, k = idx - sub * nc;
                const WaveComp cc = comps[k];
                const double2* cs = cs_lds + sub * ne;
                // (named values, not arrays: an array indexed inside this loop would live in scratch memory)
                double2 m0 = make_double2(1.0, 0.0), m1 = make_double2(0.0, 0.0), m2 = m1, m3 = m0;
                for (int j = 0; j < cc.count; ++j) {      // M <- G M, starting from the identity
                    const WaveFactor f = facts[cc.first + j];
                    const double2 a = f.expr >= 0 ? cs[f.expr] : make_double2(1.0, 0.0);
                    double2 g0, g1, g2, g3;
                    factor_matrix(f.gate, a.x, a.y, g0, g1, g2, g3);
                    const double2 n0 = cfma2(g1, m2, cmul2(g0, m0)), n1 = cfma2(g1, m3, cmul2(g0, m1));
                    const double2 n2 = cfma2(g3, m2, cmul2(g2, m0)), n3 = cfma2(g3, m3, cmul2(g2, m1));
                    m0 = n0; m1 = n1; m2 = n2; m3 = n3;
                }
                double2* o = um_lds + (sub * nc + k) * 4;
                o[0] = m0; o[1] = m1; o[2] = m2; o[3] = m3;
            }
            __syncthreads();
        }

        for (int sub = 0; sub < cnt; ++sub) {
            const double2* cs = cs_lds + sub * ne;
            const double2* um = um_lds + sub * nc * 4;
            const int64_t sample = s0 + sub;
            v2d amp[NREG];
            double scale = 1.0;
            // ---- product state of the hoisted first gates -------------------------------------------
            {
                v2d f = (v2d){1.0, 0.0};
                for (int q = 0; q < n && q < 6; ++q) {
                    const int gate = uni(init[q].gate), expr = uni(init[q].expr);
                    v2d u0, u1;
                    first_column(gate, (expr >= 0 && gate != QK_UC) ? angle_cs(cs, expr) : (v2d){1.0, 0.0}, u0, u1);
                    if (gate == QK_UC) { u0 = (v2d){um[4 * expr].x, um[4 * expr].y}; u1 = (v2d){um[4 * expr + 2].x, um[4 * expr + 2].y}; }
                    f = cmul(f, ((lane >> q) & 1) ? u1 : u0);
                }
                amp[0] = f;
                init_reg_qubits<NREG, 0>(amp, init, cs, um, n);
            }
            // ---- gates --------------------------------------------------------------------------------
            WaveOp next_op = ops[0];    // (a circuit without gates still has this slot: the blob is padded)
            for (int k = 0; k < n_ops; ++k) {
                // one definition of every amplitude per gate: left alone, the optimiser carries two copies of the
                // register file around the loop (the swaps of x / cx rename values) and spills
#pragma unroll
                for (int r = 0; r < NREG; ++r) asm volatile("" : "+v"(amp[r]));
                WaveOp op = next_op;
                if (k + 1 < n_ops) next_op = ops[k + 1];    // in flight while this gate runs
                op.kind = uni(op.kind); op.a = uni(op.a); op.b = uni(op.b); op.c = uni(op.c);
                if (op.kind == WV_DIAG) {
                    diag_run<NREG>(amp, terms, op.a, op.b, cs, lane);
                } else if (op.kind == WV_U1 && op.b == QK_UC) {
                    u1_matrix<NREG>(amp, op.a, um + 4 * op.c, lane);
                } else if (op.kind == WV_U1) {
                    if (op.b == QK_H) scale *= 0.70710678118654752440;
                    else if (op.b == QK_SX) scale *= 0.5;
                    u1_any<NREG>(amp, op.a, op.b, op.c >= 0 ? angle_cs(cs, op.c) : (v2d){1.0, 0.0}, lane);
                } else if (op.kind == WV_CX) {
                    cx_any<NREG>(amp, op.a, op.b, lane);
                } else {
                    scale *= 0.70710678118654752440;
                    ecr_any<NREG>(amp, op.a, op.b, lane);
                }
            }
            // ---- store: register r holds amplitudes [64 r, 64 r + 64) ---------------------------------
            const int64_t D = (int64_t)1 << n;
            if (lane < D) {
                if (J.f32) {
                    float2* out = (float2*)J.psi + sample * D + lane;
#pragma unroll
                    for (int r = 0; r < NREG; ++r)
                        if ((int64_t)r * 64 < D) out[r * 64] = make_float2((float)(amp[r].x * scale), (float)(amp[r].y * scale));
                } else {
                    double2* out = (double2*)J.psi + sample * D + lane;
#pragma unroll
                    for (int r = 0; r < NREG; ++r)
                        if ((int64_t)r * 64 < D) out[r * 64] = make_double2(amp[r].x * scale, amp[r].y * scale);
                }
            }
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------
namespace {
template <int NREG>
int launch_wave(const WaveArgs& args, int64_t units, size_t lds, hipStream_t stream) {
    // 64-thread workgroups; enough of them to fill every CU's wave slots, each walking its share of the units
    const int64_t blocks = std::min<int64_t>(units, (int64_t)256 * 16);
    hipLaunchKernelGGL((wave_sim_kernel<NREG>), dim3((unsigned)blocks), dim3(64), lds, stream, args);
    QK_HIP(hipGetLastError());
    return QK_OK;
}
}  // namespace

int wave_nreg_class(int n_qubits) { return n_qubits <= 6 ? 1 : (1 << (n_qubits - 6)); }

size_t wave_lds_bytes(const qk_circuit& c) {
    if (c.wave_blob.empty()) return 0;
    const WaveHeader* H = (const WaveHeader*)c.wave_blob.data();
    return (size_t)16 * std::max(1, H->spw * (H->n_exprs + 4 * H->n_comps));    // (cos, sin) pairs, then the composite matrices
}

// All jobs of one call must belong to one register class (the caller groups them).  cs_bytes: the largest
// (cos, sin) table of a unit over the jobs.
int launch_wave_jobs(const WaveJob* jobs_host, int n_jobs, const WaveJob* table_dev, int nreg, int64_t total_units,
                     size_t cs_bytes, hipStream_t stream) {
    WaveArgs args;
    std::memset(&args, 0, sizeof(args));
    args.one = jobs_host[0];
    args.table = n_jobs > 1 ? table_dev : nullptr;
    args.n_jobs = n_jobs;
    args.total_units = total_units;
    args.cs_entries = (int32_t)((cs_bytes + 15) / 16);
    int32_t biggest = 0;
    for (int j = 0; j < n_jobs; ++j) biggest = std::max(biggest, jobs_host[j].prog_bytes);
    args.stage_bytes = biggest <= kWaveStageMax ? biggest : 0;
    const size_t lds_bytes = (size_t)16 * args.cs_entries + args.stage_bytes;
    if (n_jobs > 1 && !table_dev) return fail(QK_ERR_INVALID, "wave simulator: a batch needs its job table in device memory");
    switch (nreg) {
        case 1: return launch_wave<1>(args, total_units, lds_bytes, stream);
        case 2: return launch_wave<2>(args, total_units, lds_bytes, stream);
        case 4: return launch_wave<4>(args, total_units, lds_bytes, stream);
        case 8: return launch_wave<8>(args, total_units, lds_bytes, stream);
        case 16: return launch_wave<16>(args, total_units, lds_bytes, stream);
        default: return fail(QK_ERR_INVALID, "wave simulator: unsupported register class");
    }
}

}  // namespace qk
