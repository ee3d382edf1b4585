// Per-qubit reduced density matrices of a statevector that sits in LDS, shared by the RDM kernel
// (qk_pqk.hip) and the simulator's fused mode (qk_sim.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace qk {

constexpr int RDM_MAXQ = 13;
constexpr int RDM_GROUPS = (RDM_MAXQ + 3) / 4;   // qubits are handled four at a time
constexpr int RDM_SLOTS = 64;                     // 4 values x 13 qubits, padded to a multiple of 16

// One group of G <= 4 neighbouring qubits [q0, q0 + G): a thread reads the 2^G amplitudes that differ
// in those bits once and serves all G qubits from registers (2^G LDS reads for G * 2^(G-1) pairs).
// For q0 == 0 consecutive lanes would walk LDS with a 2^G-amplitude stride (a 16-way bank conflict
// at G = 4), so lane l reads register s from amplitude s ^ (l & (2^G - 1)): registers s and s ^ m
// still hold a pair that differs in bit m, only WHICH of the two has the bit set depends on the
// lane -- fixed once per thread after the loop (swap rho00 / rho11, conjugate rho01).
template <int G>
__device__ __forceinline__ void rdm1_group(const double2* st, uint32_t D, int q0, int tid, int T, double* acc /* [G][4] */) {
    constexpr int R = 1 << G;
    const uint32_t lowmask = (1u << q0) - 1u;
    const uint32_t X = q0 == 0 ? ((uint32_t)tid & (R - 1)) : 0u;
    double A[G], B[G], xr[G], xi[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { A[g] = 0; B[g] = 0; xr[g] = 0; xi[g] = 0; }
    for (uint32_t i = tid; i < (D >> G); i += T) {
        const uint32_t base = ((i & ~lowmask) << G) | (i & lowmask);
        double2 v[R];
        double p[R];
#pragma unroll
        for (int s = 0; s < R; ++s) {
            v[s] = st[base | (((uint32_t)s ^ X) << q0)];
            p[s] = fma(v[s].x, v[s].x, v[s].y * v[s].y);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int s = 0; s < R; ++s) {
                if (s & (1 << g)) {
                    A[g] += p[s];
                } else {
                    B[g] += p[s];
                    const double2 a0 = v[s], a1 = v[s | (1 << g)];
                    xr[g] = fma(a0.x, a1.x, fma(a0.y, a1.y, xr[g]));      // a0 * conj(a1)
                    xi[g] = fma(a0.y, a1.x, fma(-a0.x, a1.y, xi[g]));
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const bool flip = (X >> g) & 1u;
        acc[4 * g + 0] = flip ? A[g] : B[g];     // rho00
        acc[4 * g + 1] = flip ? B[g] : A[g];     // rho11
        acc[4 * g + 2] = xr[g];
        acc[4 * g + 3] = flip ? -xi[g] : xi[g];
    }
}

__device__ __forceinline__ double shfl_xor_f64(double x, int mask) { return __shfl_xor(x, mask, 64); }

// st: 2^n amplitudes in LDS (all threads' writes visible), n <= RDM_MAXQ.
// red: LDS scratch of RDM_SLOTS * (T / 64) doubles.  out: n x 4 doubles (rho00, rho11, Re rho01, Im rho01).
// Ends with a barrier: st and red may be reused on return.
__device__ __forceinline__ void rdm1_from_lds(const double2* st, int n, double* red, double* __restrict__ out, int tid,
                                              int T) {
    const uint32_t D = 1u << n;
    const int lane = tid & 63, wave = tid >> 6, waves = T >> 6;
    // Per group of four qubits: 16 per-thread values, reduced over the 64 lanes by recursive halving --
    // at each of four steps a lane keeps one half of its values and hands the other half to its
    // partner (8 + 4 + 2 + 1 shuffles instead of 16 * 6); the last value is all-reduced over 4 lanes.
#pragma unroll
    for (int grp = 0; grp < RDM_GROUPS; ++grp) {
        const int q0 = 4 * grp, left = n - q0;       // uniform branches: n is a kernel argument
        if (left <= 0) break;
        double val[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) val[k] = 0.0;
        if (left >= 4) rdm1_group<4>(st, D, q0, tid, T, val);
        else if (left == 3) rdm1_group<3>(st, D, q0, tid, T, val);
        else if (left == 2) rdm1_group<2>(st, D, q0, tid, T, val);
        else rdm1_group<1>(st, D, q0, tid, T, val);
#pragma unroll
        for (int step = 0, cnt = 8, off = 32; step < 4; ++step, cnt >>= 1, off >>= 1) {
            const bool hi = lane & off;
#pragma unroll
            for (int k = 0; k < cnt; ++k) {
                double lo_v = val[k], hi_v = val[k + cnt];
                // keep the two selects as selects: folded into val[hi ? .. : ..] they become a dynamic
                // index into the register array, i.e. a 16-way compare/select chain per element
                asm volatile("" : "+v"(lo_v), "+v"(hi_v));
                const double send = hi ? lo_v : hi_v;
                const double keep = hi ? hi_v : lo_v;
                val[k] = keep + shfl_xor_f64(send, off);
            }
        }
        val[0] += shfl_xor_f64(val[0], 2);
        val[0] += shfl_xor_f64(val[0], 1);
        if ((lane & 3) == 0) {
            const int slot = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
            red[wave * RDM_SLOTS + 16 * grp + slot] = val[0];
        }
    }
    __syncthreads();
    for (int e = tid; e < 4 * n; e += T) {
        double x = 0;
        for (int w = 0; w < waves; ++w) x += red[w * RDM_SLOTS + e];
        out[e] = x;
    }
    __syncthreads();
}

}  // namespace qk
