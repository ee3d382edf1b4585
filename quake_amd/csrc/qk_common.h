// Internal declarations shared by the translation units of libquake_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "quake_hip.h"

namespace qk {

// ---- error plumbing -------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

// hipFuncSetAttribute applies to the current device only, so "done once" is kept per device (one
// bit each; two threads that race both set the attribute, which is harmless).
struct DeviceOnce {
    std::atomic<uint64_t> seen{0};
    static uint64_t bit() {
        int dev = 0;
        return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    }
    bool pending() const { const uint64_t b = bit(); return b == 0 || !(seen.load() & b); }
    void done() { seen.fetch_or(bit()); }
};

#define QK_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return ::qk::fail(QK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// ---- simulator program (device-side tables) --------------------------------
// A circuit is planned into passes.  A pass keeps a set of "local" qubits whose
// 2^L amplitudes sit in LDS; the remaining qubits select the chunk (workgroup).
// Inside a pass every qubit has a *virtual* bit position: local qubits take
// 0..L-1 (ascending qubit order), chunk qubits take L..n-1.

enum StageKind : int32_t { ST_DIAG = 0, ST_U1 = 1, ST_CX = 2, ST_ECR = 3, ST_INIT = 4 };
constexpr int kMaxRowBits = 7;  // a diagonal stage splits the local bits into h "column" and <= 7 "row" bits
constexpr int kMaxGroup = 4;  // single-qubit gates on distinct qubits fused into one sweep: 3 (radix 8) in general, 4 Hadamards (radix 16)
enum TermKind : int32_t { TM_P = 0, TM_RZ = 1, TM_CZ = 2, TM_ZZP = 3 };

struct DevTerm {      // one factor of a diagonal stage (device table)
    int32_t type;     // TermKind
    int32_t va, vb;   // virtual bit positions (vb unused for TM_P / TM_RZ)
    int32_t pad;
};

struct DevStage {
    int32_t kind;     // StageKind
    int32_t va, vb;   // CX: va = control, vb = target; ECR: va = q0, vb = q1; INIT: va = the qubit
    int32_t ng;       // U1: number of fused gates (1..kMaxGroup), on distinct local qubits
    int32_t gv[kMaxGroup];  // U1: target bits, ascending
    int32_t gm[kMaxGroup];  // U1: pass-local matrix index of the gate on gv[k]
    // DIAG: pass-local terms [t0, t0+nLo) depend on local bits < h only (plus chunk bits),
    // the next nHi on local bits >= h only, the last nX on one bit of each half; those are sorted
    // by their row bit: the terms with row bit h+a are [xo[a], xo[a+1]) of the nX
    int32_t t0, nLo, nHi, nX;
    int32_t xo[kMaxRowBits + 1];
    // the last nLoD of the nLo and the last nHiD of the nHi involve a chunk bit (>= L): only those have
    // to be re-evaluated from one chunk of a sample to the next
    int32_t nLoD, nHiD;
    // DIAG: the stage has exactly the terms (kinds, bits, angles) of the diagonal stage before it in the pass, whose
    // phase tables are still in LDS: 1 = use them as they are, 2 = that stage was the one fused with the product
    // state (its tables carry the product state's factors: usable when those are constants, see SimArgs::init_h)
    int32_t reuse;
    // INIT: the first gate on a still-|0> qubit (virtual bit va, local or chunk):
    //       amplitude *= entry [bit va][0] of matrix gm[0]
};

struct DevExpr {      // one gate angle evaluated per sample
    int32_t start, len;   // slice of the postfix program
    double scale;         // 0.5 for rx/ry/rz (half angle), 1 for p
    int32_t dst;          // >= 0: pass-local term that takes (cos, sin); < 0: matrix ~dst
    int32_t gate;         // qk_gate of that matrix (rx / ry), 0 for terms
};

struct DevMat {       // a 2x2 matrix: constant for h / x / sx, filled from its angle for rx / ry / rz / p,
    int32_t gate;     // qk_gate, or QK_UC: the product of other matrices of the pass (DevComp)
    int32_t pad;
};

// Runs of single-qubit gates on one qubit with nothing two-qubit in between (the bulk of a GA genome circuit:
// reference genetic_v3.py:326-448 draws one gate per qubit and layer) are applied as ONE 2x2 matrix, the product
// of the run, formed per sample from the factors' matrices.  QK_UC marks such a matrix; it is not a gate code
// of the public header.
constexpr int32_t QK_UC = 64;
struct DevComp {
    int32_t dst;           // pass-local matrix that takes the product
    int32_t first, count;  // its factors: pass-local matrix indices facts[first .. first + count), in order of application
    int32_t pad;
};

constexpr int kMaxRuns = 16;
constexpr int kMaxLocalQubits = 13;          // 2^13 * 16 B = 128 KiB of LDS
constexpr int kMaxTermPerPass = 256;         // 8 KiB of resolved diagonal terms in LDS
constexpr int kMaxMatPerPass = 160;          // 10 KiB of 2x2 complex matrices

struct Pass {
    int L = 0;                 // local qubits
    int lgT = 6;               // log2(threads)
    int h = 0;                 // diagonal stages: local bits < h index the low phase table, the rest the high one
    int maxg = 2;              // largest fused single-qubit group
    bool narrow = false;       // 512 threads on the 128-register kernel (four waves per SIMD with two workgroups per CU)
    int maxh = 0;              // largest group of Hadamards in one sweep (4 = radix 16, one item per thread), 0 = as maxg
    bool first = false;        // starts from |0...0> (no load)
    // Qubits no non-diagonal gate has touched yet are still |0>: every amplitude with such a bit
    // set is exactly zero.  zero_in: those bits (virtual positions) when the pass starts -- not
    // loaded; zero_out: those still untouched after a pass that is not the last -- not stored
    // (the next pass knows they are zero).
    uint32_t zero_in = 0, zero_out = 0;
    // First pass only: the leading n_init stages are ST_INIT, the hoisted first gates (see the
    // planner).  The pass starts from the product state they make instead of |0...0>.
    int n_init = 0;
    std::vector<int> phys;     // phys[v] = physical qubit at virtual position v (size n)
    int stage_off = 0, n_stage = 0;
    int expr_off = 0, n_expr = 0;
    int term_off = 0, n_term = 0;
    int mat_off = 0, n_mat = 0;
    int comp_off = 0, n_comp = 0;   // composite matrices (DevComp) of the pass; their factor lists index c.facts from fact_off
    int fact_off = 0;
    // bit-deposit runs: physical = OR_r ((virtual & mask[r]) << shift[r])
    int nruns = 0;
    uint32_t run_mask[kMaxRuns] = {0};
    int32_t run_shift[kMaxRuns] = {0};
};

}  // namespace qk

namespace qk {
// One job of the wave-resident simulator (qk_wave.hip): a circuit blob (qk_wave.h), its feature rows and
// its output.  A unit of work is `spw` consecutive samples of one job; unit_begin = units of the jobs before.
struct WaveJob {
    const void* prog;     // device: WaveHeader + tables
    const double* X;      // device: N x n_params, row stride ldx
    int64_t ldx;
    int64_t N;
    void* psi;            // device: N x 2^n complex128 (f32 == 0) or complex64
    int64_t unit_begin;
    int32_t f32;
    int32_t prog_bytes;   // size of the blob (a multiple of 16)
};
}  // namespace qk

namespace qk {
struct DevTables {
    DevStage* stages = nullptr;
    DevTerm* terms = nullptr;
    DevExpr* exprs = nullptr;
    DevMat* mats = nullptr;
    DevComp* comps = nullptr;
    int32_t* facts = nullptr;
    qk_instr* prog = nullptr;
    double* consts = nullptr;
    void* wave = nullptr;      // the wave-resident program, null when the circuit has none
    void* reg = nullptr;       // the register-resident program, null when the circuit has none
};
}  // namespace qk

struct qk_circuit {
    int n_qubits = 0;
    int n_params = 0;
    int n_gates = 0;                 // gates that touch the state (after dropping id)
    std::string plan_summary;        // what the planner made of them (development hook qk_circuit_plan_summary)
    std::vector<qk::Pass> passes;
    // host copies (uploaded lazily per device)
    std::vector<qk::DevStage> stages;
    std::vector<qk::DevTerm> terms;
    std::vector<qk::DevExpr> exprs;
    std::vector<qk::DevMat> mats;
    std::vector<qk::DevComp> comps;
    std::vector<int32_t> facts;
    std::vector<qk_instr> prog;
    std::vector<double> consts;
    // wave-resident program (n_qubits <= 10): one position-independent blob, empty when the circuit does not qualify
    std::vector<uint64_t> wave_blob;
    int wave_spw = 1;                // samples per unit of work
    // register-resident program (10..13 qubits, one pass: qk_reg.h), empty when the circuit does not qualify
    std::vector<uint64_t> reg_blob;
    size_t reg_lds_bytes = 0;
    // device copies: one set of tables per device, uploaded on first use there and kept until the circuit is
    // destroyed (under `mu`: ctypes releases the GIL, so two host threads may reach the first call together; a
    // caller works from a by-value copy of its device's pointers, so another thread on another device cannot
    // pull them away)
    std::mutex mu;
    std::map<int, qk::DevTables> tables;
    // The same circuit planned without 512-thread passes (Pass::narrow), for the fused simulate + RDM mode, whose
    // reduction needs the 256-register kernel (radix-8 sweeps on 256 threads); null when the two plans agree.
    qk_circuit* rdm_plan = nullptr;
    ~qk_circuit() { delete rdm_plan; }
};

namespace qk {
// qk_circuit.cpp
int plan_circuit(qk_circuit& c, const qk_op* ops, int n_ops, const int32_t* expr_off, int n_expr, bool allow_narrow = true);
// qk_stats.hip: noise + statistics of many small symmetric matrices in one launch (dense, ldk = N)
struct EpiJob {
    double* K;
    int64_t N;
    const int32_t* labels;   // may be null when no statistics are asked for
    double* stats;           // QK_STATS_LEN doubles
    uint64_t seed;
};
int launch_batch_epilogue(const EpiJob* table_dev, int count, int noise_mode, double noise_param, bool want_stats,
                          hipStream_t stream);
// noise on rows [i0, i1) of a matrix with `cols` columns (symmetric: entries right of the diagonal, mirrored)
int launch_noise(double* K, int64_t i0, int64_t i1, int64_t cols, int64_t ldk, bool symmetric, int mode, double param,
                 uint64_t seed, hipStream_t stream);
// qk_wave.hip
int wave_nreg_class(int n_qubits);
size_t wave_lds_bytes(const qk_circuit& c);   // the (cos, sin) table of one unit of work
int launch_wave_jobs(const WaveJob* jobs_host, int n_jobs, const WaveJob* table_dev, int nreg, int64_t total_units,
                     size_t cs_bytes, hipStream_t stream);
// qk_reg.hip: the register-resident simulator (cs: the (cos, sin) table of the blob's angle programs, [expr][sample])
//             tabs: the phase tables of its diagonal stages and its matrices, per sample group (launch_reg_side,
//             reg_side_bytes of device memory)
size_t reg_side_bytes(const qk_circuit& c, int64_t N);
int launch_reg_side(const qk_circuit& c, const void* blob_dev, const double2* cs, int64_t N, void* tabs, hipStream_t stream);
int launch_reg_sim(const qk_circuit& c, const void* blob_dev, const double2* cs, const void* tabs, int64_t N, void* psi, void* psi32,
                   hipStream_t stream, double* rdm = nullptr);      // rdm: the reduced density matrices instead of Psi (unpacked plans)
bool reg_rdm_mode_ok(const qk_circuit& c);      // the plan can run in that mode (one sample per workgroup, LDS left for the reduction)
// qk_sim.hip
int upload_circuit(qk_circuit& c, DevTables& out);   // the tables of the current device (uploaded on first use)
void free_circuit_device(qk_circuit& c);
}  // namespace qk
