// Internal declarations shared by the translation units of libquake_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "quake_hip.h"

namespace qk {

// ---- error plumbing -------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

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

enum StageKind : int32_t { ST_DIAG = 0, ST_U1 = 1, ST_CX = 2, ST_ECR = 3 };
enum TermKind : int32_t { TM_P = 0, TM_RZ = 1, TM_CZ = 2, TM_ZZP = 3 };

struct DevTerm {      // one factor of a diagonal stage
    int32_t type;     // TermKind
    int32_t va, vb;   // virtual bit positions (vb unused for TM_P / TM_RZ)
    int32_t cs;       // pass-local index of the (cos, sin) pair, -1 for TM_CZ
};

struct DevStage {
    int32_t kind;     // StageKind
    int32_t va, vb;   // U1: va = target; CX: va = control, vb = target; ECR: va = q0, vb = q1
    int32_t idx;      // U1: pass-local matrix index
    int32_t t0;       // DIAG: first term
    int32_t nA;       // DIAG: terms constant per thread (bits < lgT or chunk bits)
    int32_t nB;       // DIAG: terms evaluated per amplitude
    int32_t pad;
};

struct DevExpr {      // one gate angle evaluated per sample
    int32_t start, len;   // slice of the postfix program
    double scale;         // 0.5 for rx/ry/rz (half angle), 1 for p
};

struct DevMat {       // how to fill a 2x2 matrix from the gate kind and its (cos, sin)
    int32_t gate;     // qk_gate (H, X, SX, RX, RY)
    int32_t cs;       // pass-local (cos, sin) index or -1
};

constexpr int kMaxRuns = 16;
constexpr int kMaxLocalQubits = 13;          // 2^13 * 16 B = 128 KiB of LDS
constexpr int kMaxExprPerPass = 640;         // 10 KiB of (cos, sin)
constexpr int kMaxMatPerPass = 288;          // 18 KiB of 2x2 complex matrices

struct Pass {
    int L = 0;                 // local qubits
    int lgT = 6;               // log2(threads)
    bool first = false;        // starts from |0...0> (no load)
    std::vector<int> phys;     // phys[v] = physical qubit at virtual position v (size n)
    int stage_off = 0, n_stage = 0;
    int expr_off = 0, n_expr = 0;
    int mat_off = 0, n_mat = 0;
    // bit-deposit runs: physical = OR_r ((virtual & mask[r]) << shift[r])
    int nruns = 0;
    uint32_t run_mask[kMaxRuns] = {0};
    int32_t run_shift[kMaxRuns] = {0};
};

}  // namespace qk

struct qk_circuit {
    int n_qubits = 0;
    int n_params = 0;
    int n_gates = 0;                 // gates that touch the state (after dropping id)
    std::vector<qk::Pass> passes;
    // host copies (uploaded lazily per device)
    std::vector<qk::DevStage> stages;
    std::vector<qk::DevTerm> terms;
    std::vector<qk::DevExpr> exprs;
    std::vector<qk::DevMat> mats;
    std::vector<qk_instr> prog;
    std::vector<double> consts;
    // device copies
    int device = -1;
    qk::DevStage* d_stages = nullptr;
    qk::DevTerm* d_terms = nullptr;
    qk::DevExpr* d_exprs = nullptr;
    qk::DevMat* d_mats = nullptr;
    qk_instr* d_prog = nullptr;
    double* d_consts = nullptr;
};

namespace qk {
// qk_circuit.cpp
int plan_circuit(qk_circuit& c, const qk_op* ops, int n_ops, const int32_t* expr_off, int n_expr);
// qk_sim.hip
int upload_circuit(qk_circuit& c);
void free_circuit_device(qk_circuit& c);
}  // namespace qk
