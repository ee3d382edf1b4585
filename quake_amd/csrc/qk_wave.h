// Program format of the wave-resident simulator (qk_wave.hip): circuits of up to 10 qubits whose whole
// statevector lives in the registers of ONE wavefront.  Built by the host planner (qk_circuit.cpp) as one
// position-independent blob, so that a batch of circuits travels to the device in a single copy.
#pragma once
#include <cstdint>

namespace qk {

constexpr int kWaveMaxQubits = 10;      // 64 lanes x 16 amplitudes
constexpr int kWaveMaxExprs = 2048;     // (cos, sin) pairs of one sample in LDS: 32 KiB
constexpr int kWaveStageMax = 24 * 1024;   // largest blob staged in LDS
constexpr int kWaveLdsLimit = 64 * 1024;   // dynamic LDS of a launch (the default limit; never raised): tables + staged blob

enum WaveOpKind : int32_t { WV_U1 = 0, WV_CX = 1, WV_ECR = 2, WV_DIAG = 3 };

struct WaveOp {       // 16 bytes
    int32_t kind;     // WaveOpKind
    int32_t a;        // U1: qubit            CX: control   ECR: q0   DIAG: first term
    int32_t b;        // U1: qk_gate or QK_UC CX: target    ECR: q1   DIAG: number of terms
    int32_t c;        // U1: angle index of rx / ry, composite index of QK_UC, else -1
};

struct WaveTerm {     // one factor of a diagonal run
    int32_t type;     // TermKind
    int32_t a, b;     // qubits (b unused for TM_P / TM_RZ)
    int32_t expr;     // angle index, -1 for TM_CZ
};

struct WaveInit {     // first gate on a qubit, folded into the initial product state
    int32_t gate;     // qk_gate, QK_UC, or -1: the qubit starts in |0>
    int32_t expr;     // angle index of rx / ry, composite index of QK_UC, else -1
};

struct WaveComp {     // a run of single-qubit gates on one qubit, applied as the product of its factors
    int32_t first, count;   // factors [first, first + count), in order of application
};

struct WaveFactor {
    int32_t gate;     // qk_gate: h x sx rx ry rz p
    int32_t expr;     // angle index of a parametric factor, else -1
};

struct WaveExpr {
    int32_t start, len;   // slice of the postfix program
    double scale;         // 0.5 for rx / ry / rz (half angle), 1 for p and the ZZ phase
};

struct WaveHeader {   // 80 bytes, at offset 0 of the blob; off_* are byte offsets from the blob's start
    int32_t n_qubits, n_params, n_ops, n_terms, n_exprs, n_prog, n_consts;
    int32_t spw;      // samples whose angles one pass of the wave evaluates together: max(1, 64 / n_exprs)
    int32_t off_ops, off_terms, off_init, off_exprs, off_prog, off_consts;
    int32_t bytes, n_comps;
    int32_t off_comps, off_facts, pad0, pad1;
};

}  // namespace qk
