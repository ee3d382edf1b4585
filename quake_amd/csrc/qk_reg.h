// Program format of the register-resident simulator (qk_reg.hip): circuits of 4 to 13 qubits that run as ONE pass
// with every qubit local.  The 2^L amplitudes of a sample (below 10 qubits: of 2^(L - nq) samples side by side) live in
// the registers of one workgroup, 16 per thread, across all stages; LDS only carries them from one register layout to
// the next.  What depends on the sample -- gate angles, phase tables of the diagonal stages, 2x2 matrices -- is
// computed for the whole batch by pre-pass kernels (one thread per sample) from the records below and fetched by the
// simulator with LDS-DMA.
//
// A layout is a choice of four "register" qubits: a thread's 16 amplitudes differ in exactly those bits, the other
// L - 4 bits are the thread's index (ascending).  Every non-diagonal gate is applied where its target is a register
// qubit -- a butterfly between two registers of the lane, no data movement, no index arithmetic -- and diagonal
// stages are applied in whatever layout the state happens to be in.  The planner (qk_circuit.cpp) cuts the circuit
// into segments, one layout each; between two segments the workgroup transposes through LDS.
#pragma once
#include <cstdint>

namespace qk {

constexpr int kRegMinQubits = 10;     // 64 threads x 16 amplitudes
constexpr int kRegPackMinQubits = 4;  // narrower circuits: 2^(L - n) samples side by side in a virtual register of L index bits
constexpr int kRegPackMaxWidth = 12;  // ... of at most 12 (256 threads)
constexpr int kRegMaxQubits = 13;     // 512 threads x 16 amplitudes = 128 KiB of LDS for the transposes
constexpr int kRegMaxTabs = 96;       // phase tables of 16 entries (256 bytes each) per sample
constexpr int kRegMaxTerms = 256;
constexpr int kRegMaxMats = 255;      // (a byte per register position in RegOp::b)

enum RegOpKind : int32_t {
    RG_INIT = 0,   // the product state of the hoisted first gates (tab: three or four tables of first-column products)
    RG_DIAG = 1,   // a diagonal stage: tables A (register bits), Blo / Bhi (thread bits), cross terms evaluated per thread
    RG_U1 = 2,     // 2x2 matrices on the register positions in mask a (distinct positions commute): matrix index of
                   // position p in byte p of b, butterfly form in bits 2p, 2p + 1 of flags
    RG_CX = 3,     // control register position a, target register position b
    RG_CXT = 4,    // control thread bit a (index into the thread's bits), target register position b
    RG_ECR = 5,    // ecr(q0, q1) on register positions a, b
};

struct RegOp {        // 32 bytes
    int32_t kind;
    int32_t a, b;
    int32_t tab;      // INIT: first table: A, Blo, Bhi (two slots when the thread has nine bits; none when it has at most four).
                      // DIAG, counted from RegHeader::n_itab: bits 0..15 table A, a = Blo, b = Bhi (first slot), bits 16..31 the
                      // first ratio table (flags bit 9) -- equal tables of different stages exist once
    int32_t t0;       // DIAG: first of the terms evaluated per thread (RegTerm): nTX between two thread bits, then nRT between a
    int32_t nTX, nRT; //       register bit and a thread bit (bits 8.. of RegTerm::type = that register position)
    int32_t flags;    // DIAG: bits 0..3 = register positions that have RT terms, bit 8 = the state is still the identity; INIT: bit 0 = every first gate is a Hadamard;
                      // U1: butterfly form (0 real matrix, 1 rx-shaped, 2 general, 3 hadamard)
};

struct RegSeg {       // 64 bytes: one layout and the ops applied in it
    int32_t g[4];     // the register qubits, ascending
    int32_t op0, op1;
    // the thread's part of the amplitude index: its bits dealt to the non-register qubits, ascending, as at most five
    // runs of consecutive bits:  wt = OR_r ((tid & run_mask[r]) << run_shift[r])
    int32_t nrun;
    uint32_t run[5];  // run_mask | run_shift << 16
    int32_t pad[4];
};

struct RegTab {       // 32 bytes: one table of 16 entries; entry x covers the index fixed | deposit(x over bits)
    int32_t bits[4];
    int32_t nbits;
    uint32_t fixed;
    int32_t t0, nt;   // tables of the product state (built inside the simulator, from the sample's matrices): nt < 0, product
                      // of the first columns of the qubits in `bits` and of the qubits in the mask t0 (whose values `fixed`
                      // holds).  Tables of a diagonal stage (built by reg_table_kernel before the simulator runs and
                      // fetched by LDS-DMA): product over the RegTabTerm records [t0, t0 + nt)
};

struct RegTabTerm {   // 8 bytes: one diagonal term's part in one table
    int32_t expr;     // angle program whose (cos, sin) is the factor; < 0: the constant -1 (cz)
    uint32_t masks;   // bit x: entry x takes the factor; bit 16 + x: ... its conjugate (rz with the bit clear)
};

struct RegTerm {      // 16 bytes: a diagonal term the simulator evaluates per thread (staged in LDS as it is)
    int32_t type;     // TermKind | rpos << 8 (RT terms: register position of the term's register bit)
    int32_t va, vb;
    int32_t expr;     // its (cos, sin) among the first n_expr_k angle programs; < 0: the constant -1 (cz)
};

struct RegMat {       // a 2x2 matrix the simulator reads: the product of the factors [first, first + count), first one applied first
    int32_t first, count;
};
struct RegFact {
    int32_t gate;     // QK_H, QK_X, QK_SX, QK_RX, QK_RY, QK_RZ, QK_P
    int32_t expr;     // angle program of a parametric gate, -1 otherwise
};

struct RegHeader {    // at offset 0 of the blob; off_* are byte offsets from its start (16-byte aligned sections)
    int32_t L, n_seg, n_op, n_tab, n_term, n_expr, n_mat;      // n_mat RegMat records, n_fact RegFact records; n_term RegTerm records
    int32_t n_comp;   // (unused, 0)
    int32_t n_fact;
    int32_t init_h;   // bit 0: every first gate is a Hadamard; bit 1: ONE buffer of stage tables and matrices in LDS instead of two
    int32_t off_seg, off_op, off_tab, off_term, off_expr, off_mat, off_comp, off_fact, off_init, bytes;
    int32_t rb, nq;   // register qubits per layout: 4 (16 amplitudes per thread) or 3 (8 per thread, twice the threads);
                      // nq = qubits of the circuit: below L when 2^(L - nq) samples share a workgroup (index bits nq .. L - 1
                      // = the sample; those plans have rb = 4, no table for high thread qubits unless nq > 8)
    double init_c;    // 2^(-nq/2): the product state of an all-Hadamard first layer
    // tables [0, n_itab) belong to the product state (built in the simulator unless init_h), [n_itab, n_tab) to the diagonal
    // stages (RegOp::tab of a DIAG op counts from n_itab); n_tt RegTabTerm records at off_tt
    int32_t n_itab, n_tt, off_tt;
    int32_t n_expr_k; // the first n_expr_k angle programs are the ones the simulator reads itself (per-thread terms); the rest
                      // feed the pre-pass kernels only (table terms, factors of matrices)
};

}  // namespace qk
