// Shared device math: sine and cosine of a gate angle.
#pragma once
#include <cmath>
#ifndef QK_HD
#ifdef __HIPCC__
#define QK_HD __host__ __device__ __forceinline__
#else
#define QK_HD inline
#endif
#endif

namespace qk {

// sin and cos of x in one pass: Cody-Waite reduction by pi/2 in three 33-bit pieces (exact products for
// |k| < 2^20), then the fdlibm kernels on |r| <= pi/4.  Error below 1 ulp of the larger of the two
// results for |x| < 1e6 -- every angle a feature map produces; beyond that the library routine takes over.
QK_HD void sincos_cw(double x, double* sn, double* cs) {
    if (!(fabs(x) < 1.0e6)) {
        *sn = sin(x);
        *cs = cos(x);
        return;
    }
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(k, -1.57079632673412561417e+00, x);
    r = fma(k, -6.07710050630396597660e-11, r);
    const double w = k * 2.02226624871116645580e-21;
    const double r2 = r - w;
    // the bits lost in r - w, and the next piece of pi/2
    const double lo = fma(k, 8.47842766036889956997e-32, w - (r - r2));
    const double z = r2 * r2;
    // sin(r2 - lo)
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    const double v = z * r2;
    // fdlibm: x - ((z*(half*y - v*r) - y) - v*S1)
    const double s_r = r2 - ((z * (0.5 * lo - v * ps) + lo) - v * -1.66666666666666324348e-01);
    // cos(r2 - lo)
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double wc = 1.0 - hz;
    const double c_r = wc + (((1.0 - wc) - hz) + (z * (z * pc) + r2 * lo));
    const int q = (int)k & 3;
    const double ss = (q & 1) ? c_r : s_r;
    const double cc = (q & 1) ? s_r : c_r;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

// The postfix machine of the gate angles (programs validated on the host, qk_circuit_create: at most 8 stack
// slots), shared by both simulators.  The stack is eight named registers that shift on push and pop -- an array
// indexed by the stack pointer lives in scratch memory, and every push and pop becomes a round trip to it
// (the LDS simulator spent 11 us per sample there at 12 qubits: tools/sim_trace.py).
struct AngleStack {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    __device__ __forceinline__ void push(double v) { s7 = s6; s6 = s5; s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v; }
    __device__ __forceinline__ void drop() { s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7; }   // pop the top
};

// value of one angle program on the feature row xrow.  The lanes of a wave run different programs, so a step costs
// the sum of the distinct operations its lanes are at (about 3 us per sample for the 46 angles of a 12-qubit ZZ
// map r = 2: tools/sim_trace.py); fetching instructions and operands ahead of their use bought nothing -- the
// loop is bound by the instructions of the divergent switch, not by its reads.
template <class ProgPtr, class ConstPtr>
__device__ __forceinline__ double run_postfix(ProgPtr prog, int start, int len, ConstPtr consts, const double* xrow) {
    AngleStack st;
    for (int k = 0; k < len; ++k) {
        const qk_instr in = prog[start + k];
        const double a = st.s0, b = st.s1;    // top, second
        switch (in.op) {
            case QK_EX_CONST: st.push(consts[in.arg]); break;
            case QK_EX_PARAM: st.push(xrow[in.arg]); break;
            case QK_EX_ADD: st.drop(); st.s0 = b + a; break;
            case QK_EX_SUB: st.drop(); st.s0 = b - a; break;
            case QK_EX_MUL: st.drop(); st.s0 = b * a; break;
            case QK_EX_DIV: st.drop(); st.s0 = b / a; break;
            case QK_EX_NEG: st.s0 = -a; break;
            case QK_EX_ASIN: st.s0 = asin(a); break;
            case QK_EX_ACOS: st.s0 = acos(a); break;
            case QK_EX_SIN: st.s0 = sin(a); break;
            case QK_EX_COS: st.s0 = cos(a); break;
            default: break;
        }
    }
    return st.s0;
}

}  // namespace qk
