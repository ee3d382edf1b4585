/*
 * CPU ORACLE (C twin) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * A scalar, single-threaded restatement of what the reference's kernel object does
 * per call, written independently of the numpy oracle so the two can check each
 * other and so the CPU baseline is timed on compiled code rather than on Python:
 *
 *   qko_simulate  = [Statevector(fmap.assign_parameters(x)).data for x in x_vec]
 *                   gate by gate from |0..0>, qubit 0 = LSB, complex128
 *                   (qiskit conventions; call sites reference
 *                   qsvm/train.py:92,152,154, genetic_v3.py:618-619)
 *   qko_fidelity  = FidelityStatevectorKernel._compute_fidelity for every pair:
 *                   one |vdot(psi_i, psi_j)|^2 per entry; the
 *                   symmetric case walks j >= i and mirrors, diagonal left at 1.0
 *   qko_rdm1/qko_pqk = the pair loop of projected_quantum_kernel
 *                   (reference genetic_v3.py:793-805) with the per-sample 1-RDMs
 *                   hoisted out of the pair loop
 *
 * Gate angles arrive numerically (one double per sample and gate): the symbolic
 * binding is the numpy oracle's job.  Parity pinning: see oracle/qk_oracle.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { G_ID = 0, G_H, G_X, G_SX, G_RX, G_RY, G_RZ, G_P, G_CX, G_CZ, G_ECR };

typedef struct { double re, im; } cplx;

static inline cplx cmul(cplx a, cplx b) { cplx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
static inline cplx cadd(cplx a, cplx b) { cplx r = {a.re + b.re, a.im + b.im}; return r; }

static void apply_1q(cplx* s, int n, int q, const cplx m[4]) {
    const int64_t D = (int64_t)1 << n, st = (int64_t)1 << q;
    for (int64_t base = 0; base < D; base += 2 * st)
        for (int64_t k = 0; k < st; ++k) {
            cplx a0 = s[base + k], a1 = s[base + k + st];
            s[base + k] = cadd(cmul(m[0], a0), cmul(m[1], a1));
            s[base + k + st] = cadd(cmul(m[2], a0), cmul(m[3], a1));
        }
}

/* gates: n_ops x 3 (code, q0, q1); angles: N x n_ops; psi: N x 2^n */
int qko_simulate(int n, int n_ops, const int32_t* gates, const double* angles, int64_t N, cplx* psi) {
    const int64_t D = (int64_t)1 << n;
    const double r = 0.70710678118654752440;
    for (int64_t smp = 0; smp < N; ++smp) {
        cplx* s = psi + smp * D;
        memset(s, 0, (size_t)D * sizeof(cplx));
        s[0].re = 1.0;
        for (int g = 0; g < n_ops; ++g) {
            const int code = gates[3 * g], q0 = gates[3 * g + 1], q1 = gates[3 * g + 2];
            const double th = angles[smp * n_ops + g];
            cplx m[4];
            memset(m, 0, sizeof(m));
            switch (code) {
                case G_ID: continue;
                case G_H: m[0].re = r; m[1].re = r; m[2].re = r; m[3].re = -r; break;
                case G_X: m[1].re = 1; m[2].re = 1; break;
                case G_SX:
                    m[0].re = 0.5; m[0].im = 0.5; m[1].re = 0.5; m[1].im = -0.5;
                    m[2].re = 0.5; m[2].im = -0.5; m[3].re = 0.5; m[3].im = 0.5; break;
                case G_RX: m[0].re = cos(th / 2); m[1].im = -sin(th / 2); m[2].im = -sin(th / 2); m[3].re = cos(th / 2); break;
                case G_RY: m[0].re = cos(th / 2); m[1].re = -sin(th / 2); m[2].re = sin(th / 2); m[3].re = cos(th / 2); break;
                case G_RZ: m[0].re = cos(th / 2); m[0].im = -sin(th / 2); m[3].re = cos(th / 2); m[3].im = sin(th / 2); break;
                case G_P: m[0].re = 1; m[3].re = cos(th); m[3].im = sin(th); break;
                case G_CX: case G_CZ: case G_ECR: {
                    const int64_t b0 = (int64_t)1 << q0, b1 = (int64_t)1 << q1;
                    for (int64_t i = 0; i < D; ++i) {
                        if (i & (b0 | b1)) continue; /* i has both bits clear */
                        cplx a0 = s[i], a1 = s[i | b0], a2 = s[i | b1], a3 = s[i | b0 | b1];
                        if (code == G_CX) { /* control q0, target q1: swap |c=1,t=0> and |c=1,t=1> */
                            s[i | b0] = a3; s[i | b0 | b1] = a1;
                        } else if (code == G_CZ) {
                            s[i | b0 | b1].re = -a3.re; s[i | b0 | b1].im = -a3.im;
                        } else { /* 1/sqrt2 [[0,1,0,i],[1,0,-i,0],[0,i,0,1],[-i,0,1,0]], index 2*bit(q1)+bit(q0) */
                            s[i].re = r * (a1.re - a3.im);            s[i].im = r * (a1.im + a3.re);
                            s[i | b0].re = r * (a0.re + a2.im);       s[i | b0].im = r * (a0.im - a2.re);
                            s[i | b1].re = r * (a3.re - a1.im);       s[i | b1].im = r * (a3.im + a1.re);
                            s[i | b0 | b1].re = r * (a2.re + a0.im);  s[i | b0 | b1].im = r * (a2.im - a0.re);
                        }
                    }
                    continue;
                }
                default: return -1;
            }
            apply_1q(s, n, q0, m);
        }
    }
    return 0;
}

static inline double fid(const cplx* a, const cplx* b, int64_t D) {
    double re = 0, im = 0; /* conj(a) . b */
    for (int64_t k = 0; k < D; ++k) {
        re += a[k].re * b[k].re + a[k].im * b[k].im;
        im += a[k].re * b[k].im - a[k].im * b[k].re;
    }
    return re * re + im * im;
}

/* K: N x M row-major.  py == NULL: symmetric (M = N), upper triangle + mirror, unit diagonal. */
int qko_fidelity(const cplx* px, int64_t N, const cplx* py, int64_t M, int64_t D, double* K) {
    if (!py) {
        for (int64_t i = 0; i < N; ++i) {
            K[i * N + i] = 1.0;
            for (int64_t j = i + 1; j < N; ++j) {
                const double v = fid(px + i * D, px + j * D, D);
                K[i * N + j] = v;
                K[j * N + i] = v;
            }
        }
    } else {
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < M; ++j) K[i * M + j] = fid(px + i * D, py + j * D, D);
    }
    return 0;
}

/* rdm: N x n x 4 = (rho00, rho11, Re rho01, Im rho01) */
int qko_rdm1(const cplx* psi, int64_t N, int n, double* rdm) {
    const int64_t D = (int64_t)1 << n;
    for (int64_t s = 0; s < N; ++s)
        for (int q = 0; q < n; ++q) {
            double r00 = 0, r11 = 0, xr = 0, xi = 0;
            const int64_t bq = (int64_t)1 << q;
            for (int64_t i = 0; i < D; ++i) {
                if (i & bq) continue;
                cplx a0 = psi[s * D + i], a1 = psi[s * D + (i | bq)];
                r00 += a0.re * a0.re + a0.im * a0.im;
                r11 += a1.re * a1.re + a1.im * a1.im;
                xr += a0.re * a1.re + a0.im * a1.im;
                xi += a0.im * a1.re - a0.re * a1.im;
            }
            double* o = rdm + (s * n + q) * 4;
            o[0] = r00; o[1] = r11; o[2] = xr; o[3] = xi;
        }
    return 0;
}

int qko_pqk(const double* rdm, int64_t N, int n, double gamma, double* K) {
    for (int64_t i = 0; i < N; ++i) {
        K[i * N + i] = 1.0;
        for (int64_t j = 0; j < i; ++j) {
            double e = 0;
            for (int q = 0; q < n; ++q) {
                const double* a = rdm + (i * n + q) * 4;
                const double* b = rdm + (j * n + q) * 4;
                const double d0 = a[0] - b[0], d1 = a[1] - b[1], dr = a[2] - b[2], di = a[3] - b[3];
                e += sqrt(d0 * d0 + d1 * d1 + 2 * (dr * dr + di * di));
            }
            K[i * N + j] = K[j * N + i] = exp(-gamma * e);
        }
    }
    return 0;
}
