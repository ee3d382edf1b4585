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

#include <cmath>

#include "qk_common.h"

namespace qk {

struct SimArgs {
    const double* X;
    int64_t ldx;
    double2* psi;
    int64_t D;  // amplitudes per sample
    const DevStage* stages;
    const DevTerm* terms;
    const DevExpr* exprs;
    const DevMat* mats;
    const qk_instr* prog;
    const double* consts;
    int n_stage, n_expr, n_mat;
    int L, lgT, n_qubits, first;
    int nruns;
    uint32_t run_mask[kMaxRuns];
    int32_t run_shift[kMaxRuns];
};

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ double2 cfma(double2 a, double2 b, double2 c) {  // a*b + c
    return make_double2(fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y)));
}

__device__ __forceinline__ uint32_t deposit(uint32_t w, const SimArgs& a) {
    uint32_t g = 0;
    for (int r = 0; r < a.nruns; ++r) {
        const uint32_t m = w & a.run_mask[r];
        const int sh = a.run_shift[r];
        g |= sh >= 0 ? (m << sh) : (m >> (-sh));
    }
    return g;
}

// factor of one diagonal term for the amplitude whose virtual index is w
__device__ __forceinline__ void apply_term(const DevTerm& t, uint32_t w, const double2* cs, double& fr, double& fi) {
    const uint32_t ba = (w >> t.va) & 1u;
    if (t.type == TM_CZ) {
        const uint32_t bb = (w >> t.vb) & 1u;
        if (ba & bb) { fr = -fr; fi = -fi; }
        return;
    }
    const double2 c = cs[t.cs];
    double s;
    bool on = true;
    if (t.type == TM_RZ) {
        s = ba ? c.y : -c.y;       // diag(e^{-i th/2}, e^{+i th/2})
    } else if (t.type == TM_P) {
        on = ba;                   // diag(1, e^{i lam})
        s = c.y;
    } else {                       // TM_ZZP: e^{i lam} where the two bits differ
        const uint32_t bb = (w >> t.vb) & 1u;
        on = (ba ^ bb);
        s = c.y;
    }
    if (on) {
        const double nr = fma(fr, c.x, -(fi * s));
        const double ni = fma(fr, s, fi * c.x);
        fr = nr;
        fi = ni;
    }
}

__global__ void __launch_bounds__(1024) sim_pass_kernel(SimArgs a) {
    extern __shared__ double2 lds[];
    const int L = a.L;
    const uint32_t nloc = 1u << L;
    double2* state = lds;
    double2* cs = lds + nloc;                                   // n_expr (cos, sin) pairs
    double2* mat = cs + a.n_expr;                               // n_mat * 4 complex entries
    const int tid = threadIdx.x;
    const int T = blockDim.x;
    const int nchunk_bits = a.n_qubits - L;
    const uint32_t chunk = blockIdx.x & ((1u << nchunk_bits) - 1u);
    const int64_t sample = blockIdx.x >> nchunk_bits;
    const double* xrow = a.X + sample * a.ldx;

    // ---- per-sample gate angles -> (cos, sin) ------------------------------
    for (int e = tid; e < a.n_expr; e += T) {
        const DevExpr ex = a.exprs[e];
        double st[16];
        int sp = 0;
        for (int k = 0; k < ex.len; ++k) {
            const qk_instr in = a.prog[ex.start + k];
            switch (in.op) {
                case QK_EX_CONST: st[sp++] = a.consts[in.arg]; break;
                case QK_EX_PARAM: st[sp++] = xrow[in.arg]; break;
                case QK_EX_ADD: --sp; st[sp - 1] = st[sp - 1] + st[sp]; break;
                case QK_EX_SUB: --sp; st[sp - 1] = st[sp - 1] - st[sp]; break;
                case QK_EX_MUL: --sp; st[sp - 1] = st[sp - 1] * st[sp]; break;
                case QK_EX_DIV: --sp; st[sp - 1] = st[sp - 1] / st[sp]; break;
                case QK_EX_NEG: st[sp - 1] = -st[sp - 1]; break;
                case QK_EX_ASIN: st[sp - 1] = asin(st[sp - 1]); break;
                case QK_EX_ACOS: st[sp - 1] = acos(st[sp - 1]); break;
                case QK_EX_SIN: st[sp - 1] = sin(st[sp - 1]); break;
                case QK_EX_COS: st[sp - 1] = cos(st[sp - 1]); break;
                default: break;
            }
        }
        const double ang = st[0] * ex.scale;
        double sn, cn;
        sincos(ang, &sn, &cn);
        cs[e] = make_double2(cn, sn);
    }
    // ---- initial state / chunk load ----------------------------------------
    const uint32_t chunk_phys = deposit(chunk << L, a);
    double2* gpsi = a.psi + sample * a.D;
    if (a.first) {
        for (uint32_t l = tid; l < nloc; l += T) state[l] = make_double2(0.0, 0.0);
    } else {
        for (uint32_t l = tid; l < nloc; l += T) state[l] = gpsi[chunk_phys | deposit(l, a)];
    }
    __syncthreads();
    if (a.first && chunk == 0 && tid == 0) state[0] = make_double2(1.0, 0.0);
    // ---- 2x2 matrices -------------------------------------------------------
    for (int m = tid; m < a.n_mat; m += T) {
        const DevMat dm = a.mats[m];
        double2 m00, m01, m10, m11;
        const double r = 0.70710678118654752440;
        if (dm.gate == QK_H) {
            m00 = make_double2(r, 0); m01 = make_double2(r, 0); m10 = make_double2(r, 0); m11 = make_double2(-r, 0);
        } else if (dm.gate == QK_X) {
            m00 = make_double2(0, 0); m01 = make_double2(1, 0); m10 = make_double2(1, 0); m11 = make_double2(0, 0);
        } else if (dm.gate == QK_SX) {
            m00 = make_double2(0.5, 0.5); m01 = make_double2(0.5, -0.5);
            m10 = make_double2(0.5, -0.5); m11 = make_double2(0.5, 0.5);
        } else {
            const double2 c = cs[dm.cs];  // cos, sin of theta/2
            if (dm.gate == QK_RX) {
                m00 = make_double2(c.x, 0); m01 = make_double2(0, -c.y);
                m10 = make_double2(0, -c.y); m11 = make_double2(c.x, 0);
            } else {  // QK_RY
                m00 = make_double2(c.x, 0); m01 = make_double2(-c.y, 0);
                m10 = make_double2(c.y, 0); m11 = make_double2(c.x, 0);
            }
        }
        mat[4 * m + 0] = m00; mat[4 * m + 1] = m01; mat[4 * m + 2] = m10; mat[4 * m + 3] = m11;
    }
    __syncthreads();

    // ---- stages ---------------------------------------------------------------
    const uint32_t whi = chunk << L;
    for (int s = 0; s < a.n_stage; ++s) {
        const DevStage st = a.stages[s];
        if (st.kind == ST_DIAG) {
            double zr = 1.0, zi = 0.0;
            const DevTerm* tA = a.terms + st.t0;
            const DevTerm* tB = tA + st.nA;
            const uint32_t w0 = whi | (uint32_t)tid;
            for (int k = 0; k < st.nA; ++k) apply_term(tA[k], w0, cs, zr, zi);
            for (uint32_t l = tid; l < nloc; l += T) {
                double fr = zr, fi = zi;
                const uint32_t w = whi | l;
                for (int k = 0; k < st.nB; ++k) apply_term(tB[k], w, cs, fr, fi);
                const double2 v = state[l];
                state[l] = make_double2(fma(v.x, fr, -(v.y * fi)), fma(v.x, fi, v.y * fr));
            }
        } else if (st.kind == ST_U1) {
            const int p = st.va;
            const double2 m00 = mat[4 * st.idx + 0], m01 = mat[4 * st.idx + 1];
            const double2 m10 = mat[4 * st.idx + 2], m11 = mat[4 * st.idx + 3];
            const uint32_t lowmask = (1u << p) - 1u;
            for (uint32_t i = tid; i < (nloc >> 1); i += T) {
                const uint32_t i0 = ((i & ~lowmask) << 1) | (i & lowmask);
                const uint32_t i1 = i0 | (1u << p);
                const double2 a0 = state[i0], a1 = state[i1];
                state[i0] = cfma(m01, a1, cmul(m00, a0));
                state[i1] = cfma(m11, a1, cmul(m10, a0));
            }
        } else if (st.kind == ST_CX) {
            const int vc = st.va, vt = st.vb;
            if (vc >= L) {  // control is a chunk bit: X on the target, or nothing
                if ((whi >> vc) & 1u) {
                    const uint32_t lowmask = (1u << vt) - 1u;
                    for (uint32_t i = tid; i < (nloc >> 1); i += T) {
                        const uint32_t i0 = ((i & ~lowmask) << 1) | (i & lowmask);
                        const uint32_t i1 = i0 | (1u << vt);
                        const double2 a0 = state[i0];
                        state[i0] = state[i1];
                        state[i1] = a0;
                    }
                }
            } else {
                const int lo = vc < vt ? vc : vt, hi = vc < vt ? vt : vc;
                const uint32_t m_lo = (1u << lo) - 1u, m_hi = (1u << (hi - 1)) - 1u;
                for (uint32_t i = tid; i < (nloc >> 2); i += T) {
                    uint32_t b = ((i & ~m_lo) << 1) | (i & m_lo);      // zero at lo
                    b = ((b & ~((m_hi << 1) | 1u)) << 1) | (b & ((m_hi << 1) | 1u));  // zero at hi
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
            const uint32_t m_lo = (1u << lo) - 1u;
            const uint32_t m_hi2 = (1u << hi) - 1u;  // mask below hi after inserting lo
            const double r = 0.70710678118654752440;
            for (uint32_t i = tid; i < (nloc >> 2); i += T) {
                uint32_t b = ((i & ~m_lo) << 1) | (i & m_lo);
                b = ((b & ~m_hi2) << 1) | (b & m_hi2);
                const uint32_t j0 = b, j1 = b | (1u << v0), j2 = b | (1u << v1), j3 = j1 | (1u << v1);
                const double2 a0 = state[j0], a1 = state[j1], a2 = state[j2], a3 = state[j3];
                // 1/sqrt2 * [[0,1,0,i],[1,0,-i,0],[0,i,0,1],[-i,0,1,0]]
                state[j0] = make_double2(r * (a1.x - a3.y), r * (a1.y + a3.x));
                state[j1] = make_double2(r * (a0.x + a2.y), r * (a0.y - a2.x));
                state[j2] = make_double2(r * (a3.x - a1.y), r * (a3.y + a1.x));
                state[j3] = make_double2(r * (a2.x + a0.y), r * (a2.y - a0.x));
            }
        }
        __syncthreads();
    }
    // ---- write back -----------------------------------------------------------
    for (uint32_t l = tid; l < nloc; l += T) gpsi[chunk_phys | deposit(l, a)] = state[l];
}

int upload_circuit(qk_circuit& c) {
    int dev = 0;
    QK_HIP(hipGetDevice(&dev));
    if (c.device == dev) return QK_OK;
    free_circuit_device(c);
    auto up = [&](auto*& dptr, const auto& vec) -> hipError_t {
        using T = typename std::remove_reference<decltype(vec)>::type::value_type;
        const size_t bytes = std::max<size_t>(vec.size(), 1) * sizeof(T);
        hipError_t e = hipMalloc((void**)&dptr, bytes);
        if (e != hipSuccess) return e;
        if (!vec.empty()) e = hipMemcpy(dptr, vec.data(), vec.size() * sizeof(T), hipMemcpyHostToDevice);
        return e;
    };
    QK_HIP(up(c.d_stages, c.stages));
    QK_HIP(up(c.d_terms, c.terms));
    QK_HIP(up(c.d_exprs, c.exprs));
    QK_HIP(up(c.d_mats, c.mats));
    QK_HIP(up(c.d_prog, c.prog));
    QK_HIP(up(c.d_consts, c.consts));
    c.device = dev;
    return QK_OK;
}

void free_circuit_device(qk_circuit& c) {
    if (c.device < 0) return;
    (void)hipFree(c.d_stages); (void)hipFree(c.d_terms); (void)hipFree(c.d_exprs);
    (void)hipFree(c.d_mats); (void)hipFree(c.d_prog); (void)hipFree(c.d_consts);
    c.d_stages = nullptr; c.d_terms = nullptr; c.d_exprs = nullptr;
    c.d_mats = nullptr; c.d_prog = nullptr; c.d_consts = nullptr;
    c.device = -1;
}

}  // namespace qk

extern "C" int qk_simulate_dev(const qk_circuit* cc, const double* X_dev, int64_t N, int64_t ldx,
                               void* psi_dev, void* stream) {
    using namespace qk;
    if (!cc || !psi_dev) return fail(QK_ERR_INVALID, "qk_simulate_dev: null circuit or output");
    if (N < 0) return fail(QK_ERR_INVALID, "qk_simulate_dev: negative N");
    if (N == 0) return QK_OK;
    qk_circuit& c = *const_cast<qk_circuit*>(cc);
    if (c.n_params > 0 && (!X_dev || ldx < c.n_params))
        return fail(QK_ERR_INVALID, "qk_simulate_dev: X is null or its row stride is smaller than the parameter count");
    int rc = upload_circuit(c);
    if (rc) return rc;
    const int n = c.n_qubits;
    static bool attr_set = false;
    if (!attr_set) {
        QK_HIP(hipFuncSetAttribute((const void*)sim_pass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    for (const Pass& ps : c.passes) {
        SimArgs a{};
        a.X = X_dev;
        a.ldx = ldx;
        a.psi = (double2*)psi_dev;
        a.D = (int64_t)1 << n;
        a.stages = c.d_stages + ps.stage_off;
        a.terms = c.d_terms;
        a.exprs = c.d_exprs + ps.expr_off;
        a.mats = c.d_mats + ps.mat_off;
        a.prog = c.d_prog;
        a.consts = c.d_consts;
        a.n_stage = ps.n_stage;
        a.n_expr = ps.n_expr;
        a.n_mat = ps.n_mat;
        a.L = ps.L;
        a.lgT = ps.lgT;
        a.n_qubits = n;
        a.first = ps.first ? 1 : 0;
        a.nruns = ps.nruns;
        for (int r = 0; r < ps.nruns; ++r) { a.run_mask[r] = ps.run_mask[r]; a.run_shift[r] = ps.run_shift[r]; }
        const int64_t nblocks = N << (n - ps.L);
        if (nblocks > 0x7fffffffLL) return fail(QK_ERR_UNSUPPORTED, "qk_simulate_dev: too many workgroups; split N");
        const size_t lds = ((size_t)16 << ps.L) + (size_t)16 * ps.n_expr + (size_t)64 * ps.n_mat;
        hipLaunchKernelGGL(sim_pass_kernel, dim3((unsigned)nblocks), dim3(1u << ps.lgT), lds, (hipStream_t)stream, a);
        QK_HIP(hipGetLastError());
    }
    return QK_OK;
}
