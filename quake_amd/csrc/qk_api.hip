// C ABI glue: error reporting, circuit handles and the host-pointer entry points.
#include <hip/hip_runtime.h>

#include <cstdio>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "qk_common.h"
#include "qk_reg.h"

namespace qk {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

namespace {
// Where the wall time of the calling thread's last qk_evaluate / qk_evaluate_f32 went (seconds; development hook
// qk_evaluate_profile, used by bench.py to itemise the PCIe-inclusive call of SURVEY.md 8d):
//   [0] device allocations   [1] host-to-device copies of X / Y   [2] enqueueing the simulations and Gram bands
//   [3] waiting for the bands while copying the finished ones to the host (the device work, with all copies but
//       the last band's hidden behind it)   [4] what remained after the last kernel had finished: the last band's
//       copy   [5] releasing the device buffers   [6] the whole call
struct EvalProfile { double t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };
thread_local EvalProfile g_profile;
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Lap {      // adds the time since the previous lap to a slot of the profile
    double last = now_s();
    void to(int slot) { const double t = now_s(); g_profile.t[slot] += t - last; last = t; }
};

// RAII device buffer for the host-pointer entry points
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { release(); }
    void release() {
        if (!p) return;
        const double t0 = now_s();
        (void)hipFree(p);
        g_profile.t[5] += now_s() - t0;
        p = nullptr;
    }
    hipError_t alloc(size_t bytes) {
        const double t0 = now_s();
        const hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        g_profile.t[0] += now_s() - t0;
        return e;
    }
};

int check_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
        return fail(QK_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    return QK_OK;
}

// X (N x F, row stride F) -> device; returns pointer or nullptr when F == 0
int upload_X(const double* X, int64_t N, int F, DevBuf& buf) {
    if (F == 0 || N == 0) return QK_OK;
    QK_HIP(buf.alloc((size_t)N * F * sizeof(double)));
    const double t0 = now_s();
    QK_HIP(hipMemcpy(buf.p, X, (size_t)N * F * sizeof(double), hipMemcpyHostToDevice));
    g_profile.t[1] += now_s() - t0;
    return QK_OK;
}
}  // namespace
}  // namespace qk

using namespace qk;

extern "C" int qk_version(void) { return 100; }

extern "C" int qk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int qk_last_error(char* buf, size_t len) {
    if (!buf || len == 0) return QK_ERR_INVALID;
    const size_t k = g_last_error.size() < len - 1 ? g_last_error.size() : len - 1;
    std::memcpy(buf, g_last_error.data(), k);
    buf[k] = 0;
    return QK_OK;
}

extern "C" int qk_circuit_create(int n_qubits, int n_params, const qk_op* ops, int n_ops, const qk_instr* prog,
                                 int n_instr, const int32_t* expr_off, int n_expr, const double* consts,
                                 int n_consts, qk_circuit** out) {
    if (!out) return fail(QK_ERR_INVALID, "qk_circuit_create: out is null");
    *out = nullptr;
    if (n_qubits < 1 || n_qubits > 26) return fail(QK_ERR_INVALID, "qk_circuit_create: n_qubits must be in 1..26");
    if (n_params < 0 || n_ops < 0 || n_instr < 0 || n_expr < 0 || n_consts < 0)
        return fail(QK_ERR_INVALID, "qk_circuit_create: negative count");
    if ((n_ops && !ops) || (n_instr && !prog) || (n_expr && !expr_off) || (n_consts && !consts))
        return fail(QK_ERR_INVALID, "qk_circuit_create: null table");
    // validate the angle programs once, so the device evaluator needs no checks
    for (int e = 0; e < n_expr; ++e) {
        const int s = expr_off[e], t = expr_off[e + 1];
        if (s < 0 || t < s || t > n_instr) return fail(QK_ERR_INVALID, "qk_circuit_create: bad expr_off");
        int depth = 0;
        for (int k = s; k < t; ++k) {
            const qk_instr& in = prog[k];
            switch (in.op) {
                case QK_EX_CONST:
                    if (in.arg < 0 || in.arg >= n_consts) return fail(QK_ERR_INVALID, "angle program: constant index out of range");
                    ++depth;
                    break;
                case QK_EX_PARAM:
                    if (in.arg < 0 || in.arg >= n_params) return fail(QK_ERR_INVALID, "angle program: parameter column out of range");
                    ++depth;
                    break;
                case QK_EX_ADD: case QK_EX_SUB: case QK_EX_MUL: case QK_EX_DIV:
                    if (depth < 2) return fail(QK_ERR_INVALID, "angle program: stack underflow");
                    --depth;
                    break;
                case QK_EX_NEG: case QK_EX_ASIN: case QK_EX_ACOS: case QK_EX_SIN: case QK_EX_COS:
                    if (depth < 1) return fail(QK_ERR_INVALID, "angle program: stack underflow");
                    break;
                default:
                    return fail(QK_ERR_INVALID, "angle program: unknown opcode");
            }
            if (depth > 8) return fail(QK_ERR_INVALID, "angle program: deeper than 8 stack slots");
        }
        if (depth != 1) return fail(QK_ERR_INVALID, "angle program: does not leave exactly one value");
    }
    qk_circuit* c = new (std::nothrow) qk_circuit();
    if (!c) return fail(QK_ERR_INVALID, "qk_circuit_create: out of memory");
    c->n_qubits = n_qubits;
    c->n_params = n_params;
    c->prog.assign(prog, prog + n_instr);
    c->consts.assign(consts, consts + n_consts);
    const int rc = plan_circuit(*c, ops, n_ops, expr_off, n_expr);
    if (rc) {
        delete c;
        return rc;
    }
    bool any_narrow = false;
    for (const Pass& ps : c->passes) any_narrow |= ps.narrow;
    // ... and so does a circuit whose register-resident plan holds 8 amplitudes per thread: that kernel's reduced-density-
    // matrix mode needs the 16-amplitude form (qk_reg.hip, reg_rdm_mode_ok)
    if (!c->reg_blob.empty()) {
        const RegHeader* H = (const RegHeader*)c->reg_blob.data();
        any_narrow |= H->rb == 3 && H->nq == H->L;
    }
    if (any_narrow) {      // the fused simulate + RDM mode runs its own plan of the same gates (see qk_circuit::rdm_plan)
        qk_circuit* r = new (std::nothrow) qk_circuit();
        if (!r) { delete c; return fail(QK_ERR_INVALID, "qk_circuit_create: out of memory"); }
        r->n_qubits = n_qubits;
        r->n_params = n_params;
        r->prog.assign(prog, prog + n_instr);      // (the caller's tables: plan_circuit may have appended a constant program to c's)
        r->consts.assign(consts, consts + n_consts);
        const int rc2 = plan_circuit(*r, ops, n_ops, expr_off, n_expr, false);
        if (rc2) { delete r; delete c; return rc2; }
        c->rdm_plan = r;
    }
    *out = c;
    return QK_OK;
}

extern "C" void qk_circuit_destroy(qk_circuit* c) {
    if (!c) return;
    free_circuit_device(*c);
    if (c->rdm_plan) free_circuit_device(*c->rdm_plan);
    delete c;
}

extern "C" int qk_circuit_num_qubits(const qk_circuit* c) { return c ? c->n_qubits : QK_ERR_INVALID; }
extern "C" int qk_circuit_num_params(const qk_circuit* c) { return c ? c->n_params : QK_ERR_INVALID; }
extern "C" int qk_circuit_num_passes(const qk_circuit* c) { return c ? (int)c->passes.size() : QK_ERR_INVALID; }

// Development hook (not in the public header): what the planner made of the gate list, as text.
extern "C" int qk_circuit_plan_summary(const qk_circuit* c, char* buf, int len) {
    if (!c || !buf || len < 1) return QK_ERR_INVALID;
    std::string s = c->plan_summary + "; passes " + std::to_string(c->passes.size());
    for (size_t pi = 0; pi < c->passes.size(); ++pi) {     // per pass: local qubits, threads, sweep radices, stage counts
        const Pass& ps = c->passes[pi];
        int n_u1 = 0, n_diag = 0, n_other = 0, max_ng = 0;
        for (int k = ps.n_init; k < ps.n_stage; ++k) {
            const DevStage& st = c->stages[ps.stage_off + k];
            if (st.kind == ST_U1) { ++n_u1; max_ng = std::max(max_ng, (int)st.ng); }
            else if (st.kind == ST_DIAG) ++n_diag;
            else ++n_other;
        }
        s += "; pass" + std::to_string(pi) + " L" + std::to_string(ps.L) + " T" + std::to_string(1 << ps.lgT) + " maxg" +
             std::to_string(ps.maxg) + " maxh" + std::to_string(ps.maxh) + " narrow" + std::to_string(ps.narrow ? 1 : 0) + " u1=" +
             std::to_string(n_u1) + "(<=" + std::to_string(max_ng) + ") diag=" + std::to_string(n_diag) + " other=" +
             std::to_string(n_other);
    }
    s += c->rdm_plan ? "; rdm_plan yes" : "; rdm_plan no";
    s += c->wave_blob.empty() ? "; wave_program no" : "; wave_program yes";
    std::snprintf(buf, (size_t)len, "%s", s.c_str());
    return QK_OK;
}

extern "C" int qk_host_alloc(size_t bytes, void** out) {
    using namespace qk;
    if (!out) return fail(QK_ERR_INVALID, "qk_host_alloc: out is null");
    *out = nullptr;
    int rc = check_device();
    if (rc) return rc;
    QK_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return QK_OK;
}

extern "C" int qk_host_free(void* p) {
    using namespace qk;
    if (!p) return QK_OK;
    QK_HIP(hipHostFree(p));
    return QK_OK;
}

extern "C" int qk_block_to_host_async(const double* K_dev, int64_t ldk, int64_t rows, int64_t cols, double* K_host,
                                      int64_t ldh, void* stream) {
    using namespace qk;
    if (rows < 0 || cols < 0) return fail(QK_ERR_INVALID, "qk_block_to_host_async: negative block size");
    if (rows == 0 || cols == 0) return QK_OK;
    if (!K_dev || !K_host) return fail(QK_ERR_INVALID, "qk_block_to_host_async: null pointer");
    if (ldk < cols || ldh < cols) return fail(QK_ERR_INVALID, "qk_block_to_host_async: row stride smaller than the block");
    QK_HIP(hipMemcpy2DAsync(K_host, (size_t)ldh * sizeof(double), K_dev, (size_t)ldk * sizeof(double),
                            (size_t)cols * sizeof(double), (size_t)rows, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return QK_OK;
}

extern "C" int qk_simulate(const qk_circuit* c, const double* X, int64_t N, int64_t ldx, void* psi_host) {
    if (!c || !psi_host) return fail(QK_ERR_INVALID, "qk_simulate: null pointer");
    if (N < 0) return fail(QK_ERR_INVALID, "qk_simulate: negative N");
    if (N == 0) return QK_OK;
    int rc = check_device();
    if (rc) return rc;
    const int F = c->n_params;
    if (F > 0 && (!X || ldx < F)) return fail(QK_ERR_INVALID, "qk_simulate: X is null or ldx < n_params");
    DevBuf dX, dPsi;
    if (F > 0) {
        QK_HIP(dX.alloc((size_t)N * F * sizeof(double)));
        QK_HIP(hipMemcpy2D(dX.p, F * sizeof(double), X, ldx * sizeof(double), F * sizeof(double), N,
                           hipMemcpyHostToDevice));
    }
    const size_t bytes = ((size_t)N << c->n_qubits) * 16;
    QK_HIP(dPsi.alloc(bytes));
    rc = qk_simulate_dev(c, (const double*)dX.p, N, F, dPsi.p, nullptr);
    if (rc) return rc;
    QK_HIP(hipMemcpy(psi_host, dPsi.p, bytes, hipMemcpyDeviceToHost));
    return QK_OK;
}

namespace qk {
namespace {
// side streams of the host entry points, one pair per device, created on first use and kept
int side_streams(hipStream_t* compute, hipStream_t* copy) {
    static std::mutex mu;
    static hipStream_t streams[64][2] = {};
    int dev = 0;
    QK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(QK_ERR_UNSUPPORTED, "device index out of range");
    std::lock_guard<std::mutex> lock(mu);
    for (int k = 0; k < 2; ++k)
        if (!streams[dev][k]) QK_HIP(hipStreamCreateWithFlags(&streams[dev][k], hipStreamNonBlocking));
    *compute = streams[dev][0];
    *copy = streams[dev][1];
    return QK_OK;
}

// The Gram stage of the host entry points, with the copy of K back to the host hidden behind it.
// K is produced in row bands (each band = one multi-job launch: its diagonal block as a symmetric job,
// the block to its right as a cross job whose transposed copy fills the mirror entries of the later
// bands).  A band's rows are final once its launch has finished -- everything left of its diagonal
// block was written by earlier bands -- so the host copies band b while the GPU computes band b + 1.
// f32: the statevectors are complex64 (qk_gram32_jobs_dev).
struct NoiseSpec {
    int mode = QK_NOISE_NONE;
    double param = 0.0;
    uint64_t seed = 0;
};

int gram_to_host(const void* dPx, int64_t N, const void* dPy, int64_t M, int64_t D, bool f32, double* K_host,
                 const NoiseSpec& noise = NoiseSpec()) {
    const bool sym = (dPy == nullptr);
    const size_t esz = f32 ? 8 : 16;
    auto launch = f32 ? qk_gram32_jobs_dev : qk_gram_jobs_dev;
    DevBuf dK;
    QK_HIP(dK.alloc((size_t)N * M * sizeof(double)));
    double* K = (double*)dK.p;
    // Bands halve: rows [0, N/2), then half of the rest, ... down to about 1/32 of the rows (at least 1024, multiples of
    // 128).  What the copies need is that the LAST band is small -- its copy is the only one no kernel runs behind --
    // while a launch runs the closer to the rate of one big job the larger its blocks are: sixteen equal bands cost
    // 4.4 % over a single launch at the headline size (the rows of a late band are short symmetric blocks next to
    // a sliver of a cross block), about what their overlapped copies saved; six halving bands cost well under 1 %.
    // (Short rows -- under 11 qubits the copy of K takes longer than computing it, 57 GB/s against 8 D flops per entry --
    // keep sixteen equal bands: there the first band is what no copy overlaps.)
    std::vector<int64_t> cut{0};
    if (D >= 2048) {
        while (N - cut.back() >= 2048 && cut.size() < 6) {
            const int64_t take = ((N - cut.back()) / 2 + 127) / 128 * 128;
            if (take < 1024 || N - cut.back() - take < 1024) break;
            cut.push_back(cut.back() + take);
            if (N - cut.back() <= std::max<int64_t>(1024, N / 32)) break;
        }
    } else {
        const int64_t nb = std::min<int64_t>(16, N / 1024);
        const int64_t band = nb >= 2 ? ((N + nb - 1) / nb + 127) / 128 * 128 : N;
        while (N - cut.back() > band) cut.push_back(cut.back() + band);
    }
    cut.push_back(N);
    const int64_t nbands = (int64_t)cut.size() - 1;
    if (nbands == 1) {
        // small problems (the reference's own sizes): one launch on the null stream behind the simulation,
        // one blocking copy -- no streams or events to create, the call is latency-bound
        qk_gram_job job;
        std::memset(&job, 0, sizeof(job));
        job.A = dPx; job.B = sym ? dPx : dPy; job.K = K;
        job.ma = N; job.nb = M; job.ldk = M; job.flags = sym ? QK_GRAM_SYMMETRIC : 0;
        int rc1 = launch(&job, 1, D, nullptr);
        if (rc1) return rc1;
        if ((rc1 = launch_noise(K, 0, N, M, M, sym, noise.mode, noise.param, noise.seed, nullptr))) return rc1;
        QK_HIP(hipMemcpy(K_host, K, (size_t)N * M * sizeof(double), hipMemcpyDeviceToHost));
        return QK_OK;
    }
    // two side streams per device, created once and kept (a host application's own streams are left alone: no
    // device-wide synchronisation here)
    Lap lap;
    hipStream_t compute = nullptr, copy = nullptr;
    int rc = side_streams(&compute, &copy);
    if (rc) return rc;
    std::vector<hipEvent_t> done;
    auto cleanup = [&]() {
        for (hipEvent_t e : done) (void)hipEventDestroy(e);
    };
    // the simulations ran on the null stream: order the compute stream behind them
    {
        hipEvent_t sim_done;
        if (hipEventCreateWithFlags(&sim_done, hipEventDisableTiming) != hipSuccess)
            return fail(QK_ERR_HIP, "gram_to_host: cannot create event");
        done.push_back(sim_done);
        if (hipEventRecord(sim_done, nullptr) != hipSuccess || hipStreamWaitEvent(compute, sim_done, 0) != hipSuccess) {
            cleanup();
            return fail(QK_ERR_HIP, "gram_to_host: cannot order the compute stream behind the simulation");
        }
    }
    const size_t first_band_event = done.size();
    for (int64_t bi = 0; bi < nbands && rc == QK_OK; ++bi) {
        const int64_t r0 = cut[bi], r1 = cut[bi + 1];
        qk_gram_job jobs[2];
        std::memset(jobs, 0, sizeof(jobs));
        int nj = 0;
        const char* A = (const char*)dPx + (size_t)r0 * D * esz;
        if (sym) {
            jobs[nj].A = A; jobs[nj].B = A; jobs[nj].K = K + r0 * M + r0;
            jobs[nj].ma = jobs[nj].nb = r1 - r0; jobs[nj].ldk = M; jobs[nj].flags = QK_GRAM_SYMMETRIC;
            ++nj;
            if (r1 < N) {
                jobs[nj].A = A; jobs[nj].B = (const char*)dPx + (size_t)r1 * D * esz;
                jobs[nj].K = K + r0 * M + r1; jobs[nj].KT = K + r1 * M + r0;
                jobs[nj].ma = r1 - r0; jobs[nj].nb = N - r1; jobs[nj].ldk = M; jobs[nj].ldkt = M;
                ++nj;
            }
        } else {
            jobs[nj].A = A; jobs[nj].B = dPy; jobs[nj].K = K + r0 * M;
            jobs[nj].ma = r1 - r0; jobs[nj].nb = M; jobs[nj].ldk = M;
            ++nj;
        }
        rc = launch(jobs, nj, D, compute);
        if (rc != QK_OK) break;
        // a band's draws: its rows right of the diagonal, mirrored into the later bands' rows (symmetric), or all of
        // its rows (cross block); ordered behind the band's Gram launch on the same stream
        rc = launch_noise(K, r0, r1, M, M, sym, noise.mode, noise.param, noise.seed, compute);
        if (rc != QK_OK) break;
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            rc = fail(QK_ERR_HIP, "gram_to_host: cannot create event");
            break;
        }
        done.push_back(e);      // owned by `done` from here on, whatever happens to the record
        if (hipEventRecord(e, compute) != hipSuccess) rc = fail(QK_ERR_HIP, "gram_to_host: cannot record event");
    }
    // every launch is queued; now drain band by band while the later bands compute
    lap.to(2);
    size_t b = first_band_event;
    for (int64_t bi = 0; bi < nbands && rc == QK_OK; ++bi, ++b) {
        const int64_t r0 = cut[bi], r1 = cut[bi + 1];
        if (r1 >= N) {      // the last band: from here on no kernel runs behind the copy
            if (hipEventSynchronize(done[b]) != hipSuccess) rc = fail(QK_ERR_HIP, "gram_to_host: synchronisation failed");
            lap.to(3);
        }
        if (hipStreamWaitEvent(copy, done[b], 0) != hipSuccess ||
            hipMemcpyAsync(K_host + r0 * M, K + r0 * M, (size_t)(r1 - r0) * M * sizeof(double), hipMemcpyDeviceToHost, copy) !=
                hipSuccess ||
            hipStreamSynchronize(copy) != hipSuccess)
            rc = fail(QK_ERR_HIP, "gram_to_host: copy of K failed");
    }
    // on an error path kernels may still be running on the buffers the caller is about to free
    if (hipStreamSynchronize(compute) != hipSuccess && rc == QK_OK) rc = fail(QK_ERR_HIP, "gram_to_host: synchronisation failed");
    if (hipStreamSynchronize(copy) != hipSuccess && rc == QK_OK) rc = fail(QK_ERR_HIP, "gram_to_host: synchronisation failed");
    lap.to(4);
    cleanup();
    return rc;
}
}  // namespace
}  // namespace qk

namespace qk {
namespace {
int evaluate_host(const char* who, const qk_circuit* c, const double* X, int64_t N, const double* Y, int64_t M, int n_features,
                  bool f32, const NoiseSpec& noise, double* K_host) {
    if (!c || !K_host) return fail(QK_ERR_INVALID, std::string(who) + ": null pointer");
    if (n_features != c->n_params)
        return fail(QK_ERR_INVALID, "x_vec and class feature map have incompatible dimensions");
    if (N < 0 || (Y && M < 0)) return fail(QK_ERR_INVALID, std::string(who) + ": negative row count");
    if (c->n_params > 0 && ((N && !X))) return fail(QK_ERR_INVALID, std::string(who) + ": X is null");
    const bool sym = (Y == nullptr);
    if (sym) M = N;
    if (N == 0 || M == 0) return QK_OK;
    int rc = check_device();
    if (rc) return rc;
    g_profile = EvalProfile();
    const double t_begin = now_s();
    const int F = c->n_params;
    const int64_t D = (int64_t)1 << c->n_qubits;
    const size_t esz = f32 ? 8 : 16;
    {
        DevBuf dX, dY, dPx, dPy, dWs;
        // FP32 statevectors of a multi-pass circuit are rounded from an FP64 workspace
        const int64_t ws_rows = (f32 && c->passes.size() > 1) ? std::max(N, M) : 0;
        if (ws_rows) QK_HIP(dWs.alloc((size_t)ws_rows * D * 16));
        auto simulate = [&](const double* x_dev, int64_t rows, void* psi) {
            const double t0 = now_s();
            const int r = f32 ? qk_simulate_c64_dev(c, x_dev, rows, F, psi, dWs.p, nullptr)
                              : qk_simulate_dev(c, x_dev, rows, F, psi, nullptr);
            g_profile.t[2] += now_s() - t0;
            return r;
        };
        if ((rc = upload_X(X, N, F, dX))) return rc;
        QK_HIP(dPx.alloc((size_t)N * D * esz));
        if ((rc = simulate((const double*)dX.p, N, dPx.p))) return rc;
        if (!sym) {
            if ((rc = upload_X(Y, M, F, dY))) return rc;
            QK_HIP(dPy.alloc((size_t)M * D * esz));
            if ((rc = simulate((const double*)dY.p, M, dPy.p))) return rc;
        }
        rc = gram_to_host(dPx.p, N, sym ? nullptr : dPy.p, M, D, f32, K_host, noise);
    }
    g_profile.t[6] = now_s() - t_begin;
    return rc;
}
}  // namespace
}  // namespace qk

extern "C" int qk_evaluate(const qk_circuit* c, const double* X, int64_t N, const double* Y, int64_t M,
                           int n_features, double* K_host) {
    return evaluate_host("qk_evaluate", c, X, N, Y, M, n_features, false, NoiseSpec(), K_host);
}

extern "C" int qk_evaluate_noisy(const qk_circuit* c, const double* X, int64_t N, const double* Y, int64_t M, int n_features,
                                 int f32, int mode, double param, uint64_t seed, double* K_host) {
    NoiseSpec noise;
    noise.mode = mode; noise.param = param; noise.seed = seed;
    if (mode != QK_NOISE_NONE && mode != QK_NOISE_GAUSSIAN && mode != QK_NOISE_SHOTS)
        return fail(QK_ERR_INVALID, "qk_evaluate_noisy: unknown noise mode");
    return evaluate_host("qk_evaluate_noisy", c, X, N, Y, M, n_features, f32 != 0, noise, K_host);
}

// Development hook (not in the public header): the itemised wall time of the calling thread's last qk_evaluate /
// qk_evaluate_f32, see EvalProfile.
extern "C" int qk_evaluate_profile(double* out, int len) {
    if (!out || len < 1) return QK_ERR_INVALID;
    for (int k = 0; k < len && k < 8; ++k) out[k] = g_profile.t[k];
    return QK_OK;
}

extern "C" int qk_evaluate_f32(const qk_circuit* c, const double* X, int64_t N, const double* Y, int64_t M,
                               int n_features, double* K_host) {
    return evaluate_host("qk_evaluate_f32", c, X, N, Y, M, n_features, true, NoiseSpec(), K_host);
}

extern "C" int qk_projected_kernel(const qk_circuit* c, const double* X, int64_t N, int n_features, double gamma,
                                   double* K_host) {
    if (!c || !K_host) return fail(QK_ERR_INVALID, "qk_projected_kernel: null pointer");
    if (N < 0) return fail(QK_ERR_INVALID, "qk_projected_kernel: negative N");
    if (N == 0) return QK_OK;
    if (c->n_params == 0) {  // reference genetic_v3.py:789-791
        for (int64_t k = 0; k < N * N; ++k) K_host[k] = 1.0;
        return QK_OK;
    }
    if (n_features != c->n_params)
        return fail(QK_ERR_INVALID, "dataset and feature map have incompatible dimensions");
    if (!X) return fail(QK_ERR_INVALID, "qk_projected_kernel: X is null");
    int rc = check_device();
    if (rc) return rc;
    const int F = c->n_params, n = c->n_qubits;
    const int64_t D = (int64_t)1 << n;
    DevBuf dX, dPsi, dR, dK;
    if ((rc = upload_X(X, N, F, dX))) return rc;
    QK_HIP(dR.alloc((size_t)N * n * 4 * sizeof(double)));
    if (qk_simulate_rdm1_wants_workspace(c, N)) QK_HIP(dPsi.alloc((size_t)N * D * 16));   // fused: Psi is never materialised
    if ((rc = qk_simulate_rdm1_dev(c, (const double*)dX.p, N, F, (double*)dR.p, dPsi.p, nullptr))) return rc;
    QK_HIP(dK.alloc((size_t)N * N * sizeof(double)));
    if ((rc = qk_pqk_dev((const double*)dR.p, N, nullptr, N, n, gamma, (double*)dK.p, N, nullptr))) return rc;
    QK_HIP(hipMemcpy(K_host, dK.p, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost));
    return QK_OK;
}

namespace qk {
namespace {
constexpr int64_t kBatchSmallRows = 1024;                 // fused one-block-per-chromosome epilogue up to this many rows
constexpr size_t kBatchSmallKBytes = (size_t)1 << 30;     // ... and while all kernel matrices together stay under 1 GiB
constexpr size_t kBatchPsiBytes = (size_t)48 << 30;       // statevectors of one simulate launch

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Chromosomes whose circuit does not run on the wave-resident simulator (more than 10 qubits): one after the
// other on the null stream, one K buffer reused when the matrix is not asked for.
int batch_one_by_one(const std::vector<int>& idx, const qk_circuit* const* circuits, const double* const* X, const int64_t* N,
                     const int32_t* const* labels, int noise, const qk_batch_opts* opts, double* const* K_host, double* stats) {
    size_t psi_bytes = 0, k_bytes = 0, x_bytes = 0, l_bytes = 0;
    for (int g : idx) {
        psi_bytes = std::max(psi_bytes, ((size_t)N[g] << circuits[g]->n_qubits) * 16);
        k_bytes = std::max(k_bytes, (size_t)N[g] * N[g] * 8);
        x_bytes = std::max(x_bytes, (size_t)N[g] * circuits[g]->n_params * 8);
        l_bytes = std::max(l_bytes, (size_t)N[g] * 4);
    }
    DevBuf dX, dL, dPsi, dK, dS, dW;
    QK_HIP(dX.alloc(x_bytes));
    QK_HIP(dPsi.alloc(psi_bytes));
    QK_HIP(dK.alloc(k_bytes));
    if (stats) {
        QK_HIP(dL.alloc(l_bytes));
        QK_HIP(dS.alloc(QK_STATS_LEN * sizeof(double)));
        QK_HIP(dW.alloc((size_t)QK_STATS_WS * sizeof(double)));
    }
    for (int g : idx) {
        const qk_circuit* c = circuits[g];
        const int64_t n = N[g], D = (int64_t)1 << c->n_qubits;
        int rc;
        if (c->n_params) QK_HIP(hipMemcpy(dX.p, X[g], (size_t)n * c->n_params * 8, hipMemcpyHostToDevice));
        if ((rc = qk_simulate_dev(c, (const double*)dX.p, n, c->n_params, dPsi.p, nullptr))) return rc;
        if ((rc = qk_gram_dev(dPsi.p, n, nullptr, n, D, (double*)dK.p, n, nullptr))) return rc;
        if (noise != QK_NOISE_NONE &&
            (rc = qk_kernel_noise_dev((double*)dK.p, n, n, noise, opts->noise_param, opts->seed + (uint64_t)g, nullptr)))
            return rc;
        if (stats) {
            QK_HIP(hipMemcpy(dL.p, labels[g], (size_t)n * 4, hipMemcpyHostToDevice));
            if ((rc = qk_kernel_stats_dev((double*)dK.p, n, n, (const int32_t*)dL.p, (double*)dS.p, (double*)dW.p, nullptr))) return rc;
            QK_HIP(hipMemcpy(stats + (size_t)g * QK_STATS_LEN, dS.p, QK_STATS_LEN * sizeof(double), hipMemcpyDeviceToHost));
        }
        if (K_host && K_host[g]) QK_HIP(hipMemcpy(K_host[g], dK.p, (size_t)n * n * 8, hipMemcpyDeviceToHost));
    }
    return QK_OK;
}

// The heterogeneous batch proper: every chromosome's circuit (a position-independent blob), feature rows, labels and
// the job tables travel in ONE copy; one simulate launch per register class of the wave-resident kernel walks a job
// table over all chromosomes; the Gram stage takes 16 chromosomes per launch (qk_gram_jobs_dev).  Small matrices
// (GA sizes): all of them live at once and one fused launch does noise + statistics for every chromosome.  Large
// ones: per chromosome, through one reused K buffer.
int batch_wave(const std::vector<int>& idx, const qk_circuit* const* circuits, const double* const* X, const int64_t* N,
               const int32_t* const* labels, int noise, const qk_batch_opts* opts, double* const* K_host, double* stats) {
    const int cnt = (int)idx.size();
    bool small = true;
    size_t k_total = 0;
    for (int g : idx) {
        small = small && N[g] <= kBatchSmallRows;
        k_total += (size_t)N[g] * N[g] * 8;
    }
    small = small && k_total <= kBatchSmallKBytes;
    // ---- layout of the staging buffer -------------------------------------------------------------------
    std::vector<size_t> blob_off(cnt), x_off(cnt), l_off(cnt), psi_off(cnt), k_off(cnt);
    size_t off = 0, psi_total = 0, k_live = 0, k_max = 0;
    for (int k = 0; k < cnt; ++k) {
        const qk_circuit* c = circuits[idx[k]];
        blob_off[k] = off; off = align_up(off + c->wave_blob.size() * 8, 256);
    }
    for (int k = 0; k < cnt; ++k) {
        const qk_circuit* c = circuits[idx[k]];
        x_off[k] = off; off = align_up(off + (size_t)N[idx[k]] * c->n_params * 8, 256);
    }
    for (int k = 0; k < cnt; ++k) {
        l_off[k] = off;
        if (stats) off = align_up(off + (size_t)N[idx[k]] * 4, 256);
    }
    const size_t jobs_off = off;     off = align_up(off + (size_t)cnt * sizeof(WaveJob), 256);
    const size_t epi_off = off;      off = align_up(off + (size_t)cnt * sizeof(EpiJob), 256);
    const size_t stage_bytes = off;
    for (int k = 0; k < cnt; ++k) {
        const qk_circuit* c = circuits[idx[k]];
        psi_off[k] = psi_total;
        psi_total += align_up(((size_t)N[idx[k]] << c->n_qubits) * 16, 256);
        k_off[k] = k_live;
        k_live += align_up((size_t)N[idx[k]] * N[idx[k]] * 8, 256);
        k_max = std::max(k_max, (size_t)N[idx[k]] * N[idx[k]] * 8);
    }
    if (psi_total > kBatchPsiBytes) return QK_ERR_UNSUPPORTED;   // the caller falls back to one chromosome at a time
    DevBuf dStage, dPsi, dK, dS, dW;
    QK_HIP(dStage.alloc(stage_bytes));
    QK_HIP(dPsi.alloc(psi_total));
    QK_HIP(dK.alloc(small ? k_live : k_max));
    QK_HIP(dS.alloc((size_t)cnt * QK_STATS_LEN * sizeof(double)));
    if (stats && !small) QK_HIP(dW.alloc((size_t)QK_STATS_WS * sizeof(double)));
    char* base = (char*)dStage.p;
    // ---- fill: blobs, rows, labels; jobs ordered by register class so that each class is one launch ----------
    std::vector<char> host(stage_bytes, 0);
    std::vector<int> order(cnt);
    for (int k = 0; k < cnt; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return circuits[idx[a]]->n_qubits < circuits[idx[b]]->n_qubits;     // equal qubit counts adjacent: same class, same D
    });
    WaveJob* jobs = (WaveJob*)(host.data() + jobs_off);
    EpiJob* epis = (EpiJob*)(host.data() + epi_off);
    for (int k = 0; k < cnt; ++k) {
        const int g = idx[k];
        const qk_circuit* c = circuits[g];
        std::memcpy(host.data() + blob_off[k], c->wave_blob.data(), c->wave_blob.size() * 8);
        if (c->n_params) std::memcpy(host.data() + x_off[k], X[g], (size_t)N[g] * c->n_params * 8);
        if (stats) std::memcpy(host.data() + l_off[k], labels[g], (size_t)N[g] * 4);
    }
    struct ClassRun { int first, count, nreg; int64_t units; size_t cs_bytes; };
    std::vector<ClassRun> runs;
    for (int p = 0; p < cnt; ++p) {
        const int k = order[p], g = idx[k];
        const qk_circuit* c = circuits[g];
        const int nreg = wave_nreg_class(c->n_qubits);
        if (runs.empty() || runs.back().nreg != nreg) runs.push_back(ClassRun{p, 0, nreg, 0, 0});
        ClassRun& r = runs.back();
        WaveJob& j = jobs[p];
        j.prog = base + blob_off[k];
        j.X = (const double*)(base + x_off[k]);
        j.ldx = c->n_params;
        j.N = N[g];
        j.psi = (char*)dPsi.p + psi_off[k];
        j.unit_begin = r.units;
        j.f32 = 0;
        j.prog_bytes = (int32_t)(c->wave_blob.size() * 8);
        r.units += (N[g] + c->wave_spw - 1) / c->wave_spw;
        r.cs_bytes = std::max(r.cs_bytes, wave_lds_bytes(*c));
        ++r.count;
        EpiJob& e = epis[k];
        e.K = (double*)((char*)dK.p + (small ? k_off[k] : 0));
        e.N = N[g];
        e.labels = stats ? (const int32_t*)(base + l_off[k]) : nullptr;
        e.stats = (double*)dS.p + (size_t)k * QK_STATS_LEN;
        e.seed = (opts ? opts->seed : 0) + (uint64_t)g;
    }
    QK_HIP(hipMemcpy(dStage.p, host.data(), stage_bytes, hipMemcpyHostToDevice));
    // ---- simulate: one launch per register class, every chromosome of the class in it --------------------------
    int rc;
    for (const ClassRun& r : runs)
        if ((rc = launch_wave_jobs(jobs + r.first, r.count, (const WaveJob*)(base + jobs_off) + r.first, r.nreg, r.units,
                                   r.cs_bytes, nullptr)))
            return rc;
    if (small) {
        // ---- Gram: 16 chromosomes of one dimension per launch -------------------------------------------------
        for (int p = 0; p < cnt;) {
            qk_gram_job gj[QK_GRAM_MAX_JOBS];
            std::memset(gj, 0, sizeof(gj));
            const int nq = circuits[idx[order[p]]]->n_qubits;
            int nj = 0;
            for (; p < cnt && nj < QK_GRAM_MAX_JOBS && circuits[idx[order[p]]]->n_qubits == nq; ++p, ++nj) {
                const int k = order[p];
                gj[nj].A = gj[nj].B = (char*)dPsi.p + psi_off[k];
                gj[nj].K = (double*)((char*)dK.p + k_off[k]);
                gj[nj].ma = gj[nj].nb = gj[nj].ldk = N[idx[k]];
                gj[nj].flags = QK_GRAM_SYMMETRIC;
            }
            if ((rc = qk_gram_jobs_dev(gj, nj, (int64_t)1 << nq, nullptr))) return rc;
        }
        // ---- noise + statistics of every chromosome: one launch -----------------------------------------------
        if (noise != QK_NOISE_NONE || stats)
            if ((rc = launch_batch_epilogue((const EpiJob*)(base + epi_off), cnt, noise, opts ? opts->noise_param : 0.0,
                                            stats != nullptr, nullptr)))
                return rc;
        if (K_host)
            for (int k = 0; k < cnt; ++k)
                if (K_host[idx[k]])
                    QK_HIP(hipMemcpy(K_host[idx[k]], (char*)dK.p + k_off[k], (size_t)N[idx[k]] * N[idx[k]] * 8, hipMemcpyDeviceToHost));
    } else {
        for (int k = 0; k < cnt; ++k) {
            const int g = idx[k];
            const int64_t n = N[g], D = (int64_t)1 << circuits[g]->n_qubits;
            double* Kg = (double*)dK.p;
            if ((rc = qk_gram_dev((char*)dPsi.p + psi_off[k], n, nullptr, n, D, Kg, n, nullptr))) return rc;
            if (noise != QK_NOISE_NONE &&
                (rc = qk_kernel_noise_dev(Kg, n, n, noise, opts->noise_param, opts->seed + (uint64_t)g, nullptr)))
                return rc;
            if (stats && (rc = qk_kernel_stats_dev(Kg, n, n, (const int32_t*)(base + l_off[k]),
                                                   (double*)dS.p + (size_t)k * QK_STATS_LEN, (double*)dW.p, nullptr)))
                return rc;
            // the next chromosome reuses the buffer: a blocking copy (or, with nothing to copy, stream order) keeps it safe
            if (K_host && K_host[g]) QK_HIP(hipMemcpy(K_host[g], Kg, (size_t)n * n * 8, hipMemcpyDeviceToHost));
        }
    }
    if (stats) {
        std::vector<double> hs((size_t)cnt * QK_STATS_LEN);
        QK_HIP(hipMemcpy(hs.data(), dS.p, hs.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int k = 0; k < cnt; ++k)
            std::memcpy(stats + (size_t)idx[k] * QK_STATS_LEN, hs.data() + (size_t)k * QK_STATS_LEN, QK_STATS_LEN * sizeof(double));
    }
    QK_HIP(hipDeviceSynchronize());
    return QK_OK;
}
}  // namespace
}  // namespace qk

extern "C" int qk_evaluate_batch(int count, const qk_circuit* const* circuits, const double* const* X, const int64_t* N,
                                 const int32_t* const* labels, const qk_batch_opts* opts, double* const* K_host,
                                 double* stats) {
    if (count < 0) return fail(QK_ERR_INVALID, "qk_evaluate_batch: negative count");
    if (count == 0) return QK_OK;
    if (!circuits || !N || (!K_host && !stats)) return fail(QK_ERR_INVALID, "qk_evaluate_batch: null pointer");
    const int noise = opts ? opts->noise_mode : QK_NOISE_NONE;
    if (noise != QK_NOISE_NONE && noise != QK_NOISE_GAUSSIAN && noise != QK_NOISE_SHOTS)
        return fail(QK_ERR_INVALID, "qk_evaluate_batch: unknown noise mode");
    if (noise == QK_NOISE_GAUSSIAN && !(opts->noise_param >= 0.0))
        return fail(QK_ERR_INVALID, "qk_evaluate_batch: negative standard deviation");
    if (noise == QK_NOISE_SHOTS && (!(opts->noise_param >= 1.0) || opts->noise_param > 1e9))
        return fail(QK_ERR_INVALID, "qk_evaluate_batch: shots must be in 1..1e9");
    std::vector<int> wave, other;
    for (int g = 0; g < count; ++g) {
        const qk_circuit* c = circuits[g];
        if (!c) return fail(QK_ERR_INVALID, "qk_evaluate_batch: null circuit");
        if (N[g] < 1) return fail(QK_ERR_INVALID, "qk_evaluate_batch: every genome needs at least one row");
        if (c->n_params > 0 && (!X || !X[g])) return fail(QK_ERR_INVALID, "qk_evaluate_batch: X is null");
        if (stats && (!labels || !labels[g])) return fail(QK_ERR_INVALID, "qk_evaluate_batch: statistics need labels");
        (c->wave_blob.empty() ? other : wave).push_back(g);
    }
    int rc = check_device();
    if (rc) return rc;
    if (!wave.empty()) {
        rc = batch_wave(wave, circuits, X, N, labels, noise, opts, K_host, stats);
        if (rc == QK_ERR_UNSUPPORTED) rc = batch_one_by_one(wave, circuits, X, N, labels, noise, opts, K_host, stats);
        if (rc) return rc;
    }
    if (!other.empty() && (rc = batch_one_by_one(other, circuits, X, N, labels, noise, opts, K_host, stats))) return rc;
    QK_HIP(hipDeviceSynchronize());
    return QK_OK;
}
