/*
 * quake_hip.h -- C ABI of libquake_hip.so, the MI355X (gfx950) quantum-kernel engine.
 *
 * The reference (qismib/QuaKe) has no FFI: its seam for this path is a Python
 * duck-type whose arithmetic lives in qiskit / qiskit_machine_learning.  Each
 * entry point below names the reference call it replaces.  The Python wrapper in
 * quake_amd/kernels.py binds these with ctypes.
 *
 * Conventions
 *   - every function returns 0 on success or a negative qk_status; the message is
 *     read with qk_last_error() (thread-local);
 *   - the caller owns every buffer it passes; the library owns qk_circuit handles;
 *   - "host" entry points take host pointers, allocate device memory themselves,
 *     and are synchronous; "_dev" entry points take device pointers (hipMalloc'ed
 *     memory, e.g. a torch tensor's data_ptr()) and a hipStream_t passed as void*,
 *     and only enqueue work on that stream;
 *   - statevectors Psi are row-major [N][2^n] interleaved complex128 (re, im);
 *     Gram matrices are row-major float64;
 *   - qubit 0 is the least significant bit of the amplitude index.
 */
#ifndef QUAKE_HIP_H
#define QUAKE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum qk_status {
    QK_OK = 0,
    QK_ERR_INVALID = -1,     /* bad argument (shape, null pointer, unsupported gate ...) */
    QK_ERR_HIP = -2,         /* a HIP runtime call failed */
    QK_ERR_NO_DEVICE = -3,   /* no gfx950 device visible */
    QK_ERR_UNSUPPORTED = -4  /* valid request outside what this build implements */
} qk_status;

/* Gate alphabet reachable on the reference's hot path. */
typedef enum qk_gate {
    QK_ID = 0, QK_H, QK_X, QK_SX, QK_RX, QK_RY, QK_RZ, QK_P, QK_CX, QK_CZ, QK_ECR
} qk_gate;

/* Postfix ops of a gate-angle program, evaluated in IEEE double on the device. */
typedef enum qk_expr_op {
    QK_EX_CONST = 0, /* push consts[arg]            */
    QK_EX_PARAM,     /* push x_row[arg]             */
    QK_EX_ADD, QK_EX_SUB, QK_EX_MUL, QK_EX_DIV, QK_EX_NEG,
    QK_EX_ASIN, QK_EX_ACOS, QK_EX_SIN, QK_EX_COS
} qk_expr_op;

typedef struct qk_op {     /* one gate */
    int32_t gate;          /* qk_gate */
    int32_t q0;            /* target (1-qubit gates); first argument (cx: control) */
    int32_t q1;            /* second argument (cx: target) or -1 */
    int32_t expr;          /* index of the angle program, or -1 */
} qk_op;

typedef struct qk_instr {  /* one postfix instruction */
    int32_t op;            /* qk_expr_op */
    int32_t arg;
} qk_instr;

typedef struct qk_circuit qk_circuit; /* opaque */

int qk_version(void);
int qk_device_count(void);
/* Copies the calling thread's last error message into buf (NUL-terminated). */
int qk_last_error(char* buf, size_t len);

/*
 * Page-locked host memory for the buffers the host entry points fill (K, Psi).  The reference hands
 * back a fresh numpy array per evaluate() call ("fresh array owned by the caller");
 * an array that lives in memory from qk_host_alloc receives K at the PCIe rate (measured 53 GiB/s
 * against 9 GiB/s into fresh pageable pages).  Optional: every host entry point accepts any host
 * pointer.  The caller releases the memory with qk_host_free.
 */
int qk_host_alloc(size_t bytes, void** out);
int qk_host_free(void* p);

/*
 * No counterpart in the reference (its K is born in host memory): the "K block gathered to host" leg of
 * the multi-GPU path (BASELINE.json north_star).  Enqueues the copy of a rows x cols float64 block of a
 * device matrix (row stride ldk elements) into host memory (row stride ldh) on `stream` and returns;
 * with K_host in page-locked memory (qk_host_alloc, or a pinned torch tensor) the copy overlaps kernels
 * and exchanges on other streams.  The caller synchronises the stream before reading K_host.
 */
int qk_block_to_host_async(const double* K_dev, int64_t ldk, int64_t rows, int64_t cols,
                           double* K_host, int64_t ldh, void* stream);

/*
 * Replaces: building a qiskit QuantumCircuit over a ParameterVector
 * (reference quantum_featuremaps.py:9-387, genetic_v3.py:363-448) plus
 * QuantumCircuit.assign_parameters: angle program e reads data
 * column `arg` of QK_EX_PARAM directly, so the sorted-parameter binding is resolved
 * by the caller when it numbers the columns.
 *   expr_off has n_expr + 1 entries; program e is prog[expr_off[e] .. expr_off[e+1]).
 * The planner rewrites the gate list by exact identities only (runs of single-qubit gates multiplied into one
 * matrix, cx / cz / ecr on qubits that are still basis states, self-inverse pairs: DESIGN.md 4.2p); what the
 * simulators return equals the gate-by-gate product up to rounding (1e-12 in the parity tests, 1e-15 typical).
 */
int qk_circuit_create(int n_qubits, int n_params,
                      const qk_op* ops, int n_ops,
                      const qk_instr* prog, int n_instr,
                      const int32_t* expr_off, int n_expr,
                      const double* consts, int n_consts,
                      qk_circuit** out);
void qk_circuit_destroy(qk_circuit* c);
int qk_circuit_num_qubits(const qk_circuit* c);
int qk_circuit_num_params(const qk_circuit* c);
/* Number of HBM passes the simulator plans for this circuit (1 when 2^n amplitudes fit in LDS). */
int qk_circuit_num_passes(const qk_circuit* c);

/*
 * Replaces: [Statevector(fmap.assign_parameters(x)).data for x in x_vec]
 * (qiskit_machine_learning FidelityStatevectorKernel._evaluate; call sites
 * reference qsvm/train.py:92,152,154, genetic_v3.py:618-619).
 *   X    : N x n_params float64, row stride ldx (elements)
 *   psi  : N x 2^n complex128, written
 * Temporary tables (the gate angles' cosines and sines, the phase tables of diagonal stages: up to 1 GiB, twice
 * that when they alternate) come from the device's default memory pool in stream order (hipMallocAsync /
 * hipFreeAsync on `stream`): nothing outlives the call's work, no synchronisation.  Large calls on narrow
 * registers (2^25 amplitudes or more, tables of 3/8 of Psi or more: DESIGN.md 4.2r) also launch their table
 * kernels on a stream the library owns (one per GPU, created on the first such call), joined to `stream` by
 * events on both sides: everything the call enqueues is still ordered behind what `stream` held before the
 * call, and whatever the caller enqueues on `stream` afterwards is ordered behind all of the call's work.
 */
int qk_simulate_dev(const qk_circuit* c, const double* X_dev, int64_t N, int64_t ldx,
                    void* psi_dev, void* stream);
int qk_simulate(const qk_circuit* c, const double* X, int64_t N, int64_t ldx, void* psi_host);

/*
 * Replaces: FidelityStatevectorKernel._compute_fidelity over all pairs, i.e.
 * K[i][j] = |sum_b conj(A[i][b]) * B[j][b]|^2  (SURVEY.md 8a A5).
 *
 * One job computes a block K = |A B^H|^2 with A (ma x D) and B (nb x D):
 *   K [i*ldk  + j] = k_ij                    (always)
 *   KT[j*ldkt + i] = k_ij                    (if KT != NULL; transposed copy, used to
 *                                             hand mirror blocks to another GPU)
 * flags & QK_GRAM_SYMMETRIC: A and B are the same rows (ma == nb); only tiles on or
 *   above the diagonal are computed, the mirror is written into K itself and the
 *   diagonal is set to exactly 1.0 (the reference leaves those entries at 1.0).
 * All jobs of one call run in ONE kernel launch (one tile schedule over all jobs).
 */
#define QK_GRAM_SYMMETRIC 1
typedef struct qk_gram_job {
    const void* A;    /* device, ma x D complex128 */
    const void* B;    /* device, nb x D complex128 */
    double* K;        /* device */
    double* KT;       /* device or NULL */
    int64_t ma, nb;
    int64_t ldk, ldkt;
    int32_t flags;
    int32_t reserved;
} qk_gram_job;
#define QK_GRAM_MAX_JOBS 16
int qk_gram_jobs_dev(const qk_gram_job* jobs, int n_jobs, int64_t D, void* stream);
/* Convenience over one job: psi_y == NULL means the symmetric train x train block. */
int qk_gram_dev(const void* psi_x, int64_t N, const void* psi_y, int64_t M, int64_t D,
                double* K_dev, int64_t ldk, void* stream);

/*
 * FP32 mode (optional precision of BASELINE.json's north_star; no counterpart in the reference,
 * whose FidelityStatevectorKernel is always complex128 -- SURVEY.md 8a A4).  The circuit is
 * evolved in FP64 and rounded once when Psi is stored as complex64 [N][2^n] (re, im float32);
 * the Gram stage runs on FP32 MFMA with two-level accumulation and still delivers float64 K.
 * Tolerance: |K - K_fp64| <= 1e-5.  psi_ws_dev: complex128 workspace N x 2^n, needed only when
 * qk_circuit_num_passes(c) > 1 (NULL otherwise).  Jobs are those of qk_gram_jobs_dev with A, B
 * pointing at complex64 rows.
 */
int qk_simulate_c64_dev(const qk_circuit* c, const double* X_dev, int64_t N, int64_t ldx,
                        void* psi32_dev, void* psi_ws_dev, void* stream);
int qk_gram32_jobs_dev(const qk_gram_job* jobs, int n_jobs, int64_t D, void* stream);
int qk_gram32_dev(const void* psi32_x, int64_t N, const void* psi32_y, int64_t M, int64_t D,
                  double* K_dev, int64_t ldk, void* stream);
/* Host-pointer convenience: qk_evaluate with the statevectors and the Gram stage in FP32 mode. */
int qk_evaluate_f32(const qk_circuit* c, const double* X, int64_t N,
                    const double* Y, int64_t M, int n_features, double* K_host);

/*
 * Replaces: kernel.evaluate(x_vec, y_vec=None) of FidelityStatevectorKernel
 * (reference qsvm/train.py:92,152,154; qsvm_tester.py:684-686; genetic_v3.py:619).
 * Host pointers; Y == NULL gives the symmetric N x N matrix with unit diagonal.
 * n_features must equal the circuit's parameter count (else QK_ERR_INVALID, the
 * wrapper turns it into the reference's ValueError).
 */
int qk_evaluate(const qk_circuit* c, const double* X, int64_t N,
                const double* Y, int64_t M, int n_features, double* K_host);

/*
 * Replaces: DensityMatrix(...) + partial_trace(dm, all-but-q) for every qubit
 * (reference genetic_v3.py:795-803).  rdm is N x n x 4 float64:
 * (rho00, rho11, Re rho01, Im rho01) of qubit q.
 */
int qk_rdm1_dev(const void* psi_dev, int64_t N, int n_qubits, double* rdm_dev, void* stream);

/*
 * Replaces: the same two steps of projected_quantum_kernel (reference genetic_v3.py:795-803:
 * assign_parameters -> DensityMatrix -> partial_trace per qubit) without ever writing the
 * statevectors: when the circuit runs as one pass over a state that stays on the chip
 * (qk_simulate_rdm1_is_fused: up to 13 qubits), the reduced density matrices are reduced from LDS
 * at the end of the simulation.  Otherwise psi_ws_dev (N x 2^n complex128 workspace) is required and
 * the call is qk_simulate_dev + qk_rdm1_dev.
 */
int qk_simulate_rdm1_is_fused(const qk_circuit* c);
/*
 * No counterpart in the reference (a scheduling hint of this library).  1 when a call over N samples should be given the
 * N x 2^n complex128 workspace: the circuit cannot run fused, or the two-step route is the faster one for this circuit
 * and batch (circuits the register-resident simulator runs much faster than the LDS simulator: GA genomes, maps with
 * layers of rotations; at most 8 GiB of workspace).  qk_simulate_rdm1_dev takes the two-step route exactly when this
 * returns 1 AND the workspace is passed; with a null workspace a circuit that can run fused still does.
 */
int qk_simulate_rdm1_wants_workspace(const qk_circuit* c, int64_t N);
int qk_simulate_rdm1_dev(const qk_circuit* c, const double* X_dev, int64_t N, int64_t ldx,
                         double* rdm_dev, void* psi_ws_dev, void* stream);

/*
 * Replaces: the pair loop of projected_quantum_kernel (reference genetic_v3.py:793-805):
 * K[i][j] = exp(-gamma * sum_q ||rho_i^q - rho_j^q||_F)  (un-squared Frobenius norm).
 * rdm_y == NULL: symmetric N x N with exact unit diagonal.
 */
int qk_pqk_dev(const double* rdm_x, int64_t N, const double* rdm_y, int64_t M, int n_qubits,
               double gamma, double* K_dev, int64_t ldk, void* stream);

/*
 * Replaces: projected_quantum_kernel(fmap, dataset, gamma) (reference genetic_v3.py:766-806).
 * Host pointers.  A circuit without parameters yields all ones (reference :789-791).
 */
int qk_projected_kernel(const qk_circuit* c, const double* X, int64_t N, int n_features,
                        double gamma, double* K_host);

/*
 * Replaces: the kernel-statistics block of the GA fitness function (reference
 * genetic_v3.py:643-660) on a symmetric K held on the device.  labels: N int32 (0 / 1) or
 * NULL.  stats (device, QK_STATS_LEN float64) receives raw sums over the strict upper
 * triangle i < j -- the caller forms the reference's means from them:
 *   [0] sum K_ij            [1] sum K_ij^2
 *   [2] sum over pairs with both labels 0     [3] ... both labels 1
 *   [4] sum over pairs with different labels  (K symmetric: = sum of K[labels==0][:, labels==1])
 *   [5] number of label-0 rows   [6] number of label-1 rows   [7] 0
 * offdiagonal_mean = [0]/N^2 and offdiagonal_std = sqrt([1]/N^2 - mean^2): the reference takes
 * both over the WHOLE N x N array np.triu(K, 1), zeros included (genetic_v3.py:659-660).
 * Deterministic (fixed reduction tree, no atomics).  workspace: device, QK_STATS_WS float64.
 */
#define QK_STATS_LEN 8
#define QK_STATS_WS (1024 * 8)
int qk_kernel_stats_dev(const double* K_dev, int64_t N, int64_t ldk, const int32_t* labels_dev,
                        double* stats_dev, double* workspace_dev, void* stream);

/*
 * No counterpart in the reference.  For callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops such as a GA fitness loop, harnesses that forbid hipMalloc): the same
 * statistics as qk_kernel_stats_dev, bitwise, with the partial sums kept in a static device array inside
 * the library (QK_STATS_WS float64 = 64 KiB per GPU, held until the process ends) instead of a caller
 * workspace.  Calls no hipMalloc / hipFree.  N above QK_STATS_NOALLOC_MAX_N is refused
 * (QK_ERR_UNSUPPORTED).  All calls on one GPU share that array: calls that overlap on one GPU (two
 * streams, two threads) race -- serialise them, or use qk_kernel_stats_dev with a workspace per stream.
 * The GPU that K lives on must be the current device (QK_ERR_INVALID otherwise: each GPU has its own copy of
 * the array).  The first call on a device looks the array's address up (hipGetSymbolAddress): make that call
 * outside stream capture, e.g. once at start-up; later calls only launch kernels.
 */
#define QK_STATS_NOALLOC_MAX_N 131072
int qk_kernel_stats_noalloc_dev(const double* K_dev, int64_t N, int64_t ldk, const int32_t* labels_dev,
                                double* stats_dev, void* stream);

/*
 * Replaces: the noise models the reference puts on a statevector kernel matrix, applied in
 * place to a symmetric device K (upper triangle drawn, mirrored, diagonal untouched):
 *   QK_NOISE_GAUSSIAN  K_ij <- clip(K_ij + N(0, param), 0, 1)        (genetic_v3.py:620-627)
 *   QK_NOISE_SHOTS     K_ij <- Binomial(param shots, K_ij) / shots   (P(0...0) estimate of the QPU runs,
 *                      genetic_v4.py:451-459, 792: shots=8000; FidelityStatevectorKernel(shots=...))
 * Counter-based generator (Philox4x32-10) keyed by seed and (i, j): reproducible, independent of
 * launch geometry.  The reference draws from numpy's unseeded global stream, so parity is
 * statistical (tests check moments), not bitwise.
 */
#define QK_NOISE_NONE 0
#define QK_NOISE_GAUSSIAN 1
#define QK_NOISE_SHOTS 2
int qk_kernel_noise_dev(double* K_dev, int64_t N, int64_t ldk, int mode, double param, uint64_t seed,
                        void* stream);
/* The same models on a rectangular rows x cols block (a test-versus-train kernel: every entry drawn,
 * no mirror), entry (i, j) keyed by (i, j) as above. */
int qk_kernel_noise_rect_dev(double* K_dev, int64_t rows, int64_t cols, int64_t ldk, int mode, double param,
                             uint64_t seed, void* stream);
/*
 * Replaces: kernel.evaluate(x_vec, y_vec=None) of FidelityStatevectorKernel(shots=...) (the finite-shot
 * kernel qiskit-machine-learning samples from the exact fidelities) and the GA's noisy statevector fitness
 * (reference genetic_v3.py:619-627).  qk_evaluate (f32 != 0: qk_evaluate_f32) with the noise model applied on
 * the device, band by band behind the Gram launches, before K leaves the GPU.  Y == NULL: symmetric draws,
 * unit diagonal; otherwise every entry of the N x M block is drawn independently.
 */
int qk_evaluate_noisy(const qk_circuit* c, const double* X, int64_t N, const double* Y, int64_t M,
                      int n_features, int f32, int mode, double param, uint64_t seed, double* K_host);

/*
 * Replaces: one GA generation's fitness evaluations (pygad calls fitness_func once per
 * chromosome: reference genetic_v3.py:587-661) as ONE heterogeneous batch.  Genome g has its own
 * circuit (any qubit / parameter count), its own feature matrix X[g] (N[g] x n_params(g), dense)
 * and optionally labels[g] (N[g] int32).  For each genome: simulate, symmetric fidelity Gram,
 * optional noise (seed + g), optional statistics -- all enqueued back to back on one stream with
 * a single synchronisation at the end.  Host pointers.
 *   K_host : count pointers (entries may be NULL: statistics only), N[g] x N[g] float64 each
 *   stats  : NULL or count x QK_STATS_LEN float64 (needs labels)
 */
typedef struct qk_batch_opts {
    int32_t noise_mode;     /* QK_NOISE_* */
    int32_t reserved;
    double noise_param;
    uint64_t seed;
} qk_batch_opts;
int qk_evaluate_batch(int count, const qk_circuit* const* circuits, const double* const* X,
                      const int64_t* N, const int32_t* const* labels, const qk_batch_opts* opts,
                      double* const* K_host, double* stats);

#ifdef __cplusplus
}
#endif
#endif /* QUAKE_HIP_H */
