"""Quantum-kernel objects with the reference's API, computed by hand-written HIP.

* :class:`HipFidelityKernel` stands where the reference constructs
  ``qiskit_machine_learning.kernels.FidelityStatevectorKernel(feature_map=fmap)``
  (reference ``qsvm_tester.py:412``, ``genetic_v3.py:618``) and is consumed through
  ``kernel.evaluate(x_vec, y_vec=None)`` (``qsvm/train.py:92,152,154``).
* :func:`make_kernels` keeps the signature of reference ``qsvm_tester.py:382-413``.
* :func:`projected_quantum_kernel` keeps the signature of reference
  ``genetic_v3.py:766-806``.

All arithmetic (angle evaluation, statevector simulation, the N x M fidelity
contraction, reduced density matrices) runs on the GPU through
``libquake_hip.so``; there is no CPU path in this module.
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .circuit import Parameter, QuantumCircuit

logger = logging.getLogger("quake_amd.kernels")

# Above this many rows the optional PSD projection (an O(N^3) dense eigendecomposition
# on the host in the reference) is skipped with a warning.
PSD_MAX_ROWS = 4096


def _as_2d(a, name: str) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype.kind not in "fiu":
        raise ValueError(f"{name} must be numeric")
    a = np.ascontiguousarray(a, dtype=np.float64)  # float32 widens exactly
    if a.ndim > 2:
        raise ValueError(f"{name} must be a 1D or 2D array")
    if a.ndim == 1:
        a = a.reshape(1, -1)   # one sample, as qiskit_machine_learning's _validate_input
    if a.ndim == 0:
        a = a.reshape(1, 1)
    return a


def _pin_equal_rows(K: np.ndarray, x: np.ndarray, y: Optional[np.ndarray]) -> None:
    """The reference leaves ``K[i, j]`` at exactly 1.0 whenever the two statevectors compare equal
    (FidelityStatevectorKernel._evaluate) -- also off the diagonal (duplicate samples)
    and in cross blocks (a test row that is also a train row).  Equal input rows give equal
    statevectors (the simulator is deterministic), so those entries are pinned here; a circuit without
    parameters maps every row to the same state.  Costs one row sort; does nothing without duplicates."""
    if x.shape[1] == 0:
        K.fill(1.0)
        return
    # 64-bit hash of every row (wrap-around dot product of its bit patterns with fixed odd multipliers):
    # numeric sorts instead of row-wise comparisons; candidate pairs are verified bit for bit below
    mult = (np.arange(1, x.shape[1] + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    def h64(a):
        return (np.ascontiguousarray(a).view(np.uint64) * mult).sum(axis=1, dtype=np.uint64)
    hx = h64(x)
    n = hx.shape[0]
    if y is None:
        if np.unique(hx).size == n:
            return
        hy, yy = hx, x
    else:
        hy, yy = h64(y), y
    # all (i, j) with hx[i] == hy[j], without a Python loop: equal hashes are runs of the sorted hy
    order = np.argsort(hy, kind="stable")
    ys = hy[order]
    lo, hi = np.searchsorted(ys, hx, "left"), np.searchsorted(ys, hx, "right")
    cnt = hi - lo
    total = int(cnt.sum())
    if total == 0:
        return
    rows = np.repeat(np.arange(n), cnt)
    within = np.arange(total) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    cols = order[np.repeat(lo, cnt) + within]
    same = (x[rows].view(np.uint64) == yy[cols].view(np.uint64)).all(axis=1)
    K[rows[same], cols[same]] = 1.0


class HipFidelityKernel:
    """``K[i, j] = |<psi(x_i)|psi(y_j)>|^2`` for a parameterised feature map.

    Parameters mirror ``FidelityStatevectorKernel``:

    feature_map:  a :class:`quake_amd.circuit.QuantumCircuit`.  Data column k binds
                  the k-th element of ``feature_map.parameters`` (sorted order).
    enforce_psd:  the reference default projects the symmetric matrix onto the PSD
                  cone with a dense ``eig`` on the host.  K is PSD by construction
                  up to round-off (entries move by ~1e-15), the projection is
                  O(N^3), so it is off by default here and, when requested, only
                  applied up to ``PSD_MAX_ROWS`` rows.
    training_parameters / parameter_values: bind a subset of the circuit's
                  parameters (the ``align=True`` theta prefix) to fixed values, the
                  role ``TrainableFidelityQuantumKernel.assign_training_parameters``
                  plays in the reference (qsvm_tester.py:404-410).
    """

    def __init__(self, feature_map: QuantumCircuit, *, enforce_psd: bool = False,
                 training_parameters: Optional[Sequence[Parameter]] = None,
                 parameter_values: Optional[Sequence[float]] = None,
                 shots=None, cache_size=None, auto_clear_cache: bool = True, precision: str = "double",
                 seed: Optional[int] = None):
        if precision not in ("double", "single"):
            raise ValueError("precision must be 'double' or 'single'")
        # 'single': statevectors stored as complex64 and the Gram stage on FP32 MFMA (|dK| <= 1e-5); a
        # build option with no counterpart in the reference, whose kernel is always complex128
        self._precision = precision
        if shots is not None and int(shots) < 1:
            raise ValueError("shots must be a positive integer")
        self._shots = int(shots) if shots is not None else None
        # finite-shot draws: a counter-based generator keyed by (seed + call number, i, j).  The reference draws
        # from numpy's unseeded global stream, so two kernel objects are independent there; here each object
        # takes fresh entropy unless a seed is given (then its draws are reproducible)
        self._shot_seed = (int(seed) if seed is not None else int(np.random.SeedSequence().entropy)) & (2**64 - 1)
        self._shot_calls = 0
        self._feature_map = feature_map
        self._training_parameters = list(training_parameters) if training_parameters is not None else []
        self._enforce_psd = bool(enforce_psd)
        self._handle = None
        self._bound = None
        self.parameter_values = None
        if self._training_parameters:
            if parameter_values is None:
                parameter_values = np.zeros(len(self._training_parameters))
            self.assign_training_parameters(parameter_values)
        else:
            self._rebuild()

    # -- reference-visible attributes ------------------------------------------
    @property
    def feature_map(self) -> QuantumCircuit:
        return self._feature_map

    @property
    def training_parameters(self):
        return list(self._training_parameters)

    # old-API alias read by the reference at qsvm_tester.py:324,758
    user_parameters = training_parameters

    @property
    def num_features(self) -> int:
        return self._bound.num_parameters

    @property
    def enforce_psd(self) -> bool:
        return self._enforce_psd

    def assign_training_parameters(self, values) -> None:
        values = np.asarray(values, dtype=np.float64).reshape(-1)
        if len(values) != len(self._training_parameters):
            raise ValueError("wrong number of training-parameter values")
        self.parameter_values = values
        self._rebuild()

    def _rebuild(self) -> None:
        fm = self._feature_map
        if self._training_parameters:
            fm = fm.assign_parameters(dict(zip(self._training_parameters, self.parameter_values)))
        self._bound = fm
        self._compiled = fm.compile()
        self._handle = None  # created on first use: needs the shared library

    def _circuit(self) -> _lib.CircuitHandle:
        if self._handle is None:
            self._handle = _lib.CircuitHandle(self._compiled)
        return self._handle

    # -- the hot path -------------------------------------------------------------
    def _validate(self, x_vec, y_vec):
        x = _as_2d(x_vec, "x_vec")
        nf = self._compiled.n_params
        if x.shape[1] != nf:
            raise ValueError(
                f"x_vec and class feature map have incompatible dimensions.\n"
                f"x_vec has {x.shape[1]} dimensions, but feature map has {nf}.")
        y = None
        if y_vec is not None:
            y = _as_2d(y_vec, "y_vec")
            if y.shape[1] != nf:
                raise ValueError(
                    f"y_vec and class feature map have incompatible dimensions.\n"
                    f"y_vec has {y.shape[1]} dimensions, but feature map has {nf}.")
        return x, y

    def evaluate(self, x_vec, y_vec=None, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Gram matrix, ``float64 [len(x_vec), len(y_vec or x_vec)]``, C-contiguous.

        ``y_vec is None`` gives the symmetric matrix with an exact unit diagonal.
        ``out`` (an extension; the reference always returns a fresh array): a C-contiguous float64 array of the
        result's shape that receives K and is returned.  A caller that evaluates again and again (a learning
        curve, a GA) can keep ONE page-locked array from ``quake_amd._lib.host_empty`` and spare every call the
        page-locking of a fresh one -- 32 GiB at the headline size."""
        x, y = self._validate(x_vec, y_vec)
        if out is not None:
            want = (x.shape[0], x.shape[0] if y is None else y.shape[0])
            if not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.shape == want
                    and out.flags.c_contiguous and out.flags.writeable):
                raise ValueError(f"out must be a writable C-contiguous float64 array of shape {want}")
            if self._shots is not None or (y is None and self._enforce_psd):
                raise ValueError("out= is not supported with shots or enforce_psd")
        lib = _lib.load()
        circ = self._circuit()
        n, m = x.shape[0], (x.shape[0] if y is None else y.shape[0])
        if self._shots is not None:
            # FidelityStatevectorKernel(shots=...): each entry becomes Binomial(shots, K_ij) / shots, drawn on the
            # device behind the Gram launches (qk_evaluate_noisy): symmetric draws with an exact diagonal for
            # evaluate(x), independent draws for every entry of a cross block
            self._shot_calls += 1
            seed = (self._shot_seed + self._shot_calls) & (2**64 - 1)
            K = _lib.host_empty((n, m))
            _lib.check(lib.qk_evaluate_noisy(
                circ.handle, _lib.ptr(x), n, _lib.ptr(y) if y is not None else None, m, x.shape[1],
                0 if self._precision == "double" else 1, _lib.NOISE_SHOTS, float(self._shots), seed, _lib.ptr(K)))
            return K
        K = out if out is not None else _lib.host_empty((n, m))
        _lib.check((lib.qk_evaluate if self._precision == "double" else lib.qk_evaluate_f32)(
            circ.handle, _lib.ptr(x), n,
            _lib.ptr(y) if y is not None else None, m, x.shape[1], _lib.ptr(K)))
        _pin_equal_rows(K, x, y)
        if y is None and self._enforce_psd:
            K = self._make_psd(K)
        return K

    def statevectors(self, x_vec) -> np.ndarray:
        """``complex128 [N, 2^n]`` statevectors (``Statevector(fmap.assign_parameters(x)).data``)."""
        x, _ = self._validate(x_vec, None)
        lib = _lib.load()
        circ = self._circuit()
        psi = _lib.host_empty((x.shape[0], 1 << circ.n_qubits), np.complex128)
        _lib.check(lib.qk_simulate(circ.handle, _lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(psi)))
        return psi

    @staticmethod
    def _make_psd(K: np.ndarray) -> np.ndarray:
        if K.shape[0] > PSD_MAX_ROWS:
            logger.warning("enforce_psd skipped: %d rows > %d (K is PSD up to round-off)", K.shape[0], PSD_MAX_ROWS)
            return K
        w, v = np.linalg.eig(K)
        return np.ascontiguousarray((v @ np.diag(np.maximum(0, w)) @ v.transpose()).real)


def make_kernels(maps: List[QuantumCircuit], backend=None) -> List[HipFidelityKernel]:
    """Quantum kernels for a list of feature maps (reference qsvm_tester.py:382-413).

    ``backend`` is accepted and ignored, as in the reference's statevector branch.
    A map with more parameters than qubits (the ``align=True`` theta prefix) gets its
    trailing parameters as training parameters, like the reference's
    ``TrainableFidelityQuantumKernel`` branch (initialised to zero)."""
    kernels = []
    for fmap in maps:
        params = fmap.parameters
        if len(params) > fmap.num_qubits:
            theta = params[fmap.num_qubits:]
            kernels.append(HipFidelityKernel(feature_map=fmap, training_parameters=theta))
        else:
            kernels.append(HipFidelityKernel(feature_map=fmap))
    return kernels


def projected_quantum_kernel(fmap: QuantumCircuit, dataset: np.ndarray, gamma: float) -> np.ndarray:
    """1-RDM projected quantum kernel (reference genetic_v3.py:766-806):
    ``K_ij = exp(-gamma * sum_q ||rho_i^q - rho_j^q||_F)``, unit diagonal; all ones
    for a circuit without parameters."""
    X = _as_2d(dataset, "dataset")
    N = X.shape[0]
    if not fmap.parameters:
        return np.ones((N, N))
    compiled = fmap.compile()
    if X.shape[1] != compiled.n_params:
        raise ValueError(
            f"Mismatching number of values and parameters: dataset has {X.shape[1]} columns, "
            f"feature map has {compiled.n_params} parameters.")
    lib = _lib.load()
    circ = _lib.CircuitHandle(compiled)
    K = _lib.host_empty((N, N))
    _lib.check(lib.qk_projected_kernel(circ.handle, _lib.ptr(X), N, X.shape[1], float(gamma), _lib.ptr(K)))
    return K


def statistics_from_sums(sums: np.ndarray, N: int) -> dict:
    """The reference's kernel statistics (genetic_v3.py:643-660) from the raw sums of
    ``qk_kernel_stats_dev``: class-block means over the strict upper triangles, the cross-block
    mean, and mean / std of ``np.triu(K, 1)`` over the whole N x N array (zeros included)."""
    s, ss, s00, s11, s01, n0, n1 = (float(v) for v in sums[:7])
    with np.errstate(divide="ignore", invalid="ignore"):
        m00 = np.float64(s00) / (n0 * (n0 - 1) / 2)
        m11 = np.float64(s11) / (n1 * (n1 - 1) / 2)
        m01 = np.float64(s01) / (n0 * n1)
    mean = s / (N * N)
    return {"sparsity": (m00 + m11) / 2 - m01,
            "offdiagonal_mean": mean,
            "offdiagonal_std": float(np.sqrt(max(ss / (N * N) - mean * mean, 0.0)))}


def evaluate_population(feature_maps: Sequence[QuantumCircuit], x_vecs: Sequence[np.ndarray], labels=None,
                        noise: Optional[tuple] = None, seed: int = 0, return_kernels: bool = True):
    """Symmetric fidelity kernels of many (circuit, data) pairs in ONE library call: the GA's
    generation (reference genetic_v3.py:587-661 runs them one by one).

    feature_maps / x_vecs: one circuit and one ``[N_g, F_g]`` array per genome (sizes may differ).
    labels: 0/1 labels (one array for all genomes, or one per genome) -> kernel statistics are
            reduced on the device and returned as dicts (``statistics_from_sums``).
    noise:  None, ``("gaussian", std)`` or ``("shots", shots)``, applied on the device before the
            statistics (genome g uses ``seed + g``).
    Returns ``(kernels or None, stats or None)``."""
    count = len(feature_maps)
    if len(x_vecs) != count:
        raise ValueError("one data array per feature map")
    if count == 0:
        return ([] if return_kernels else None), ([] if labels is not None else None)
    handles, xs = [], []
    for fmap, x in zip(feature_maps, x_vecs):
        h = _lib.CircuitHandle(fmap.compile())
        x = _as_2d(x, "x_vec")
        if x.shape[1] != fmap.num_parameters:
            raise ValueError(f"x_vec and class feature map have incompatible dimensions.\n"
                             f"x_vec has {x.shape[1]} dimensions, but feature map has {fmap.num_parameters}.")
        if x.shape[0] < 1:
            raise ValueError("every genome needs at least one data row")
        handles.append(h)
        xs.append(x)
    lab = None
    if labels is not None:
        per_genome = isinstance(labels, (list, tuple)) and len(labels) == count and np.ndim(labels[0]) == 1
        if not per_genome:
            labels = [labels] * count
        lab = [np.ascontiguousarray(l, dtype=np.int32) for l in labels]
        for l, x in zip(lab, xs):
            if l.shape != (x.shape[0],):
                raise ValueError("labels must have one entry per data row")
    lib = _lib.load()
    P = C.c_void_p
    circ_arr = (P * count)(*[h.handle for h in handles])
    x_arr = (P * count)(*[x.ctypes.data if x.size else None for x in xs])
    n_arr = (C.c_int64 * count)(*[x.shape[0] for x in xs])
    lab_arr = (P * count)(*[l.ctypes.data for l in lab]) if lab is not None else None
    Ks = [_lib.host_empty((x.shape[0], x.shape[0])) for x in xs] if return_kernels else None
    k_arr = (P * count)(*[k.ctypes.data for k in Ks]) if Ks is not None else None
    sums = np.zeros((count, _lib.STATS_LEN), dtype=np.float64) if lab is not None else None
    if Ks is None and sums is None:
        raise ValueError("nothing to compute: return_kernels=False needs labels")
    opts = None
    if noise is not None:
        mode, param = noise
        code = {"gaussian": _lib.NOISE_GAUSSIAN, "shots": _lib.NOISE_SHOTS}[mode]
        opts = C.pointer(_lib.qk_batch_opts(code, 0, float(param), int(seed) & (2**64 - 1)))
    _lib.check(lib.qk_evaluate_batch(count, circ_arr, x_arr, n_arr, lab_arr, opts, k_arr,
                                     sums.ctypes.data if sums is not None else None))
    if Ks is not None and noise is None:
        for K, x in zip(Ks, xs):          # duplicate samples: exactly 1.0, as evaluate() and the reference have it
            _pin_equal_rows(K, x, None)
    stats = [statistics_from_sums(sums[g], xs[g].shape[0]) for g in range(count)] if sums is not None else None
    return Ks, stats
