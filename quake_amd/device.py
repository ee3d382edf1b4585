"""Device-resident entry points: torch tensors in HBM in, torch tensors out.

torch is plumbing here (device memory, the current HIP stream, torch.distributed);
every kernel is launched through the ``*_dev`` functions of libquake_hip.so on
torch's current stream.  Used by ``bench.py`` and :mod:`quake_amd.sharded`.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib
from .circuit import QuantumCircuit


def _stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dp(t: Optional[torch.Tensor]) -> Optional[C.c_void_p]:
    return None if t is None else C.c_void_p(t.data_ptr())


def from_dlpack(x) -> Optional[torch.Tensor]:
    """DLPack hand-off at the device boundary (BASELINE.json north_star: "DLPack/numpy hand-off"): anything that
    implements ``__dlpack__`` (a cupy / jax / numba array, another framework's tensor) or a raw DLPack capsule is
    viewed -- zero copy -- as the torch tensor the launchers work on; torch tensors pass through.  Every tensor
    this module returns implements ``__dlpack__`` itself, so results travel back the same way
    (``consumer.from_dlpack(K)``)."""
    if x is None or isinstance(x, torch.Tensor):
        return x
    if hasattr(x, "__dlpack__"):
        return torch.from_dlpack(x)
    if type(x).__name__ == "PyCapsule":
        return torch.utils.dlpack.from_dlpack(x)
    raise TypeError(f"expected a torch tensor, an object with __dlpack__ or a DLPack capsule, got {type(x).__name__}")


def to_dlpack(t: torch.Tensor):
    """A DLPack capsule of a result tensor (for consumers that take capsules rather than ``__dlpack__`` objects)."""
    return torch.utils.dlpack.to_dlpack(t)


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise ValueError("device entry points need tensors in GPU memory")


class DeviceCircuit:
    """A feature map compiled once and kept as a ``qk_circuit`` handle."""

    def __init__(self, feature_map: QuantumCircuit):
        self.feature_map = feature_map
        self.compiled = feature_map.compile()
        self.handle = _lib.CircuitHandle(self.compiled)
        self.n_qubits = self.compiled.n_qubits
        self.n_params = self.compiled.n_params
        self.dim = 1 << self.n_qubits

    def simulate(self, X: torch.Tensor, out: Optional[torch.Tensor] = None, dtype=torch.complex128) -> torch.Tensor:
        """X float64 [N, n_params] (row stride free) -> Psi [N, 2^n], complex128 or (FP32 mode:
        evolved in FP64, rounded at the store) complex64 -- the dtype of ``out`` if given."""
        X, out = from_dlpack(X), from_dlpack(out)
        _require_gpu(X, out)
        if X.dtype != torch.float64 or X.dim() != 2 or X.stride(1) != 1:
            raise ValueError("X must be a float64 [N, F] tensor with unit column stride")
        if X.shape[1] != self.n_params:
            raise ValueError(
                f"x_vec and class feature map have incompatible dimensions.\n"
                f"x_vec has {X.shape[1]} dimensions, but feature map has {self.n_params}.")
        N = X.shape[0]
        if out is None:
            out = torch.empty((N, self.dim), dtype=dtype, device=X.device)
        elif out.dtype not in (torch.complex128, torch.complex64) or tuple(out.shape) != (N, self.dim) \
                or not out.is_contiguous():
            raise ValueError("out must be a contiguous complex128 / complex64 [N, 2^n] tensor")
        ldx = X.stride(0) if N > 1 else max(self.n_params, 1)
        lib = _lib.load()
        if out.dtype == torch.complex128:
            _lib.check(lib.qk_simulate_dev(self.handle.handle, _dp(X), N, ldx, _dp(out), _stream_ptr()))
        else:
            ws = None
            if self.handle.num_passes > 1:     # earlier passes go through a complex128 workspace
                ws = torch.empty((N, self.dim), dtype=torch.complex128, device=X.device)
            _lib.check(lib.qk_simulate_c64_dev(self.handle.handle, _dp(X), N, ldx, _dp(out), _dp(ws), _stream_ptr()))
        return out


def simulate_rdm1(circuit: "DeviceCircuit", X: torch.Tensor) -> torch.Tensor:
    """Per-qubit reduced density matrices of the encoded states, float64 [N, n, 4], without writing
    the statevectors when the circuit fits one LDS-resident pass (``qk_simulate_rdm1_dev``)."""
    X = from_dlpack(X)
    _require_gpu(X)
    lib = _lib.load()
    if X.dtype != torch.float64 or X.dim() != 2 or X.shape[1] != circuit.n_params or (X.numel() and X.stride(1) != 1):
        raise ValueError("X must be float64 [N, n_params] with unit column stride")
    N, n = X.shape[0], circuit.n_qubits
    out = torch.empty((N, n, 4), dtype=torch.float64, device=X.device)
    ws = None
    if lib.qk_simulate_rdm1_wants_workspace(circuit.handle.handle, N):      # not fused, or the two-step route is the faster one
        ws = torch.empty((N, 1 << n), dtype=torch.complex128, device=X.device)
    ldx = X.stride(0) if (N > 1 and X.shape[1]) else max(X.shape[1], 1)
    _lib.check(lib.qk_simulate_rdm1_dev(circuit.handle.handle, _dp(X) if X.numel() else None, N, ldx, _dp(out), _dp(ws),
                                        _stream_ptr()))
    return out


def gram_jobs(jobs: Sequence[dict], D: int) -> None:
    """Run several block jobs in one launch.  Each job: dict(A=, B=, K=, KT=None, symmetric=False);
    A, B complex128 (or, for the FP32 Gram stage, complex64) [rows, D] contiguous; K (and KT) float64
    2-D views with unit column stride."""
    jobs = [dict(j, A=from_dlpack(j["A"]), B=from_dlpack(j["B"]), K=from_dlpack(j["K"]), KT=from_dlpack(j.get("KT")))
            for j in jobs]
    arr = (_lib.qk_gram_job * len(jobs))()
    dtype = jobs[0]["A"].dtype if jobs else torch.complex128
    if dtype not in (torch.complex128, torch.complex64):
        raise ValueError("statevectors must be complex128 or complex64")
    for k, j in enumerate(jobs):
        A, B, K, KT = j["A"], j["B"], j["K"], j.get("KT")
        _require_gpu(A, B, K, KT)
        for t in (A, B):
            if t.dtype != dtype or t.dim() != 2 or t.shape[1] != D or not t.is_contiguous():
                raise ValueError("A and B must be contiguous [rows, D] of one complex dtype")
        if K.dtype != torch.float64 or K.stride(1) != 1 or tuple(K.shape) != (A.shape[0], B.shape[0]):
            raise ValueError("K must be float64 [rows(A), rows(B)] with unit column stride")
        if KT is not None and (KT.dtype != torch.float64 or KT.stride(1) != 1
                               or tuple(KT.shape) != (B.shape[0], A.shape[0])):
            raise ValueError("KT must be float64 [rows(B), rows(A)] with unit column stride")
        arr[k].A, arr[k].B = A.data_ptr(), B.data_ptr()
        arr[k].K = K.data_ptr()
        arr[k].KT = KT.data_ptr() if KT is not None else None
        arr[k].ma, arr[k].nb = A.shape[0], B.shape[0]
        arr[k].ldk = K.stride(0) if K.shape[0] > 1 else max(K.shape[1], 1)
        arr[k].ldkt = (KT.stride(0) if KT.shape[0] > 1 else max(KT.shape[1], 1)) if KT is not None else 0
        arr[k].flags = _lib.QK_GRAM_SYMMETRIC if j.get("symmetric") else 0
    fn = _lib.load().qk_gram_jobs_dev if dtype == torch.complex128 else _lib.load().qk_gram32_jobs_dev
    _lib.check(fn(arr, len(jobs), D, _stream_ptr()))


def gram(psi_x: torch.Tensor, psi_y: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K = |Psi_x Psi_y^H|^2; ``psi_y is None`` -> symmetric block with exact unit diagonal."""
    psi_x, psi_y, out = from_dlpack(psi_x), from_dlpack(psi_y), from_dlpack(out)
    _require_gpu(psi_x, psi_y, out)
    N, D = psi_x.shape
    M = N if psi_y is None else psi_y.shape[0]
    if out is None:
        out = torch.empty((N, M), dtype=torch.float64, device=psi_x.device)
    gram_jobs([dict(A=psi_x, B=psi_x if psi_y is None else psi_y, K=out, symmetric=psi_y is None)], D)
    return out


def block_to_host(K_dev: torch.Tensor, K_host: torch.Tensor, stream: Optional["torch.cuda.Stream"] = None) -> None:
    """Enqueue the copy of a 2-D float64 device view (unit column stride) into a host view of the same
    shape (page-locked memory for a copy that overlaps other streams) on ``stream`` (default: torch's
    current stream): ``qk_block_to_host_async``, one strided copy, no temporaries."""
    _require_gpu(K_dev)
    if K_host.is_cuda or K_dev.dtype != torch.float64 or K_host.dtype != torch.float64:
        raise ValueError("block_to_host copies a float64 device view into a float64 host view")
    if K_dev.dim() != 2 or tuple(K_dev.shape) != tuple(K_host.shape):
        raise ValueError("device and host views must be 2-D and of one shape")
    rows, cols = K_dev.shape
    if rows == 0 or cols == 0:
        return
    if K_dev.stride(1) != 1 or K_host.stride(1) != 1:
        raise ValueError("both views need unit column stride")
    s = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
    _lib.check(_lib.load().qk_block_to_host_async(
        _dp(K_dev), K_dev.stride(0) if rows > 1 else cols, rows, cols,
        C.c_void_p(K_host.data_ptr()), K_host.stride(0) if rows > 1 else cols, s))


def rdm1(psi: torch.Tensor, n_qubits: int) -> torch.Tensor:
    """Per-qubit reduced density matrices, float64 [N, n, 4] = (rho00, rho11, Re rho01, Im rho01)."""
    psi = from_dlpack(psi)
    _require_gpu(psi)
    if psi.dtype != torch.complex128 or not psi.is_contiguous() or psi.shape[1] != (1 << n_qubits):
        raise ValueError("psi must be contiguous complex128 [N, 2^n]")
    out = torch.empty((psi.shape[0], n_qubits, 4), dtype=torch.float64, device=psi.device)
    _lib.check(_lib.load().qk_rdm1_dev(_dp(psi), psi.shape[0], n_qubits, _dp(out), _stream_ptr()))
    return out


def pqk(rdm_x: torch.Tensor, rdm_y: Optional[torch.Tensor], gamma: float,
        out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Projected kernel block from RDM tables; ``rdm_y is None`` -> symmetric with unit diagonal."""
    rdm_x, rdm_y, out = from_dlpack(rdm_x), from_dlpack(rdm_y), from_dlpack(out)
    _require_gpu(rdm_x, rdm_y, out)
    N, n, _ = rdm_x.shape
    M = N if rdm_y is None else rdm_y.shape[0]
    if out is None:
        out = torch.empty((N, M), dtype=torch.float64, device=rdm_x.device)
    ldk = out.stride(0) if N > 1 else max(M, 1)
    _lib.check(_lib.load().qk_pqk_dev(_dp(rdm_x.contiguous()), N, _dp(rdm_y.contiguous()) if rdm_y is not None else None,
                                      M, n, float(gamma), _dp(out), ldk, _stream_ptr()))
    return out


def kernel_stats(K: torch.Tensor, labels: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Raw sums of the GA's kernel statistics over the strict upper triangle of a symmetric device
    K (see ``qk_kernel_stats_dev``): float64 [8] on the device."""
    K, labels = from_dlpack(K), from_dlpack(labels)
    _require_gpu(K, labels)
    if K.dtype != torch.float64 or K.dim() != 2 or K.shape[0] != K.shape[1] or K.stride(1) != 1:
        raise ValueError("K must be a square float64 matrix with unit column stride")
    if labels is not None:
        labels = labels.to(device=K.device, dtype=torch.int32).contiguous()
        if labels.numel() != K.shape[0]:
            raise ValueError("labels must have one entry per row of K")
    out = torch.empty(_lib.STATS_LEN, dtype=torch.float64, device=K.device)
    ws = torch.empty(_lib.STATS_WS, dtype=torch.float64, device=K.device)
    N = K.shape[0]
    _lib.check(_lib.load().qk_kernel_stats_dev(_dp(K), N, K.stride(0) if N > 1 else N, _dp(labels), _dp(out), _dp(ws),
                                               _stream_ptr()))
    return out


def kernel_noise_(K: torch.Tensor, mode: str, param: float, seed: int = 0, symmetric: bool = True) -> torch.Tensor:
    """In-place noise on a device K: ``mode`` = "gaussian" (param = std, reference genetic_v3.py:620-627) or "shots"
    (param = shots, the QPU-style P(0...0) estimate).  ``symmetric`` (a square train kernel): the entries right of
    the diagonal are drawn and mirrored, the diagonal stays exact; otherwise (a test-versus-train block of any
    shape) every entry is drawn."""
    K = from_dlpack(K)
    _require_gpu(K)
    if K.dtype != torch.float64 or K.dim() != 2 or K.stride(1) != 1 or K.shape[0] < 1 or K.shape[1] < 1:
        raise ValueError("K must be a float64 matrix with unit column stride")
    if symmetric and K.shape[0] != K.shape[1]:
        raise ValueError("a symmetric K must be square")
    code = {"gaussian": _lib.NOISE_GAUSSIAN, "shots": _lib.NOISE_SHOTS}[mode]
    rows, cols = K.shape
    ldk = K.stride(0) if rows > 1 else cols
    if symmetric:
        _lib.check(_lib.load().qk_kernel_noise_dev(_dp(K), rows, ldk, code, float(param), int(seed) & (2**64 - 1),
                                                   _stream_ptr()))
    else:
        _lib.check(_lib.load().qk_kernel_noise_rect_dev(_dp(K), rows, cols, ldk, code, float(param),
                                                        int(seed) & (2**64 - 1), _stream_ptr()))
    return K
