"""ctypes binding of libquake_hip.so (C ABI declared in include/quake_hip.h).

There is no CPU fallback: if the shared library is missing or no GPU is visible,
the first call that needs it raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "libquake_hip.so"

QK_OK = 0
QK_ERR_INVALID = -1
QK_ERR_HIP = -2
QK_ERR_NO_DEVICE = -3
QK_ERR_UNSUPPORTED = -4
QK_GRAM_SYMMETRIC = 1
QK_GRAM_MAX_JOBS = 16


class QkError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libquake_hip error {code}: {message}")
        self.code = code
        self.message = message


class qk_gram_job(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("K", C.c_void_p), ("KT", C.c_void_p),
        ("ma", C.c_int64), ("nb", C.c_int64), ("ldk", C.c_int64), ("ldkt", C.c_int64),
        ("flags", C.c_int32), ("reserved", C.c_int32),
    ]


class qk_batch_opts(C.Structure):
    _fields_ = [("noise_mode", C.c_int32), ("reserved", C.c_int32), ("noise_param", C.c_double),
                ("seed", C.c_uint64)]


STATS_LEN = 8
STATS_WS = 1024 * 8
NOISE_NONE, NOISE_GAUSSIAN, NOISE_SHOTS = 0, 1, 2

# every symbol include/quake_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_I64 = C.c_int64
SIGNATURES = {
    "qk_version": (C.c_int, []),
    "qk_device_count": (C.c_int, []),
    "qk_last_error": (C.c_int, [C.c_char_p, C.c_size_t]),
    "qk_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "qk_host_free": (C.c_int, [C.c_void_p]),
    "qk_block_to_host_async": (C.c_int, [_P, _I64, _I64, _I64, _P, _I64, _P]),
    "qk_circuit_create": (C.c_int, [C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_int,
                                    C.POINTER(_P)]),
    "qk_circuit_destroy": (None, [_P]),
    "qk_circuit_num_qubits": (C.c_int, [_P]),
    "qk_circuit_num_params": (C.c_int, [_P]),
    "qk_circuit_num_passes": (C.c_int, [_P]),
    "qk_simulate_dev": (C.c_int, [_P, _P, _I64, _I64, _P, _P]),
    "qk_simulate": (C.c_int, [_P, _P, _I64, _I64, _P]),
    "qk_gram_jobs_dev": (C.c_int, [C.POINTER(qk_gram_job), C.c_int, _I64, _P]),
    "qk_gram_dev": (C.c_int, [_P, _I64, _P, _I64, _I64, _P, _I64, _P]),
    "qk_simulate_c64_dev": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P]),
    "qk_gram32_jobs_dev": (C.c_int, [C.POINTER(qk_gram_job), C.c_int, _I64, _P]),
    "qk_gram32_dev": (C.c_int, [_P, _I64, _P, _I64, _I64, _P, _I64, _P]),
    "qk_evaluate_f32": (C.c_int, [_P, _P, _I64, _P, _I64, C.c_int, _P]),
    "qk_evaluate": (C.c_int, [_P, _P, _I64, _P, _I64, C.c_int, _P]),
    "qk_rdm1_dev": (C.c_int, [_P, _I64, C.c_int, _P, _P]),
    "qk_simulate_rdm1_is_fused": (C.c_int, [_P]),
    "qk_simulate_rdm1_wants_workspace": (C.c_int, [_P, _I64]),
    "qk_simulate_rdm1_dev": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P]),
    "qk_pqk_dev": (C.c_int, [_P, _I64, _P, _I64, C.c_int, C.c_double, _P, _I64, _P]),
    "qk_projected_kernel": (C.c_int, [_P, _P, _I64, C.c_int, C.c_double, _P]),
    "qk_kernel_stats_dev": (C.c_int, [_P, _I64, _I64, _P, _P, _P, _P]),
    "qk_kernel_stats_noalloc_dev": (C.c_int, [_P, _I64, _I64, _P, _P, _P]),
    "qk_kernel_noise_dev": (C.c_int, [_P, _I64, _I64, C.c_int, C.c_double, C.c_uint64, _P]),
    "qk_kernel_noise_rect_dev": (C.c_int, [_P, _I64, _I64, _I64, C.c_int, C.c_double, C.c_uint64, _P]),
    "qk_evaluate_noisy": (C.c_int, [_P, _P, _I64, _P, _I64, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64, _P]),
    "qk_evaluate_batch": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64), C.POINTER(_P),
                                    C.POINTER(qk_batch_opts), C.POINTER(_P), _P]),
}

_lib = None


def _share_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7) and
    look it up by the unversioned file name, so a process that loads /opt/rocm's copy
    first (through this library) and imports torch later ends up with two HIP
    runtimes, and the second finds no GPU.  Loading torch's copy first -- without
    importing torch -- makes both sides resolve to the same runtime in either order."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    hip = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if hip.exists():
        C.CDLL(str(hip), mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """Load the library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m quake_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _share_hip_runtime_with_torch()
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    load().qk_last_error(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != QK_OK:
        raise QkError(rc, last_error())


PINNED_MIN_BYTES = 1 << 20   # below this a pageable numpy array costs nothing measurable


class _PinnedBlock:
    """Page-locked host memory from qk_host_alloc, released when the last numpy view of it dies."""

    def __init__(self, nbytes: int, shape, dtype):
        out = C.c_void_p()
        check(load().qk_host_alloc(nbytes, C.byref(out)))
        self._ptr = out.value
        self.__array_interface__ = {"shape": tuple(shape), "typestr": np.dtype(dtype).str,
                                    "data": (self._ptr, False), "version": 3}

    def __del__(self):
        p, self._ptr = getattr(self, "_ptr", None), None
        if p and _lib is not None:
            _lib.qk_host_free(p)


def host_empty(shape, dtype=np.float64) -> np.ndarray:
    """``np.empty`` for an output the library fills: page-locked (qk_host_alloc) when it is large, so the
    device-to-host copy runs at the PCIe rate; an ordinary array when it is small or pinning fails.
    Either way a fresh, writable, C-contiguous array the caller owns."""
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    if nbytes >= PINNED_MIN_BYTES:
        try:
            return np.asarray(_PinnedBlock(nbytes, shape, dtype))
        except QkError:
            pass    # e.g. the locked-memory limit: the pageable path below is only slower
    return np.empty(shape, dtype=dtype)


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class CircuitHandle:
    """Owns a ``qk_circuit*`` built from a :class:`quake_amd.circuit.CompiledCircuit`."""

    def __init__(self, compiled):
        lib = load()
        self.compiled = compiled
        ops = np.ascontiguousarray(compiled.ops, dtype=np.int32)
        prog = np.ascontiguousarray(compiled.prog, dtype=np.int32)
        off = np.ascontiguousarray(compiled.expr_off, dtype=np.int32)
        consts = np.ascontiguousarray(compiled.consts, dtype=np.float64)
        h = C.c_void_p()
        check(lib.qk_circuit_create(
            compiled.n_qubits, compiled.n_params,
            ptr(ops), ops.shape[0], ptr(prog), prog.shape[0],
            ptr(off), off.shape[0] - 1, ptr(consts), consts.shape[0], C.byref(h)))
        self._h = h
        self.n_qubits = compiled.n_qubits
        self.n_params = compiled.n_params

    @property
    def handle(self):
        return self._h

    @property
    def num_passes(self) -> int:
        return load().qk_circuit_num_passes(self._h)

    @property
    def plan_summary(self) -> str:
        """What the planner made of the gate list (development hook; text)."""
        lib = load()
        lib.qk_circuit_plan_summary.restype = C.c_int
        lib.qk_circuit_plan_summary.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        buf = C.create_string_buffer(2048)
        check(lib.qk_circuit_plan_summary(self._h, buf, len(buf)))
        return buf.value.decode()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.qk_circuit_destroy(h)
