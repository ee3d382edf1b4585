"""ctypes loader for the C twin of the oracle (oracle/qk_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from . import qk_oracle as O

HERE = Path(__file__).resolve().parent
LIB = HERE / "libqk_oracle.so"
_lib = None


def build(force=False):
    if force or not LIB.exists() or LIB.stat().st_mtime < (HERE / "qk_oracle.c").stat().st_mtime:
        subprocess.check_call(["make", "-s", "-C", str(HERE), "-B", "libqk_oracle.so"])
    return LIB


def load():
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        _lib = C.CDLL(str(LIB))
        for f in ("qko_simulate", "qko_fidelity", "qko_rdm1", "qko_pqk"):
            getattr(_lib, f).restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def simulate(ops, n, X):
    """Statevectors via the C simulator; ops are oracle op tuples, binding by qk_oracle.bind."""
    P = O.bind(ops, X)
    gates, ang = O.flatten(ops, P)
    psi = np.empty((P.shape[0], 1 << n), dtype=np.complex128)
    rc = load().qko_simulate(C.c_int(n), C.c_int(len(ops)), _p(gates), _p(ang), C.c_int64(P.shape[0]), _p(psi))
    assert rc == 0
    return psi


def fidelity(psi_x, psi_y=None):
    N, D = psi_x.shape
    M = N if psi_y is None else psi_y.shape[0]
    K = np.empty((N, M))
    load().qko_fidelity(_p(psi_x), C.c_int64(N), _p(psi_y) if psi_y is not None else None, C.c_int64(M), C.c_int64(D), _p(K))
    return K


def rdm1(psi, n):
    out = np.empty((psi.shape[0], n, 4))
    load().qko_rdm1(_p(psi), C.c_int64(psi.shape[0]), C.c_int(n), _p(out))
    return out


def projected_kernel(ops, n, X, gamma):
    N = np.asarray(X).shape[0]
    if not O.used_parameters(ops):
        return np.ones((N, N))
    r = rdm1(simulate(ops, n, X), n)
    K = np.empty((N, N))
    load().qko_pqk(_p(r), C.c_int64(N), C.c_int(n), C.c_double(gamma), _p(K))
    return K
