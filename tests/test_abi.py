"""The C-ABI library loads on a CPU-only machine and exports every symbol that
include/quake_hip.h declares.  Circuit planning is host code, so it is tested
here too; nothing in this file launches a kernel."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from quake_amd import _lib, build
from quake_amd import circuit as qc
from quake_amd import featuremaps as fm

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    if build.needs_build():
        build.build_library(verbose=False)
    return _lib.load()


def test_exports_match_header(lib):
    header = (ROOT / "include" / "quake_hip.h").read_text()
    declared = set(re.findall(r"\b(qk_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.qk_version() >= 100


def test_struct_layout_matches_header():
    assert C.sizeof(_lib.qk_gram_job) == 4 * 8 + 4 * 8 + 2 * 4


def _handle(circ):
    return _lib.CircuitHandle(circ.compile())


def test_circuit_create_and_pass_planning(lib):
    # one repeat = first gates (hoisted into the initial product state) + one diagonal block: a single
    # pass at any width
    for n, passes in [(1, 1), (8, 1), (13, 1), (14, 1), (16, 1)]:
        h = _handle(fm.zz_featuremap(qc.ParameterVector("x", n), n, 1))
        assert lib.qk_circuit_num_qubits(h.handle) == n
        assert lib.qk_circuit_num_params(h.handle) == n
        assert h.num_passes == passes, (n, h.num_passes)
    # two repeats at 16 qubits: the second H layer needs every qubit local once -> 2 passes
    h = _handle(fm.zz_featuremap(qc.ParameterVector("x", 16), 16, 2))
    assert h.num_passes == 2
    # diagonal-only circuit needs a single pass even above 13 qubits
    c = qc.QuantumCircuit(18)
    x = qc.ParameterVector("x", 18)
    for q in range(18):
        c.rz(x[q], q)
    for q in range(17):
        c.cz(q, q + 1)
    assert _handle(c).num_passes == 1


def test_circuit_create_rejects_bad_programs(lib):
    ops = np.array([[qc.GATE_RX, 0, -1, 0]], dtype=np.int32)
    off = np.array([0, 1], dtype=np.int32)
    consts = np.array([1.0])
    out = C.c_void_p()

    def create(ops, prog, n_params=1, n_qubits=2):
        prog = np.asarray(prog, dtype=np.int32).reshape(-1, 2)
        o = np.array([0, prog.shape[0]], dtype=np.int32)
        return lib.qk_circuit_create(n_qubits, n_params, _lib.ptr(ops), ops.shape[0], _lib.ptr(prog), prog.shape[0],
                                     _lib.ptr(o), 1, _lib.ptr(consts), 1, C.byref(out))

    assert create(ops, [[qc.EX_PARAM, 0]]) == 0
    lib.qk_circuit_destroy(out)
    assert create(ops, [[qc.EX_PARAM, 3]]) == _lib.QK_ERR_INVALID          # column out of range
    assert "parameter column" in _lib.last_error()
    assert create(ops, [[qc.EX_ADD, 0]]) == _lib.QK_ERR_INVALID            # stack underflow
    assert create(ops, [[qc.EX_CONST, 0], [qc.EX_CONST, 0]]) == _lib.QK_ERR_INVALID   # leaves two values
    assert create(ops, [[99, 0]]) == _lib.QK_ERR_INVALID
    bad_q = np.array([[qc.GATE_H, 5, -1, -1]], dtype=np.int32)
    assert create(bad_q, [[qc.EX_PARAM, 0]]) == _lib.QK_ERR_INVALID
    same = np.array([[qc.GATE_CX, 1, 1, -1]], dtype=np.int32)
    assert create(same, [[qc.EX_PARAM, 0]]) == _lib.QK_ERR_INVALID
    assert create(ops, [[qc.EX_PARAM, 0]], n_qubits=0) == _lib.QK_ERR_INVALID
    noexpr = np.array([[qc.GATE_RZ, 0, -1, -1]], dtype=np.int32)
    assert create(noexpr, [[qc.EX_PARAM, 0]]) == _lib.QK_ERR_INVALID


def test_evaluate_without_gpu_fails_loudly(lib):
    """No CPU fallback: on a box without a GPU the hot path raises instead of computing."""
    if lib.qk_device_count() > 0:
        pytest.skip("a GPU is visible")
    from quake_amd.kernels import HipFidelityKernel
    k = HipFidelityKernel(fm.z_featuremap(qc.ParameterVector("x", 2), 2))
    with pytest.raises(_lib.QkError) as e:
        k.evaluate(np.zeros((3, 2)))
    assert e.value.code == _lib.QK_ERR_NO_DEVICE
    # shape validation still comes first, as the reference's ValueError
    with pytest.raises(ValueError):
        k.evaluate(np.zeros((3, 5)))
