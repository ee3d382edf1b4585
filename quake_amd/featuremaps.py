"""Hand-written quantum feature maps, same signatures as the reference.

Restates ``src/quake/models/qsvm/quantum_featuremaps.py`` of the reference on the
build's own circuit IR (:mod:`quake_amd.circuit`); each builder cites the lines it
follows.  The gate sequences are the definition of the kernel, so they are kept
gate for gate (including ``align=True`` prefixes and where a barrier ends a
repeat); nothing here touches the GPU.
"""
from __future__ import annotations

import numpy as np

from .circuit import ParameterVector, QuantumCircuit


def _align_prefix(circ: QuantumCircuit, qubits: int) -> None:
    # reference quantum_featuremaps.py:32-35 (same block opens every builder)
    theta = ParameterVector("θ", qubits)
    for q in range(qubits):
        circ.rx(theta[q], q)


def z_featuremap(x: ParameterVector, qubits: int, repeats: int = 1, align: bool = False) -> QuantumCircuit:
    """The Z feature map (reference quantum_featuremaps.py:9-41)."""
    circ = QuantumCircuit(qubits)
    if align:
        _align_prefix(circ, qubits)
    for _ in range(repeats):
        for q in range(qubits):
            circ.h(q)
            circ.p(2 * x[q], q)
        circ.barrier()
    return circ


def zz_featuremap(x: ParameterVector, qubits: int, repeats: int = 1, align: bool = False,
                  entanglement: str = "linear") -> QuantumCircuit:
    """The ZZ feature map (reference quantum_featuremaps.py:44-81).  The reference builds LINEAR
    entanglement (neighbouring qubits); ``entanglement="full"`` is the Qiskit library default
    (every pair i < j, in ``itertools.combinations`` order), offered because SURVEY 8d asks for it
    as a separately reported variant -- no reference run uses it."""
    if entanglement not in ("linear", "full"):
        raise ValueError(f"entanglement must be 'linear' or 'full', got {entanglement!r}")
    pairs = [(q, q + 1) for q in range(qubits - 1)] if entanglement == "linear" else \
        [(i, j) for i in range(qubits) for j in range(i + 1, qubits)]
    circ = QuantumCircuit(qubits)
    if align:
        _align_prefix(circ, qubits)
    for _ in range(repeats):
        for q in range(qubits):
            circ.h(q)
            circ.p(2 * x[q], q)
        for i, j in pairs:
            circ.cx(i, j)
            circ.p(2 * (np.pi - x[i]) * (np.pi - x[j]), j)
            circ.cx(i, j)
        circ.barrier()
    return circ


def custom1_featuremap(x: ParameterVector, qubits: int, repeats: int = 1, align: bool = False) -> QuantumCircuit:
    """Custom entangled map (reference quantum_featuremaps.py:84-126)."""
    circ = QuantumCircuit(qubits)
    if align:
        _align_prefix(circ, qubits)
    for _ in range(repeats):
        for q in range(qubits):
            if q % 2 == 0:
                circ.ry(2 * x[q], q)
            else:
                circ.rx(2 * x[q], q)
            circ.p(2 * x[q], q)
        for i in range(qubits - 1):
            circ.cx(i, i + 1)
            circ.p(x[i] * x[i + 1], i + 1)
            circ.cx(i, i + 1)
        circ.barrier()
    return circ


def custom2_featuremap(x: ParameterVector, qubits: int, repeats: int = 1, align: bool = False) -> QuantumCircuit:
    """Custom non-entangled map (reference quantum_featuremaps.py:129-169).

    The reference's even/odd branches are identical; one loop body covers both."""
    circ = QuantumCircuit(qubits)
    if align:
        _align_prefix(circ, qubits)
    for _ in range(repeats):
        for q in range(qubits):
            circ.rx(np.arcsin(2 / np.pi * x[q]), q)
            circ.rz(np.arccos(0.4 * x[q] * x[q]), q)
        circ.barrier()
    return circ


def genetic_featuremap(x: ParameterVector, repeats: int = 1, align: bool = False) -> QuantumCircuit:
    """Entangled 2-qubit map found by the genetic search (reference quantum_featuremaps.py:172-207)."""
    circ = QuantumCircuit(2)
    if align:
        _align_prefix(circ, 2)
    for _ in range(repeats):
        circ.ry(np.arcsin(2 / np.pi * x[0]), 0)
        circ.ry(2 * x[1], 0)
        circ.rx(np.arcsin(2 / np.pi * x[1]), 1)
        circ.ry(np.arcsin(2 / np.pi * x[1]), 1)
        circ.cx(0, 1)
        circ.ry(2 * x[0], 1)
        circ.barrier()
    return circ


def genetic_featuremap_2(x: ParameterVector, repeats: int = 1, align: bool = False) -> QuantumCircuit:
    """Non-entangled 2-qubit genetic map (reference quantum_featuremaps.py:210-244)."""
    circ = QuantumCircuit(2)
    if align:
        _align_prefix(circ, 2)
    for _ in range(repeats):
        circ.ry(2 * x[0], 0)
        circ.rx(np.arccos(0.4 * x[1] * x[1]), 1)
        circ.rz(np.arccos(0.4 * x[0] * x[0]), 0)
        circ.h(1)
        circ.rz(2 * x[1], 1)
    return circ


def genetic_featuremap_highres(x: ParameterVector, repeats: int = 1, align: bool = False) -> QuantumCircuit:
    """Entangled 2-qubit genetic map, high-resolution run (reference quantum_featuremaps.py:247-281)."""
    circ = QuantumCircuit(2)
    if align:
        _align_prefix(circ, 2)
    for _ in range(repeats):
        circ.rx(0.4 * x[1] * x[1], 0)
        circ.ry(2 * x[0], 1)
        circ.rz(2 * x[0], 1)
        circ.cx(0, 1)
        circ.ry(2 * x[0], 1)
        circ.barrier()
    return circ


def genetic_featuremap_highres_2(x: ParameterVector, repeats: int = 1, align: bool = False) -> QuantumCircuit:
    """Non-entangled 2-qubit genetic map, high-resolution run (reference quantum_featuremaps.py:284-317)."""
    circ = QuantumCircuit(2)
    if align:
        _align_prefix(circ, 2)
    for _ in range(repeats):
        circ.h(0)
        circ.rx(np.arcsin(2 / np.pi * x[1]), 1)
        circ.rz(np.arccos(0.4 * x[1] * x[1]), 0)
        circ.rz(2 * x[0], 1)
        circ.p(np.arcsin(2 / np.pi * x[0]), 0)
    return circ


def genetic_attention(x: ParameterVector, repeats: int = 1, align: bool = False) -> QuantumCircuit:
    """Non-entangled genetic map on attention features (reference quantum_featuremaps.py:320-352)."""
    circ = QuantumCircuit(2)
    if align:
        _align_prefix(circ, 2)
    for _ in range(repeats):
        circ.rx(np.arccos(0.4 * x[0] * x[0]), 0)
        circ.rx(np.arccos(2 / np.pi * x[0]), 1)
        circ.rz(4 * x[0] * x[1], 1)
    return circ


def genetic_attention_highres(x: ParameterVector, repeats: int = 1, align: bool = False) -> QuantumCircuit:
    """Entangled genetic map on attention features (reference quantum_featuremaps.py:355-387)."""
    circ = QuantumCircuit(2)
    if align:
        _align_prefix(circ, 2)
    for _ in range(repeats):
        circ.rx(4 * x[0] * x[1], 0)
        circ.cx(0, 1)
        circ.ry(np.arcsin(2 / np.pi * x[1]), 0)
        circ.rx(2 * x[0], 1)
    return circ


FEATURE_MAPS = {
    "z": z_featuremap,
    "zz": zz_featuremap,
    "custom1": custom1_featuremap,
    "custom2": custom2_featuremap,
    "genetic": genetic_featuremap,
    "genetic_2": genetic_featuremap_2,
    "genetic_highres": genetic_featuremap_highres,
    "genetic_highres_2": genetic_featuremap_highres_2,
    "genetic_attention": genetic_attention,
    "genetic_attention_highres": genetic_attention_highres,
}
_TWO_QUBIT_ONLY = {
    "genetic", "genetic_2", "genetic_highres", "genetic_highres_2",
    "genetic_attention", "genetic_attention_highres",
}


def build_featuremap(name: str, qubits: int, repeats: int = 1, align: bool = False, **options) -> QuantumCircuit:
    """Runcard entry point: build a named map over ``x = ParameterVector("x", qubits)``
    (what ``load_quantum_kernels`` does at reference qsvm/train.py:180-184).  ``options`` go to the
    builder (``entanglement: full`` for the ZZ map)."""
    key = name.lower()
    if key not in FEATURE_MAPS:
        raise KeyError(f"unknown feature map '{name}'; known: {sorted(FEATURE_MAPS)}")
    x = ParameterVector("x", length=max(int(qubits), 2 if key in _TWO_QUBIT_ONLY else 1))
    if key in _TWO_QUBIT_ONLY:
        return FEATURE_MAPS[key](x, repeats=repeats, align=align, **options)
    return FEATURE_MAPS[key](x, qubits, repeats=repeats, align=align, **options)
