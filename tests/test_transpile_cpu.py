"""The native depth counter that stands in for transpile(...).depth() (quake_amd/transpile.py)."""
import numpy as np
import pytest

from quake_amd import circuit as qc
from quake_amd.transpile import lower, transpiled_depth

IBM = ["cx", "id", "rz", "sx", "x"]


def test_plain_depth_drops_identities():
    c = qc.QuantumCircuit(3)
    c.id(0); c.h(0); c.id(1); c.cx(0, 1); c.x(2); c.cx(1, 2)
    assert c.depth() == 4                       # the IR counts id like Qiskit's depth()
    assert transpiled_depth(c) == 3             # h, cx(0,1), cx(1,2)


def test_basis_lowering_and_rz_merging():
    x = qc.ParameterVector("x", 2)
    c = qc.QuantumCircuit(2)
    c.h(0)                                       # rz sx rz
    c.rz(x[0], 0); c.p(x[1], 0)                  # both merge into the trailing rz of h
    assert [g for g, _ in lower(c, None, IBM)] == ["rz", "sx", "rz"]
    assert transpiled_depth(c, None, IBM) == 3
    c.cz(0, 1)                                   # h(1) cx h(1): the control-side rz chain is untouched
    assert transpiled_depth(c, None, IBM) == 3 + 1 + 3
    d = qc.QuantumCircuit(2); d.ecr(0, 1)
    assert [g for g, _ in lower(d, None, ["ecr", "rz", "sx", "x"])] == ["ecr"]
    assert sum(g == "cx" for g, _ in lower(d, None, IBM)) == 1
    e = qc.QuantumCircuit(2); e.cx(0, 1)
    assert sum(g == "ecr" for g, _ in lower(e, None, ["ecr", "rz", "sx", "x"])) == 1
    with pytest.raises(NotImplementedError):
        lower(e, None, ["rz", "sx"])


def test_routing_on_a_line():
    c = qc.QuantumCircuit(4)
    c.cx(0, 3)
    line = [[0, 1], [1, 2], [2, 3]]
    g = lower(c, line, None)
    assert len(g) == 2 * 2 * 3 + 1               # two swaps in, the gate, two swaps out
    assert all(abs(a - b) == 1 for _, (a, b) in g)
    assert transpiled_depth(c, line, None) == 13
    c2 = qc.QuantumCircuit(4); c2.cx(1, 2)
    assert transpiled_depth(c2, line, None) == 1
    with pytest.raises(ValueError):
        lower(c, [[0, 1], [2, 3]], None)


def test_rotation_costs():
    x = qc.ParameterVector("x", 1)
    for name, want in (("rx", 5), ("ry", 4), ("sx", 1), ("x", 1), ("p", 1)):
        c = qc.QuantumCircuit(1)
        getattr(c, name)(*((x[0], 0) if name in ("rx", "ry", "p") else (0,)))
        assert transpiled_depth(c, None, IBM) == want, name
