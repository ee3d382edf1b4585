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


def test_transpile_returns_the_lowered_circuit_a_fitness_function_can_read():
    """reference genetic_v3.py:712 hands fit_fun transpile(fmap, coupling_map, basis_gates, optimization_level=0);
    examples/genetic_featuremap_v3.py:77-85 reads count_ops() from it"""
    from quake_amd.transpile import transpile
    c = qc.QuantumCircuit(3)
    c.h(0); c.id(1); c.cz(0, 2); c.rx(0.3, 1)
    t = transpile(c, [[0, 1], [1, 2]], IBM, optimization_level=0)
    ops = t.count_ops()
    assert "id" not in ops and "h" not in ops and "cz" not in ops and set(ops) <= set(IBM)
    assert ops["cx"] == 1 + 2 * 3                       # cz -> one cx, routed over one hop: 3 cx there, 3 back
    assert t.depth() == transpiled_depth(c, [[0, 1], [1, 2]], IBM)
    assert t.size() == len(t) == sum(ops.values()) and t.num_nonlocal_gates() == ops["cx"]
    assert list(ops.values()) == sorted(ops.values(), reverse=True)
    plain = transpile(c)
    assert plain.count_ops() == {"h": 1, "cz": 1, "rx": 1} and plain.num_qubits == 3


def test_without_kernel_mode_gets_the_lowered_circuit():
    """ga.generation_fitness: a fit_fun without an 'accuracy' parameter never touches the GPU and is called
    with the lowered circuit (reference genetic_v3.py:608-611, 710-712)"""
    import numpy as np
    from quake_amd import ga, genetic
    gate_dict = {"single_non_parametric": ["Id", "H"], "single_parametric": ["RX"],
                 "two_non_parametric": ["CX"], "two_parametric": []}
    pop = genetic.initial_population(3, 4, 3, gate_dict, 5, rng=np.random.default_rng(3))
    seen = []

    def fit_fun(fmap):
        seen.append(fmap)
        return -float(fmap.count_ops().get("cx", 0)) - 0.01 * fmap.depth()

    vals, stats = ga.generation_fitness(pop, None, None, gate_dict, 5, 3, 4, fit_fun, coupling_map=[[0, 1], [1, 2], [2, 3]],
                                        basis_gates=IBM)
    assert stats is None and len(vals) == 3 and len(seen) == 3
    for sol, t, v in zip(pop, seen, vals):
        fmap, _ = genetic.to_quantum(sol, gate_dict, 5, 3, 4, verbose=False)
        assert set(t.count_ops()) <= set(IBM)
        assert t.depth() == transpiled_depth(fmap, [[0, 1], [1, 2], [2, 3]], IBM)
        assert v == -float(t.count_ops().get("cx", 0)) - 0.01 * t.depth()
    # a caller's own lowering is used when given
    got = ga.generation_fitness(pop[:1], None, None, gate_dict, 5, 3, 4, lambda fmap: fmap, lower_fn=lambda f, cm, bg, optimization_level: ("mine", cm, bg, optimization_level))
    assert got[0][0] == ("mine", None, None, 0)
