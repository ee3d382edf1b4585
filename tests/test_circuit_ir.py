"""Host logic on CPU: the product's circuit IR, builders, genome decoder and the
flat program it hands to the C ABI, checked against the independent oracle."""
import numpy as np
import pytest

from oracle import qk_oracle as O
from quake_amd import circuit as qc
from quake_amd import featuremaps as fm
from quake_amd import genetic
from tests._helpers import map_pairs, psi_from_compiled, synth_X, two_qubit_pairs

GATE_DICT_FULL = {
    "single_non_parametric": ["Id", "H", "X", "SX"],
    "single_parametric": ["RX", "RY", "RZ", "P"],
    "two_non_parametric": ["CX", "CZ", "ECR"],
    "two_parametric": [],
}


@pytest.mark.parametrize("n,repeats", [(1, 1), (2, 2), (3, 1), (5, 2)])
def test_builders_match_oracle(n, repeats):
    X = synth_X(6, n, "scaled", 11)
    for name, (circ, oops, nq) in map_pairs(n, repeats).items():
        assert [i.name for i in circ.data] == [o[0] for o in oops], name
        assert [i.qubits for i in circ.data] == [o[1] for o in oops], name
        got = psi_from_compiled(circ, X)
        ref = O.statevectors(oops, nq, X)
        assert np.abs(got - ref).max() < 1e-14, name


@pytest.mark.parametrize("repeats", [1, 2])
def test_two_qubit_builders_match_oracle(repeats):
    X = synth_X(6, 2, "scaled", 12)
    for name, (circ, oops, nq) in two_qubit_pairs(repeats).items():
        Xn = X[:, :circ.num_parameters]
        got = psi_from_compiled(circ, Xn)
        ref = O.statevectors(oops, nq, Xn)
        assert np.abs(got - ref).max() < 1e-14, name


def test_parameter_order_and_binding():
    x = qc.ParameterVector("x", 12)
    c = qc.QuantumCircuit(2)
    c.rz(2 * x[10], 0)
    c.rx(x[2] * x[7], 1)
    c.p(np.arcsin(x[2]), 0)
    assert [p.name for p in c.parameters] == ["x[2]", "x[7]", "x[10]"]   # numeric, not lexical
    comp = c.compile()
    assert comp.n_params == 3
    # column 2 feeds x[10], column 0 feeds x[2]
    bound = c.assign_parameters([0.1, 0.2, 0.3])
    angles = [ins.angle.evaluate({}) for ins in bound.data]
    assert angles == pytest.approx([2 * 0.3, 0.1 * 0.2, np.arcsin(0.1)], abs=1e-16)
    with pytest.raises(ValueError):
        c.assign_parameters([0.1, 0.2])
    # theta vector sorts after x (the align=True prefix, reference qsvm_tester.py:402-403)
    a = fm.zz_featuremap(qc.ParameterVector("x", 3), 3, align=True)
    names = [p.name for p in a.parameters]
    assert names == ["x[0]", "x[1]", "x[2]", "θ[0]", "θ[1]", "θ[2]"]


def test_expression_forms_and_numpy_dispatch():
    x = qc.ParameterVector("x", 2)
    e = (2 * np.pi * x[0] * (1 - x[1]) - np.pi) * (2 * np.pi * x[1] * (1 - x[0]) - np.pi) / np.pi
    v = e.evaluate({x[0]: 0.3, x[1]: 0.8})
    assert v == pytest.approx((2 * np.pi * 0.3 * 0.2 - np.pi) * (2 * np.pi * 0.8 * 0.7 - np.pi) / np.pi, abs=1e-15)
    assert isinstance(np.arccos(0.4 * x[0] * x[0]), qc.ParameterExpression)
    assert isinstance(np.float64(2.0) * x[0], qc.ParameterExpression)
    assert (-x[0] + 1).evaluate({x[0]: 0.25}) == 0.75
    assert (1 / (x[0] + 1)).evaluate({x[0]: 1.0}) == 0.5


def test_to_quantum_pins(pins, data8):
    p = pins["P1"]
    fmap, x_idxs = genetic.to_quantum(p["genes"], p["gate_dict"], 8, 8, 4, verbose=False)
    oops, ox = O.to_quantum(p["genes"], p["gate_dict"], 8, 8, 4)
    assert x_idxs == ox == [3, 1, 0, 5, 4, 7, 2]
    assert [i.name for i in fmap.data] == [o[0] for o in oops]
    assert [i.qubits for i in fmap.data] == [o[1] for o in oops]
    assert [p_.index for p_ in fmap.parameters] == sorted(x_idxs)
    # the full kernel through the product's compiled program, simulated by the oracle
    psi = psi_from_compiled(fmap, data8[:, x_idxs])
    K = O.fidelity_kernel(psi)
    assert abs(K[25, 41] - p["expected"]) < 2e-15


def test_to_quantum_p4_message_count(pins, capsys):
    p = pins["P4"]
    genetic.to_quantum(p["genes"], p["gate_dict"], p["nb_features"], p["gates_per_qubits"], p["nb_qubits"])
    out = capsys.readouterr().out
    assert out.count("[0, 1, 2, 3] for qubit 2 I did nothing") == p["expected_skipped_two_qubit_genes"]


@pytest.mark.parametrize("variant", ["v3", "v4", "v5"])
def test_to_quantum_random_genomes_match_oracle(variant):
    rng = np.random.default_rng(21)
    G, Q, F = 6, 4, 9
    cmap = [(0, 1), (1, 0), (1, 2), (2, 1), (3, 2)] if variant != "v3" else None
    if variant == "v5":
        cmap = [[0, 1], [1, 0], (1, 2), [2, 3], [3, 2]]
    for trial in range(8):
        genes = genetic.initial_population(1, Q, G, GATE_DICT_FULL, F, rng=rng)[0]
        kw = dict(coupling_map=cmap, variant=variant)
        fmap, xi = genetic.to_quantum(genes, GATE_DICT_FULL, F, G, Q, verbose=False, **kw)
        oops, oxi = O.to_quantum(genes, GATE_DICT_FULL, F, G, Q, **kw)
        assert xi == oxi
        assert [(i.name, i.qubits) for i in fmap.data] == [(o[0], o[1]) for o in oops]
        if fmap.num_parameters == len(xi) and xi:
            X = synth_X(5, F, "unit", trial)[:, xi]
            assert np.abs(psi_from_compiled(fmap, X) - O.statevectors(oops, Q, X)).max() < 1e-13


def test_to_quantum_edge_cases():
    gd = {"single_non_parametric": ["I", "H"], "single_parametric": ["RZ"],
          "two_non_parametric": ["CX"], "two_parametric": ["CRZ"]}
    # "I" is skipped entirely (reference genetic_v3.py:379); two_parametric only pollutes x_idxs
    genes = [0, 0, 0, 0, 1,   1, 0, 0, 0, 1,   4, 0, 0, 2, 1,   2, 0, 1, 1, 0]
    fmap, xi = genetic.to_quantum(genes, gd, 3, 1, 4, verbose=False)
    assert [(i.name, i.qubits) for i in fmap.data] == [("h", (1,)), ("rz", (3,))]
    assert xi == [2, 1, 0]
    assert fmap.num_parameters == 2 and len(xi) == 3     # evaluate() would raise on the column count
    # two-qubit gene on qubit >= 4 with the default 4-entry physical_qubits: IndexError as in the reference
    genes5 = [0, 0, 0, 0, 1] * 4 + [3, 0, 0, 0, 1]
    with pytest.raises(IndexError):
        genetic.to_quantum(genes5, {"single_non_parametric": ["I", "H"], "single_parametric": ["RZ"],
                                    "two_non_parametric": ["CX"], "two_parametric": []}, 3, 1, 5, verbose=False)
    with pytest.raises(TypeError):
        genetic.to_quantum(genes, gd, 3, 1, 4, variant="v4")
    # param_shift moves the parameter indices (reference genetic_v4.py:261,269-270)
    fmap2, xi2 = genetic.to_quantum(genes, gd, 3, 1, 4, param_shift=5, coupling_map=[(0, 1)], variant="v4", verbose=False)
    assert xi2 == [7, 6, 5]


def test_population_and_gene_space():
    rng = np.random.default_rng(3)
    pop = genetic.initial_population(7, 4, 6, GATE_DICT_FULL, 9, rng=rng)
    assert pop.shape == (7, 4 * 6 * 5)
    g = pop.reshape(7, 6, 4, 5)
    assert g[..., 0].max() < 11 and g[..., 1].max() <= 2 and g[..., 2].max() <= 1 and g[..., 3].max() < 9
    assert (g[..., 3] != g[..., 4]).all()
    space = genetic.get_gene_space(GATE_DICT_FULL, 9, 4, 6)
    assert len(space) == 4 * 6 * 5 and space[1] == range(2) and space[0] == range(11)


def test_kernel_statistics_match_reference_formula():
    rng = np.random.default_rng(4)
    A = rng.uniform(size=(20, 20))
    K = (A + A.T) / 2
    np.fill_diagonal(K, 1)
    y = (rng.uniform(size=20) > 0.5).astype(float)
    s = genetic.kernel_statistics(K, y)
    assert s["offdiagonal_mean"] == np.mean(np.triu(K, 1)) and s["offdiagonal_std"] == np.std(np.triu(K, 1))
    k0 = K[y == 0][:, y == 0]
    k1 = K[y == 1][:, y == 1]
    exp = (k0[np.triu_indices(k0.shape[0], 1)].mean() + k1[np.triu_indices(k1.shape[0], 1)].mean()) / 2 - K[y == 0][:, y == 1].mean()
    assert s["sparsity"] == pytest.approx(exp, abs=1e-15)


def test_unsupported_gate_is_loud():
    c = qc.QuantumCircuit(2)
    with pytest.raises(NotImplementedError):
        c.crx(0.1, 0, 1)
    with pytest.raises(IndexError):
        c.h(2)
    with pytest.raises(ValueError):
        c.cx(1, 1)
