"""The C twin of the oracle agrees with the numpy oracle (two independent restatements)
and reproduces the reference's golden value P1 on its own."""
import numpy as np

from oracle import qk_oracle as O
from oracle import qk_oracle_c as OC
from tests._helpers import synth_X


def test_c_twin_P1(pins, data8):
    p = pins["P1"]
    ops, xi = O.to_quantum(p["genes"], p["gate_dict"], 8, 8, 4)
    K = OC.fidelity(OC.simulate(ops, 4, data8[:, xi]))
    assert abs(K[25, 41] - p["expected"]) < 2e-15
    assert np.array_equal(K, K.T) and np.array_equal(np.diag(K), np.ones(100))


def test_c_twin_matches_numpy_oracle():
    for n, ops in [(5, O.zz_featuremap(5, 2)), (4, O.custom1_featuremap(4, 2)), (3, O.custom2_featuremap(3, 1)),
                   (2, O.genetic_featuremap(2)), (2, O.genetic_attention_highres(1))]:
        X = synth_X(21, len(O.used_parameters(ops)), "scaled", n)
        a, b = OC.simulate(ops, n, X), O.statevectors(ops, n, X)
        assert np.abs(a - b).max() < 1e-14
        Y = X[:9] * 0.5
        assert np.abs(OC.fidelity(a) - O.fidelity_kernel(b)).max() < 1e-14
        assert np.abs(OC.fidelity(a, OC.simulate(ops, n, Y)) - O.evaluate(ops, n, X, Y)).max() < 1e-14
        assert np.abs(OC.projected_kernel(ops, n, X, 0.7) - O.projected_kernel(ops, n, X, 0.7)).max() < 1e-14


def test_c_twin_two_qubit_gates():
    ops = [("h", (0,), None, ()), ("sx", (2,), None, ()), ("cx", (0, 2), None, ()), ("ecr", (2, 1), None, ()),
           ("cz", (1, 0), None, ()), ("cx", (2, 0), None, ()), ("ecr", (0, 1), None, ()), ("x", (1,), None, ())]
    X = np.zeros((1, 0))
    assert np.abs(OC.simulate(ops, 3, X) - O.statevectors(ops, 3, X)).max() < 1e-15
