"""Pin the CPU oracle against every known answer the reference holds for the
quantum-kernel path (pins P1-P5)."""
import numpy as np
import pytest
from sklearn.model_selection import cross_val_score
from sklearn.svm import SVC

from oracle import qk_oracle as O


def _decode(p, nb_features):
    return O.to_quantum(p["genes"], p["gate_dict"], nb_features, p["gates_per_qubits"], p["nb_qubits"])


def test_P1_gram_entry(pins, data8):
    p = pins["P1"]
    ops, x_idxs = _decode(p, 8)
    assert x_idxs == [3, 1, 0, 5, 4, 7, 2]
    K = O.evaluate(ops, 4, data8[:, x_idxs])
    i, j = p["entry"]
    assert abs(K[i, j] - p["expected"]) < 2e-15
    assert K.shape == (100, 100) and np.array_equal(np.diag(K), np.ones(100))
    assert np.abs(K - K.T).max() < 1e-15   # BLAS zgemm is not bit-symmetric


def test_P1_needs_sorted_binding(pins, data8):
    """Naive binding x[f] := column f gives 0.9627 instead of the golden 0.2638."""
    p = pins["P1"]
    ops, x_idxs = _decode(p, 8)
    P = np.zeros((100, 8))
    P[:, :] = data8.astype(np.float64)          # x[f] = feature f  (NOT what qiskit does)
    psi = O.simulate(ops, 4, P)
    K = O.fidelity_kernel(psi)
    assert abs(K[25, 41] - 0.9627) < 1e-3
    assert abs(K[25, 41] - p["expected"]) > 0.5


def test_P2_cv_accuracy(pins, data8, data_labels):
    p = pins["P2"]
    ops, x_idxs = _decode(p, 8)
    assert x_idxs == [2, 5, 0, 4, 3, 6, 1, 7]
    K = O.evaluate(ops, 4, data8[:, x_idxs])
    acc = cross_val_score(SVC(kernel="precomputed", C=p["svc_C"]), K, data_labels, cv=p["cv"], scoring="accuracy").mean()
    assert acc == pytest.approx(p["expected_cv_accuracy"], abs=1e-12)
    # PSD up to round-off, so skipping the reference's eig projection is safe at 1e-10
    assert np.linalg.eigvalsh(K).min() > -1e-12
    assert np.abs(O.fidelity_kernel(O.statevectors(ops, 4, data8[:, x_idxs]), enforce_psd=True) - K).max() < 1e-12


def test_P3_decoded_first_layers(pins):
    p = pins["P1"]
    ops, _ = _decode(p, 8)
    head = [(name, list(q)) for name, q, _, _ in ops[:9]]
    # layer 0: sx q0, sx q1, id q2, x q3 ; layer 1: rz q0, ecr(1,2), (k=2 skipped), ...
    assert head[:4] == [("sx", [0]), ("sx", [1]), ("id", [2]), ("x", [3])]
    assert head[4] == ("rz", [0]) and head[5] == ("ecr", [1, 2])
    # angles of the drawing: Rz(2*pi*x[3] - pi) then Rz(2*asin(2*x[1]-1) - pi) on q0
    P = np.random.default_rng(0).uniform(0.05, 0.95, size=(5, 8))
    rz_q0 = [fn for name, q, fn, _ in ops if name == "rz" and q == (0,)]
    assert np.allclose(rz_q0[0](P), 2 * np.pi * P[:, 3] - np.pi, atol=1e-15)
    assert np.allclose(rz_q0[1](P), 2 * np.arcsin(2 * P[:, 1] - 1) - np.pi, atol=1e-15)
    rz_q1 = [fn for name, q, fn, _ in ops if name == "rz" and q == (1,)]
    assert np.allclose(rz_q1[0](P), 2 * np.arcsin((2 * P[:, 0] - 1) * (2 * P[:, 5] - 1)), atol=1e-15)


def test_P4_skipped_two_qubit_genes(pins):
    p = pins["P4"]
    ops, _ = O.to_quantum(p["genes"], p["gate_dict"], p["nb_features"], p["gates_per_qubits"], p["nb_qubits"])
    g = np.reshape(p["genes"], [p["gates_per_qubits"], p["nb_qubits"], 5])
    ecr_idx = 4  # gate_list = Id, X, SX, RZ, ECR
    total = int((g[:, :, 0] == ecr_idx).sum())
    emitted = sum(1 for name, *_ in ops if name == "ecr")
    assert total - emitted == p["expected_skipped_two_qubit_genes"]
    assert int((g[:, 2, 0] == ecr_idx).sum()) == 3


def test_P5_zz_two_qubit_layout():
    """docs/825617-Roberto_Moretti.pdf p.10: H,P(2x0) / H,P(2x1), CX, P(2(pi-x0)(pi-x1)) on q1, CX."""
    ops = O.zz_featuremap(2, 1)
    assert [(n, q) for n, q, _, _ in ops] == [
        ("h", (0,)), ("p", (0,)), ("h", (1,)), ("p", (1,)), ("cx", (0, 1)), ("p", (1,)), ("cx", (0, 1))]
    P = np.array([[0.3, -0.7]])
    assert ops[5][2](P)[0] == pytest.approx(2 * (np.pi - 0.3) * (np.pi + 0.7), abs=1e-14)


def test_zz_closed_form():
    """repeats=1: psi_b = 2^{-n/2} exp(i phi_b(x)), phi_b = sum_q 2 x_q b_q + sum_q 2(pi-x_q)(pi-x_{q+1}) (b_q xor b_{q+1})."""
    n, N = 6, 7
    X = np.random.default_rng(5).uniform(-1.5, 1.5, size=(N, n))
    psi = O.statevectors(O.zz_featuremap(n, 1), n, X)
    b = (np.arange(1 << n)[:, None] >> np.arange(n)) & 1
    phi = 2 * (X @ b.T) + np.einsum("sq,bq->sb", 2 * (np.pi - X[:, :-1]) * (np.pi - X[:, 1:]), b[:, :-1] ^ b[:, 1:])
    assert np.abs(psi - np.exp(1j * phi) / 2 ** (n / 2)).max() < 1e-13


def test_zz_full_entanglement_closed_form():
    """entanglement="full" (Qiskit library default, not built by the reference): every pair i < j."""
    n, N = 5, 6
    X = np.random.default_rng(8).uniform(-1.5, 1.5, size=(N, n))
    psi = O.statevectors(O.zz_featuremap(n, 1, "full"), n, X)
    b = (np.arange(1 << n)[:, None] >> np.arange(n)) & 1
    phi = 2 * (X @ b.T)
    for i in range(n):
        for j in range(i + 1, n):
            phi = phi + np.outer(2 * (np.pi - X[:, i]) * (np.pi - X[:, j]), b[:, i] ^ b[:, j])
    assert np.abs(psi - np.exp(1j * phi) / 2 ** (n / 2)).max() < 1e-13
    with pytest.raises(ValueError):
        from quake_amd import featuremaps as fm
        from quake_amd.circuit import ParameterVector
        fm.zz_featuremap(ParameterVector("x", 3), 3, 1, entanglement="circular")


def test_unitarity_and_psd():
    n, N = 5, 40
    X = np.random.default_rng(6).uniform(-1.5, 1.5, size=(N, n))
    for ops in (O.zz_featuremap(n, 2), O.custom1_featuremap(n, 2), O.custom2_featuremap(n, 1), O.z_featuremap(n, 3)):
        psi = O.statevectors(ops, n, X)
        assert np.abs(np.linalg.norm(psi, axis=1) - 1).max() < 1e-14
        K = O.fidelity_kernel(psi)
        assert np.linalg.eigvalsh(K).min() > -1e-12 and K.max() <= 1 + 1e-14


def test_rdm_matches_explicit_density_matrix():
    n = 4
    X = np.random.default_rng(7).uniform(-1.5, 1.5, size=(3, n))
    psi = O.statevectors(O.custom1_featuremap(n, 2), n, X)
    r = O.rdm1(psi, n)
    for s in range(3):
        assert np.abs(r[s] - O.rdm1_explicit(psi[s], n)).max() < 1e-14
    assert np.abs(np.trace(r, axis1=2, axis2=3) - 1).max() < 1e-14


def test_projected_kernel_properties():
    n, N = 3, 12
    X = np.random.default_rng(8).uniform(-1.5, 1.5, size=(N, n))
    ops = O.zz_featuremap(n, 1)
    K = O.projected_kernel(ops, n, X, 1.0)
    assert np.array_equal(np.diag(K), np.ones(N)) and np.array_equal(K, K.T)
    assert (K > 0).all() and (K <= 1).all()
    # brute force pair via explicit density matrices (reference genetic_v3.py:795-804)
    psi = O.statevectors(ops, n, X)
    ri, rj = O.rdm1_explicit(psi[5], n), O.rdm1_explicit(psi[2], n)
    expo = sum(np.linalg.norm(ri[q] - rj[q]) for q in range(n))
    assert K[5, 2] == pytest.approx(np.exp(-expo), abs=1e-14)
    # no parameters -> all ones (reference :789-791)
    assert np.array_equal(O.projected_kernel([("h", (0,), None, ())], 1, np.zeros((4, 0)), 1.0), np.ones((4, 4)))


def test_bind_column_mismatch_raises():
    with pytest.raises(ValueError):
        O.bind(O.zz_featuremap(3, 1), np.zeros((2, 4)))
