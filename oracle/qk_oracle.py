"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  Nothing under ``quake_amd/`` does.

A plain-numpy (complex128) restatement of the reference's quantum-kernel path:

* feature-map definitions        -> reference ``src/quake/models/qsvm/quantum_featuremaps.py``
* genome decoder                 -> reference ``src/quake/models/qsvm/genetic_v3.py:326-448``
                                    (+ ``genetic_v4.py:221-347``, ``genetic_v5.py:221-344``)
* parameter binding / statevector / fidelity kernel
                                 -> third-party ``qiskit`` + ``qiskit_machine_learning``
                                    (un-vendored, unpinned in the reference's ``setup.cfg:23-40``;
                                    NOT installed here).  Restated from their published
                                    conventions as recorded in SURVEY.md section 8a rows A3-A5;
                                    call sites: reference ``qsvm/train.py:92,152,154``,
                                    ``genetic_v3.py:618-619``.
* projected quantum kernel       -> reference ``genetic_v3.py:766-806``

Pinning: checked in ``tests/test_oracle_golden.py`` against every known answer the
reference holds for this path (pins P1-P4).  P1 pins
{id, x, sx, rz, ecr} + sorted binding + |<.|.>|^2 to 1e-15; P2 pins cz through an
SVC accuracy.  Gates {h, p, cx, rx, ry}, the projected kernel and everything above
four qubits are PARITY UNPINNED by any reference-owned number: for those this file
*is* the standard (textbook gate matrices), cross-checked by an independent C twin
(``oracle/qk_oracle.c``) and by invariants in the tests.

It is written independently of ``quake_amd.circuit`` on purpose: circuits here are
plain lists ``(gate name, qubits, angle function)`` with angles evaluated directly
in numpy, so a builder bug in the product cannot hide in a shared code path.
"""
from __future__ import annotations

import numpy as np

PI = np.pi
SQ2 = 1.0 / np.sqrt(2.0)

# --------------------------------------------------------------------------- #
# circuit definitions: list of (name, qubits, angle_fn | None, params_used)
# angle_fn maps P (N x vector_length, P[:, i] = value of parameter x[i]) -> (N,)
# --------------------------------------------------------------------------- #


def _g(name, qubits, fn=None, used=()):
    return (name, tuple(qubits), fn, tuple(used))


def zz_featuremap(n, repeats=1, entanglement="linear"):
    """reference quantum_featuremaps.py:66-81 (linear); "full" = the Qiskit library default pair list
    (itertools.combinations order), which no reference run uses"""
    pairs = [(q, q + 1) for q in range(n - 1)] if entanglement == "linear" else \
        [(i, j) for i in range(n) for j in range(i + 1, n)]
    ops = []
    for _ in range(repeats):
        for q in range(n):
            ops.append(_g("h", [q]))
            ops.append(_g("p", [q], (lambda P, q=q: 2 * P[:, q]), [q]))
        for i, j in pairs:
            ops.append(_g("cx", [i, j]))
            ops.append(_g("p", [j], (lambda P, i=i, j=j: 2 * (PI - P[:, i]) * (PI - P[:, j])), [i, j]))
            ops.append(_g("cx", [i, j]))
    return ops


def z_featuremap(n, repeats=1):
    """reference quantum_featuremaps.py:31-41"""
    ops = []
    for _ in range(repeats):
        for q in range(n):
            ops.append(_g("h", [q]))
            ops.append(_g("p", [q], (lambda P, q=q: 2 * P[:, q]), [q]))
    return ops


def custom1_featuremap(n, repeats=1):
    """reference quantum_featuremaps.py:106-126"""
    ops = []
    for _ in range(repeats):
        for q in range(n):
            ops.append(_g("ry" if q % 2 == 0 else "rx", [q], (lambda P, q=q: 2 * P[:, q]), [q]))
            ops.append(_g("p", [q], (lambda P, q=q: 2 * P[:, q]), [q]))
        for i in range(n - 1):
            ops.append(_g("cx", [i, i + 1]))
            ops.append(_g("p", [i + 1], (lambda P, i=i: P[:, i] * P[:, i + 1]), [i, i + 1]))
            ops.append(_g("cx", [i, i + 1]))
    return ops


def custom2_featuremap(n, repeats=1):
    """reference quantum_featuremaps.py:151-169"""
    ops = []
    for _ in range(repeats):
        for q in range(n):
            ops.append(_g("rx", [q], (lambda P, q=q: np.arcsin(2 / PI * P[:, q])), [q]))
            ops.append(_g("rz", [q], (lambda P, q=q: np.arccos(0.4 * P[:, q] * P[:, q])), [q]))
    return ops


def genetic_featuremap(repeats=1):
    """reference quantum_featuremaps.py:194-207"""
    ops = []
    for _ in range(repeats):
        ops += [
            _g("ry", [0], lambda P: np.arcsin(2 / PI * P[:, 0]), [0]),
            _g("ry", [0], lambda P: 2 * P[:, 1], [1]),
            _g("rx", [1], lambda P: np.arcsin(2 / PI * P[:, 1]), [1]),
            _g("ry", [1], lambda P: np.arcsin(2 / PI * P[:, 1]), [1]),
            _g("cx", [0, 1]),
            _g("ry", [1], lambda P: 2 * P[:, 0], [0]),
        ]
    return ops


def genetic_featuremap_2(repeats=1):
    """reference quantum_featuremaps.py:232-244"""
    ops = []
    for _ in range(repeats):
        ops += [
            _g("ry", [0], lambda P: 2 * P[:, 0], [0]),
            _g("rx", [1], lambda P: np.arccos(0.4 * P[:, 1] * P[:, 1]), [1]),
            _g("rz", [0], lambda P: np.arccos(0.4 * P[:, 0] * P[:, 0]), [0]),
            _g("h", [1]),
            _g("rz", [1], lambda P: 2 * P[:, 1], [1]),
        ]
    return ops


def genetic_featuremap_highres(repeats=1):
    """reference quantum_featuremaps.py:269-281"""
    ops = []
    for _ in range(repeats):
        ops += [
            _g("rx", [0], lambda P: 0.4 * P[:, 1] * P[:, 1], [1]),
            _g("ry", [1], lambda P: 2 * P[:, 0], [0]),
            _g("rz", [1], lambda P: 2 * P[:, 0], [0]),
            _g("cx", [0, 1]),
            _g("ry", [1], lambda P: 2 * P[:, 0], [0]),
        ]
    return ops


def genetic_featuremap_highres_2(repeats=1):
    """reference quantum_featuremaps.py:306-317"""
    ops = []
    for _ in range(repeats):
        ops += [
            _g("h", [0]),
            _g("rx", [1], lambda P: np.arcsin(2 / PI * P[:, 1]), [1]),
            _g("rz", [0], lambda P: np.arccos(0.4 * P[:, 1] * P[:, 1]), [1]),
            _g("rz", [1], lambda P: 2 * P[:, 0], [0]),
            _g("p", [0], lambda P: np.arcsin(2 / PI * P[:, 0]), [0]),
        ]
    return ops


def genetic_attention(repeats=1):
    """reference quantum_featuremaps.py:342-352"""
    ops = []
    for _ in range(repeats):
        ops += [
            _g("rx", [0], lambda P: np.arccos(0.4 * P[:, 0] * P[:, 0]), [0]),
            _g("rx", [1], lambda P: np.arccos(2 / PI * P[:, 0]), [0]),
            _g("rz", [1], lambda P: 4 * P[:, 0] * P[:, 1], [0, 1]),
        ]
    return ops


def genetic_attention_highres(repeats=1):
    """reference quantum_featuremaps.py:377-387"""
    ops = []
    for _ in range(repeats):
        ops += [
            _g("rx", [0], lambda P: 4 * P[:, 0] * P[:, 1], [0, 1]),
            _g("cx", [0, 1]),
            _g("ry", [0], lambda P: np.arcsin(2 / PI * P[:, 1]), [1]),
            _g("rx", [1], lambda P: 2 * P[:, 0], [0]),
        ]
    return ops


def to_quantum(genes, gate_dict, nb_features, gates_per_qubits, nb_qubits,
               param_shift=0, physical_qubits=(0, 1, 2, 3), coupling_map=None, variant="v3"):
    """Genome decoder -> (ops, x_idxs).  reference genetic_v3.py:358-448
    (v4: genetic_v4.py:256-347, v5: genetic_v5.py:256-344)."""
    gate_list = []
    for s in gate_dict.values():
        gate_list += list(s)
    g = np.reshape(np.asarray(genes), [gates_per_qubits, nb_qubits, 5])
    ops, x_idxs = [], []
    for j in range(gates_per_qubits):
        for k in range(nb_qubits):
            gi, t, m, f1, f2 = (int(v) for v in g[j, k])
            f1 += param_shift
            f2 += param_shift
            name = gate_list[gi]
            if name in gate_dict["single_non_parametric"]:
                if name != "I":
                    ops.append(_g(name.lower(), [k]))
            elif name in gate_dict["two_non_parametric"]:
                pair = None
                if variant == "v3":
                    if k in (0, 1):
                        pair = (k, k + 1)
                    elif k == 3:
                        pair = (3, 2)
                    else:
                        physical_qubits[k]  # the reference's message indexes this list
                elif variant == "v4":
                    if k in (0, 1):
                        if (physical_qubits[k], physical_qubits[k + 1]) in coupling_map:
                            pair = (k, k + 1)
                    elif k == 3:
                        if (physical_qubits[k], physical_qubits[k - 1]) in coupling_map:
                            pair = (3, 2)
                    if pair is None:
                        physical_qubits[k]
                else:  # v5
                    if k < 3:
                        a, b = physical_qubits[k], physical_qubits[k + 1]
                        if [a, b] in coupling_map or (a, b) in coupling_map:
                            pair = (k, k + 1)
                    elif k == 3:
                        a, b = physical_qubits[3], physical_qubits[2]
                        if [a, b] in coupling_map or (a, b) in coupling_map:
                            pair = (3, 2)
                if pair is not None:
                    ops.append(_g(name.lower(), pair))
            else:
                if f1 not in x_idxs:
                    x_idxs.append(f1)
                if m == 1 and f2 not in x_idxs:
                    x_idxs.append(f2)
                if m == 0 and t == 0:
                    fn, used = (lambda P, a=f1: 2 * PI * (P[:, a] - 0.5)), [f1]
                elif m == 1 and t == 0:
                    fn, used = (lambda P, a=f1, b=f2: 2 * PI * P[:, a] * (1 - P[:, b]) - PI), [f1, f2]
                elif m == 0 and t == 1:
                    fn, used = (lambda P, a=f1: 2 * PI * P[:, a] * (1 - P[:, a]) - PI), [f1]
                elif m == 1 and t == 1:
                    fn, used = (lambda P, a=f1, b=f2: (2 * PI * P[:, a] * (1 - P[:, b]) - PI)
                                * (2 * PI * P[:, b] * (1 - P[:, a]) - PI) / PI), [f1, f2]
                elif m == 0 and t == 2:
                    fn, used = (lambda P, a=f1: 2 * np.arcsin(2 * P[:, a] - 1) - PI), [f1]
                elif m == 1 and t == 2:
                    fn, used = (lambda P, a=f1, b=f2: 2 * np.arcsin((2 * P[:, a] - 1) * (2 * P[:, b] - 1))), [f1, f2]
                else:
                    raise ValueError("invalid gene")
                if name in gate_dict["single_parametric"]:
                    ops.append(_g(name.lower(), [k], fn, used))
    return ops, x_idxs


# --------------------------------------------------------------------------- #
# binding (Qiskit assign_parameters)
# --------------------------------------------------------------------------- #
def used_parameters(ops):
    """Sorted indices of the parameters that reach a gate (= circuit.parameters)."""
    s = set()
    for _, _, _, used in ops:
        s.update(used)
    return sorted(s)


def bind(ops, X):
    """Positional binding: column k of X -> k-th parameter in sorted-index order.
    Returns P with P[:, i] = value of x[i].  Raises ValueError on a column-count
    mismatch like FidelityStatevectorKernel._validate_input."""
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X.reshape(1, -1)
    params = used_parameters(ops)
    if X.shape[1] != len(params):
        raise ValueError(
            f"x_vec and class feature map have incompatible dimensions: "
            f"{X.shape[1]} columns, {len(params)} parameters")
    width = (max(params) + 1) if params else 0
    P = np.zeros((X.shape[0], width), dtype=np.float64)
    for k, i in enumerate(params):
        P[:, i] = X[:, k]
    return P


# --------------------------------------------------------------------------- #
# statevector simulation (Qiskit Statevector conventions)
# --------------------------------------------------------------------------- #
def _mat1(name, theta):
    """(N, 2, 2) complex128 gate matrices; theta is (N,) or None."""
    if name == "h":
        return np.array([[[SQ2, SQ2], [SQ2, -SQ2]]], dtype=np.complex128)
    if name == "x":
        return np.array([[[0, 1], [1, 0]]], dtype=np.complex128)
    if name == "sx":
        return 0.5 * np.array([[[1 + 1j, 1 - 1j], [1 - 1j, 1 + 1j]]], dtype=np.complex128)
    if name in ("id", "i"):
        return np.array([[[1, 0], [0, 1]]], dtype=np.complex128)
    c, s = np.cos(theta / 2), np.sin(theta / 2)
    m = np.zeros((len(theta), 2, 2), dtype=np.complex128)
    if name == "rx":
        m[:, 0, 0] = c; m[:, 0, 1] = -1j * s; m[:, 1, 0] = -1j * s; m[:, 1, 1] = c
    elif name == "ry":
        m[:, 0, 0] = c; m[:, 0, 1] = -s; m[:, 1, 0] = s; m[:, 1, 1] = c
    elif name == "rz":
        m[:, 0, 0] = np.exp(-0.5j * theta); m[:, 1, 1] = np.exp(0.5j * theta)
    elif name == "p":
        m[:, 0, 0] = 1; m[:, 1, 1] = np.exp(1j * theta)
    else:
        raise ValueError(f"unsupported gate {name}")
    return m


# basis index of a two-qubit matrix: 2*bit(q1) + bit(q0) for gate(q0, q1)
_CX = np.array([[1, 0, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0], [0, 1, 0, 0]], dtype=np.complex128)
_CZ = np.diag([1, 1, 1, -1]).astype(np.complex128)
_ECR = SQ2 * np.array([[0, 1, 0, 1j], [1, 0, -1j, 0], [0, 1j, 0, 1], [-1j, 0, 1, 0]], dtype=np.complex128)
_MAT2 = {"cx": _CX, "cz": _CZ, "ecr": _ECR}


def simulate(ops, n, P):
    """Psi (N x 2^n complex128): |0..0> evolved gate by gate; qubit 0 = LSB."""
    N = P.shape[0]
    psi = np.zeros((N, 1 << n), dtype=np.complex128)
    psi[:, 0] = 1.0
    for name, qubits, fn, _ in ops:
        if len(qubits) == 1:
            q = qubits[0]
            m = _mat1(name, fn(P) if fn is not None else None)
            v = psi.reshape(N, 1 << (n - 1 - q), 2, 1 << q)
            # new[s, a, r, b] = sum_c m[s, r, c] v[s, a, c, b]
            psi = np.einsum("src,sacb->sarb", np.broadcast_to(m, (N, 2, 2)), v).reshape(N, 1 << n)
        else:
            q0, q1 = qubits
            m = _MAT2[name].reshape(2, 2, 2, 2)  # [b1', b0', b1, b0]
            lo, hi = min(q0, q1), max(q0, q1)
            v = psi.reshape(N, 1 << (n - 1 - hi), 2, 1 << (hi - lo - 1), 2, 1 << lo)
            if q1 > q0:   # axis 2 = bit(q1), axis 4 = bit(q0)
                psi = np.einsum("xyzw,sazbwc->saxbyc", m, v).reshape(N, 1 << n)
            else:         # axis 2 = bit(q0), axis 4 = bit(q1)
                psi = np.einsum("xyzw,sawbzc->saybxc", m, v).reshape(N, 1 << n)
    return np.ascontiguousarray(psi)


def statevectors(ops, n, X):
    return simulate(ops, n, bind(ops, X))


# --------------------------------------------------------------------------- #
# kernels
# --------------------------------------------------------------------------- #
def fidelity_kernel(psi_x, psi_y=None, enforce_psd=False):
    """K[i, j] = |<psi_x[i] | psi_y[j]>|^2 (FidelityStatevectorKernel._compute_fidelity,
    SURVEY.md 8a A5).  Symmetric case: exact unit diagonal; optional PSD projection
    (qiskit_machine_learning's _make_psd) for small N."""
    sym = psi_y is None
    if sym:
        psi_y = psi_x
    K = np.abs(psi_x.conj() @ psi_y.T) ** 2
    # "if np.array_equal(x_sv, y_sv): continue" -- entries of equal statevectors stay exactly 1.0,
    # on the diagonal, for duplicate rows and for rows shared by x_vec and y_vec
    kx = [r.tobytes() for r in np.ascontiguousarray(psi_x)]
    ky = kx if sym else [r.tobytes() for r in np.ascontiguousarray(psi_y)]
    where = {}
    for j, k in enumerate(ky):
        where.setdefault(k, []).append(j)
    if sym or set(kx) & set(where):
        for i, k in enumerate(kx):
            for j in where.get(k, ()):
                K[i, j] = 1.0
    if sym:
        np.fill_diagonal(K, 1.0)
        if enforce_psd:
            w, v = np.linalg.eig(K)
            K = (v @ np.diag(np.maximum(w, 0)) @ v.T).real
    return K


def evaluate(ops, n, x_vec, y_vec=None, enforce_psd=False):
    px = statevectors(ops, n, x_vec)
    py = None if y_vec is None else statevectors(ops, n, y_vec)
    return fidelity_kernel(px, py, enforce_psd)


def rdm1(psi, n):
    """Per-qubit reduced density matrices, (N, n, 2, 2) complex128
    (qiskit.quantum_info.partial_trace of |psi><psi| onto qubit q)."""
    N = psi.shape[0]
    out = np.zeros((N, n, 2, 2), dtype=np.complex128)
    for q in range(n):
        v = psi.reshape(N, 1 << (n - 1 - q), 2, 1 << q)
        out[:, q] = np.einsum("sarb,sacb->src", v, v.conj())
    return out


def rdm1_explicit(psi_row, n):
    """Same from the explicit 4^n density matrix (check for small n)."""
    rho = np.outer(psi_row, psi_row.conj())
    out = []
    for q in range(n):
        r = rho.reshape([2] * (2 * n))
        # row bits are axes 0..n-1 (MSB first), column bits n..2n-1
        ax_r, ax_c = n - 1 - q, 2 * n - 1 - q
        keep = np.zeros((2, 2), dtype=np.complex128)
        for a in range(2):
            for b in range(2):
                sl = [slice(None)] * (2 * n)
                sl[ax_r], sl[ax_c] = a, b
                sub = r[tuple(sl)].reshape(1 << (n - 1), 1 << (n - 1))
                keep[a, b] = np.trace(sub)
        out.append(keep)
    return np.array(out)


def projected_kernel(ops, n, X, gamma):
    """reference genetic_v3.py:789-806: K_ij = exp(-gamma * sum_q ||rho_i^q - rho_j^q||_F),
    all-ones when the circuit has no parameters."""
    X = np.asarray(X, dtype=np.float64)
    N = X.shape[0]
    if not used_parameters(ops):
        return np.ones((N, N))
    r = rdm1(statevectors(ops, n, X), n)
    K = np.zeros((N, N))
    for i in range(N):
        d = r[i][None] - r[:i]                       # (i, n, 2, 2)
        fro = np.sqrt((np.abs(d) ** 2).sum(axis=(2, 3)))   # un-squared Frobenius norm per qubit
        K[i, :i] = np.exp(-gamma * fro.sum(axis=1))
    return K + K.T + np.identity(N)


# --------------------------------------------------------------------------- #
# flat export for the C twin (oracle/qk_oracle.c): numeric angles per sample
# --------------------------------------------------------------------------- #
_CODES = {"id": 0, "i": 0, "h": 1, "x": 2, "sx": 3, "rx": 4, "ry": 5, "rz": 6, "p": 7, "cx": 8, "cz": 9, "ecr": 10}


def flatten(ops, P):
    """(gate int32 [n_ops, 3] = code, q0, q1 ; angles float64 [N, n_ops])"""
    g = np.zeros((len(ops), 3), dtype=np.int32)
    ang = np.zeros((P.shape[0], len(ops)), dtype=np.float64)
    for k, (name, qubits, fn, _) in enumerate(ops):
        g[k] = (_CODES[name], qubits[0], qubits[1] if len(qubits) > 1 else -1)
        if fn is not None:
            ang[:, k] = fn(P)
    return g, np.ascontiguousarray(ang)
