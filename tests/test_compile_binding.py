"""``QuantumCircuit.compile()`` -- sorted positional binding + the postfix angle compiler -- pinned
independently of the product IR.

Every circuit here is built TWICE from one random recipe: as a product circuit (``ParameterExpression``
arithmetic over a ``ParameterVector``) and as an oracle op list whose angle functions are plain numpy
lambdas over the parameter table the ORACLE's own ``bind`` produces (column k -> k-th parameter in
sorted-index order).  Nothing of ``quake_amd.circuit`` is on the oracle side, so a
binding or lowering bug in an unusual expression form (nested products, a parameter used twice, unused
and offset parameters, every ``qk_expr_op``) cannot cancel out.
"""
import numpy as np
import pytest

from _helpers import run_postfix
from oracle import qk_oracle as O
from quake_amd import circuit as qc


class _Recipe:
    """Random expression trees as (product expression, numpy function of P, parameter indices used)."""

    def __init__(self, rng, x, usable):
        self.rng, self.x, self.usable = rng, x, list(usable)

    def leaf(self):
        if self.rng.random() < 0.3:
            c = float(self.rng.uniform(-2, 2))
            return c, (lambda P, c=c: np.full(P.shape[0], c)), set()
        i = int(self.rng.choice(self.usable))
        return self.x[i], (lambda P, i=i: P[:, i]), {i}

    def bounded(self, depth):
        """an expression with values in [-1, 1] (an asin / acos argument)"""
        e, f, u = self.tree(depth)
        kind = int(self.rng.integers(0, 2))
        if kind == 0:
            return np.sin(e) * 0.95 if not isinstance(e, float) else float(np.sin(e) * 0.95), \
                (lambda P: np.sin(f(P)) * 0.95), u
        return np.cos(e) if not isinstance(e, float) else float(np.cos(e)), (lambda P: np.cos(f(P))), u

    def tree(self, depth):
        if depth == 0 or self.rng.random() < 0.15:
            return self.leaf()
        op = int(self.rng.integers(0, 9))
        if op <= 3:
            (a, fa, ua), (b, fb, ub) = self.tree(depth - 1), self.tree(depth - 1)
            if op == 0:
                return a + b, (lambda P: fa(P) + fb(P)), ua | ub
            if op == 1:
                return a - b, (lambda P: fa(P) - fb(P)), ua | ub
            if op == 2:
                return a * b, (lambda P: fa(P) * fb(P)), ua | ub
            den, fd = (b * b + 1.5), (lambda P: fb(P) * fb(P) + 1.5)      # never near zero
            return a / den, (lambda P: fa(P) / fd(P)), ua | ub
        a, fa, ua = self.tree(depth - 1)
        if op == 4:
            return -a, (lambda P: -fa(P)), ua
        if op == 5:
            if isinstance(a, float):
                return float(np.sin(a)), (lambda P: np.sin(fa(P))), ua
            return np.sin(a), (lambda P: np.sin(fa(P))), ua
        if op == 6:
            if isinstance(a, float):
                return float(np.cos(a)), (lambda P: np.cos(fa(P))), ua
            return np.cos(a), (lambda P: np.cos(fa(P))), ua
        b, fb, ub = self.bounded(depth - 1)
        if isinstance(b, float):
            v = float(np.arcsin(b) if op == 7 else np.arccos(b))
            return v, (lambda P, v=v: np.full(P.shape[0], v)), ub
        if op == 7:
            return np.arcsin(b), (lambda P: np.arcsin(fb(P))), ub
        return np.arccos(b), (lambda P: np.arccos(fb(P))), ub


def _random_pair(seed, n, n_gates, vec_len, shift):
    """(product circuit, oracle ops).  Parameters x[shift ..] of a vector of ``vec_len`` are usable; some
    never reach a gate (unused), some reach several (repeated)."""
    rng = np.random.default_rng(seed)
    x = qc.ParameterVector("x", vec_len)
    usable = rng.choice(np.arange(shift, vec_len), size=max(2, (vec_len - shift) * 2 // 3), replace=False)
    recipe = _Recipe(rng, x, usable)
    circ, ops = qc.QuantumCircuit(n), []
    for _ in range(n_gates):
        kind = rng.choice(["rx", "ry", "rz", "p", "h", "sx", "x", "cx", "cz", "ecr", "id"],
                          p=[.16, .16, .12, .16, .08, .05, .03, .08, .06, .05, .05])
        q = int(rng.integers(0, n))
        if kind in ("cx", "cz", "ecr"):
            if n < 2:
                continue
            q2 = int((q + 1 + rng.integers(0, n - 1)) % n)
            getattr(circ, kind)(q, q2)
            ops.append(O._g(kind, [q, q2]))
        elif kind in ("rx", "ry", "rz", "p"):
            e, f, used = recipe.tree(int(rng.integers(1, 4)))
            while not used:                        # a constant angle has no parameter: make it depend on one
                e2, f2, used = recipe.leaf()
                if used:
                    e, f = e2 * 0.5 + e, (lambda P, f=f, f2=f2: f2(P) * 0.5 + f(P))
            getattr(circ, kind)(e, q)
            ops.append(O._g(kind, [q], f, sorted(used)))
        else:
            getattr(circ, kind)(q)
            ops.append(O._g(kind, [q]))
    return circ, ops


CASES = [(s, n, g, L, sh) for s, (n, g, L, sh) in enumerate(
    [(1, 6, 3, 0), (2, 10, 5, 0), (3, 14, 7, 2), (4, 18, 9, 0), (5, 24, 12, 3), (3, 12, 18, 5), (6, 30, 8, 1),
     (8, 40, 10, 0), (10, 40, 16, 4), (11, 30, 6, 0), (12, 30, 14, 2), (13, 20, 9, 1)])]


@pytest.mark.parametrize("seed,n,gates,vec_len,shift", CASES[:8])
def test_compiled_programs_equal_oracle_angles(seed, n, gates, vec_len, shift):
    """CPU: every compiled postfix program, evaluated over the DATA COLUMNS, gives the angle the oracle's
    lambda gives over the parameter table of the oracle's own bind"""
    circ, ops = _random_pair(100 + seed, n, gates, vec_len, shift)
    comp = circ.compile()
    used = O.used_parameters(ops)
    assert comp.n_params == len(used)
    assert [p.index for p in circ.parameters] == used            # sorted-index order, unused parameters absent
    X = np.random.default_rng(seed).uniform(-1.4, 1.4, size=(17, comp.n_params))
    P = O.bind(ops, X)
    angle_ops = [op for op in ops if op[2] is not None]
    assert len(angle_ops) == len(comp.expr_off) - 1
    k = 0
    for gate, q0, q1, e in comp.ops:
        if e < 0:
            continue
        prog = [tuple(map(int, r)) for r in comp.prog[comp.expr_off[e]:comp.expr_off[e + 1]]]
        got = run_postfix(prog, comp.consts, X)
        want = angle_ops[k][2](P)
        assert np.abs(got - want).max() < 1e-12, (k, prog)
        k += 1
    # every opcode of the ABI appears somewhere in the suite's programs (checked over all cases once)


def test_every_expression_opcode_is_exercised():
    seen = set()
    for seed, n, gates, vec_len, shift in CASES:
        comp = _random_pair(100 + seed, n, gates, vec_len, shift)[0].compile()
        seen |= set(int(o) for o in comp.prog[:, 0])
    assert seen == set(range(11)), seen


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,gates,vec_len,shift", CASES)
def test_statevectors_of_random_expressions_match_oracle(seed, n, gates, vec_len, shift):
    """GPU: product circuit through compile() + the C ABI against the oracle op list through the oracle's bind"""
    from quake_amd.kernels import HipFidelityKernel
    circ, ops = _random_pair(100 + seed, n, gates, vec_len, shift)
    F = len(O.used_parameters(ops))
    X = np.random.default_rng(seed).uniform(-1.4, 1.4, size=(33, F))
    psi = HipFidelityKernel(circ).statevectors(X)
    want = O.statevectors(ops, n, X)
    assert np.abs(psi - want).max() < 1e-12
    with pytest.raises(ValueError):
        HipFidelityKernel(circ).evaluate(np.zeros((3, F + 1)))
