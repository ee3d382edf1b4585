"""Simulator parity at 18-22 qubits (multi-pass, many chunk bits) against the numpy oracle."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import qk_oracle as O
from quake_amd import circuit as qc, featuremaps as fm
from quake_amd.kernels import HipFidelityKernel
from tests._helpers import psi_from_compiled
for n in (18, 20, 22):
    x = qc.ParameterVector("x", n)
    c = fm.zz_featuremap(x, n, 2)
    X = np.random.default_rng(n).uniform(-1.5, 1.5, size=(2, n))
    k = HipFidelityKernel(c)
    got = k.statevectors(X)
    ref = O.statevectors(O.zz_featuremap(n, 2), n, X)
    print(n, "zz_r2 passes", k._circuit().num_passes, "err", np.abs(got - ref).max(), flush=True)
    rng = np.random.default_rng(100 + n)
    c = qc.QuantumCircuit(n)
    xs = qc.ParameterVector("x", 4)
    for _ in range(6 * n):
        g = rng.integers(0, 9); q, r = (int(v) for v in rng.choice(n, 2, replace=False))
        a = xs[int(rng.integers(0, 4))] * 1.3
        [lambda: c.h(q), lambda: c.sx(q), lambda: c.rx(a, q), lambda: c.ry(a, q), lambda: c.rz(a, q), lambda: c.p(a, q),
         lambda: c.cx(q, r), lambda: c.cz(q, r), lambda: c.ecr(q, r)][g]()
    X = rng.uniform(-1.5, 1.5, size=(1, 4))
    k = HipFidelityKernel(c)
    got = k.statevectors(X)
    print(n, "random passes", k._circuit().num_passes, "err", np.abs(got - psi_from_compiled(c, X)).max(), flush=True)
