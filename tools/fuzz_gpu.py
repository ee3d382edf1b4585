#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (not part of the pytest suite: minutes, not seconds).
Random circuits over the whole gate alphabet at 1..15 qubits against the numpy oracle, and random
Gram shapes (ragged N / M, strided K, symmetric / cross, multi-job, both precisions)."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import qk_oracle as O
from quake_amd import circuit as qc, device as qdev
from quake_amd.kernels import HipFidelityKernel
from tests._helpers import psi_from_compiled

ap = argparse.ArgumentParser()
ap.add_argument("--circuits", type=int, default=150)
ap.add_argument("--grams", type=int, default=150)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
worst = worst_rdm = 0.0
worst_rdm_case = None
for t in range(a.circuits):
    n = int(rng.integers(1, 16))
    F = int(rng.integers(1, 6))
    x = qc.ParameterVector("x", F)
    c = qc.QuantumCircuit(n)
    for _ in range(int(rng.integers(1, 10 * n + 2))):
        g = int(rng.integers(0, 12))
        q = int(rng.integers(0, n))
        r = int(rng.integers(0, n))
        ang = x[int(rng.integers(0, F))] * float(rng.uniform(0.3, 2.0)) + float(rng.uniform(-1, 1))
        if g >= 7 and (n == 1 or q == r): g = g % 7
        if g == 0: c.h(q)
        elif g == 1: c.sx(q)
        elif g == 2: c.x(q)
        elif g == 3: c.rx(ang, q)
        elif g == 4: c.ry(ang, q)
        elif g == 5: c.rz(ang, q)
        elif g == 6: c.p(ang, q)
        elif g == 7: c.cx(q, r)
        elif g == 8: c.cz(q, r)
        elif g == 9: c.ecr(q, r)
        elif g == 10: c.id(q)
        else:
            c.cx(q, r); c.p(ang, r); c.cx(q, r)
    if c.num_parameters == 0: continue
    N = int(rng.integers(1, 6 if n > 12 else 40))
    X = rng.uniform(-1.5, 1.5, size=(N, c.num_parameters))
    got = HipFidelityKernel(c).statevectors(X)
    ref = psi_from_compiled(c, X)
    err = float(np.abs(got - ref).max())
    worst = max(worst, err)
    assert err < 1e-12, ("circuit", t, n, err)
    # reduced density matrices: fused with the simulation when the circuit is one LDS-resident pass
    r = O.rdm1(ref, n)
    want = np.stack([r[..., 0, 0].real, r[..., 1, 1].real, r[..., 0, 1].real, r[..., 0, 1].imag], axis=-1)
    rd = qdev.simulate_rdm1(qdev.DeviceCircuit(c), torch.from_numpy(X).cuda()).cpu().numpy()
    e2 = float(np.abs(rd - want).max())
    if e2 > worst_rdm:
        worst_rdm, worst_rdm_case = e2, (t, n, N, float(np.abs(ref).max()), float(np.abs(np.linalg.norm(ref, axis=1) - 1).max()))
    assert e2 < 1e-12, ("rdm", t, n, e2)
print("circuits ok, worst |dpsi| =", worst, "worst |drdm| =", worst_rdm, worst_rdm_case, flush=True)

worst64 = worst32 = 0.0
for t in range(a.grams):
    n = int(rng.integers(0, 11))
    D = 1 << n
    N, M = int(rng.integers(1, 700)), int(rng.integers(1, 500))
    def rows(k):
        v = rng.normal(size=(k, D)) + 1j * rng.normal(size=(k, D))
        return v / np.linalg.norm(v, axis=1, keepdims=True)
    A, B = rows(N), rows(M)
    for dtype, tol in ((torch.complex128, 1e-12), (torch.complex64, 1e-5)):
        Ad, Bd = torch.from_numpy(A).to(dtype).cuda(), torch.from_numpy(B).to(dtype).cuda()
        pad = int(rng.integers(0, 9))
        Ks = torch.full((N, N + pad), -7.0, dtype=torch.float64, device="cuda")
        Kc = torch.full((N, M + pad), -7.0, dtype=torch.float64, device="cuda")
        KT = torch.full((M, N + pad), -7.0, dtype=torch.float64, device="cuda")
        qdev.gram_jobs([dict(A=Ad, B=Ad, K=Ks[:, :N], symmetric=True),
                        dict(A=Ad, B=Bd, K=Kc[:, :M], KT=KT[:, :N])], D)
        ref_s, ref_c = np.abs(A.conj() @ A.T) ** 2, np.abs(A.conj() @ B.T) ** 2
        np.fill_diagonal(ref_s, 1.0)
        ks, kc, kt = Ks.cpu().numpy(), Kc.cpu().numpy(), KT.cpu().numpy()
        e = max(np.abs(ks[:, :N] - ref_s).max(), np.abs(kc[:, :M] - ref_c).max())
        assert e < tol, ("gram", t, n, N, M, dtype, e)
        assert np.array_equal(ks[:, :N], ks[:, :N].T) and np.array_equal(kt[:, :N], kc[:, :M].T)
        assert (ks[:, N:] == -7).all() and (kc[:, M:] == -7).all() and (kt[:, N:] == -7).all()
        if dtype == torch.complex128: worst64 = max(worst64, e)
        else: worst32 = max(worst32, e)
print("grams ok, worst |dK| fp64 =", worst64, "fp32 =", worst32, flush=True)
