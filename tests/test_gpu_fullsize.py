"""BASELINE.json's full sizes on one MI355X, checked through properties that do not need an oracle of
that size: exact symmetry and unit diagonal, the range of a fidelity, agreement of two independent
kernel paths (symmetric job with mirrored tiles vs plain cross job), the closed form of the ZZ map,
and parity with the oracle on rows sampled across the whole matrix (first, last, tile borders)."""
import numpy as np
import pytest

from oracle import qk_oracle as O
from quake_amd import featuremaps as fm
from quake_amd.circuit import ParameterVector

pytestmark = pytest.mark.gpu

LO, HI = -np.pi / 2 + 1e-5, np.pi / 2 - 1e-5


def _symmetric_and_unit_diagonal(K, torch, block=4096):
    N = K.shape[0]
    assert bool((torch.diagonal(K) == 1.0).all())
    for r0 in range(0, N, block):
        r1 = min(N, r0 + block)
        assert bool((K[r0:r1] == K[:, r0:r1].T).all()), (r0, r1)     # mirrored entries are bitwise copies


def _sample_rows(N):
    rows = [0, 1, 63, 64, 127, 128, 129, 4095, 4096, N // 2 - 1, N // 2, N // 2 + 77, N - 130, N - 129, N - 128, N - 2, N - 1]
    return np.array(sorted(set(r for r in rows if 0 <= r < N)))


def test_config5_headline_size():
    """configs[4] / the north star: N = 65536, 16 qubits, ZZ r = 1, train x train (Psi 64 GiB, K 32 GiB)."""
    import torch
    from quake_amd import device as qdev
    N, n = 65536, 16
    D = 1 << n
    free, _ = torch.cuda.mem_get_info()
    if free < 110 * 2**30:
        pytest.skip("needs ~100 GiB of free HBM")
    X_host = np.random.default_rng(1005).uniform(LO, HI, size=(N, n))
    X = torch.from_numpy(X_host).cuda()
    circ = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 1))
    psi = circ.simulate(X)
    # closed form of the r = 1 map: every amplitude has modulus 2^(-n/2), so every row has norm 1
    mod = torch.view_as_real(psi[::997]).pow(2).sum(-1)
    assert float((mod - 1.0 / D).abs().max()) < 1e-18
    rows = _sample_rows(N)
    ref_psi = O.statevectors(O.zz_featuremap(n, 1), n, X_host[rows])
    assert np.abs(psi[torch.from_numpy(rows).cuda()].cpu().numpy() - ref_psi).max() < 1e-12
    K = qdev.gram(psi)
    torch.cuda.synchronize()
    _symmetric_and_unit_diagonal(K, torch)
    assert float(K.min()) >= 0.0 and float(K.max()) <= 1.0 + 1e-12
    # oracle parity on the sampled rows x sampled columns (entries from diagonal, upper and mirrored tiles)
    ref = np.abs(ref_psi.conj() @ ref_psi.T) ** 2
    np.fill_diagonal(ref, 1.0)
    idx = torch.from_numpy(rows).cuda()
    got = K[idx][:, idx].cpu().numpy()
    assert np.abs(got - ref).max() < 1e-12
    # independent path: the same rows as a plain cross block (no symmetry, no mirroring, other tile walk)
    band = torch.arange(N - 1024 - 37, N - 37, device="cuda")
    Kc = qdev.gram(psi[band[0]:band[-1] + 1], psi)
    d = (Kc - K[band[0]:band[-1] + 1]).abs()
    d[torch.arange(band.numel(), device="cuda"), band] = 0.0       # the symmetric job pins its diagonal to exactly 1
    assert float(d.max()) < 1e-13
    del K, Kc, d
    torch.cuda.empty_cache()
    # configs[4]'s second half: the 16384 x 65536 test x train block, all of it computed, checked on sampled
    # test rows x sampled train rows (matrix corners, tile borders) against the oracle
    T = 16384
    Xt_host = np.random.default_rng(2005).uniform(LO, HI, size=(T, n))
    psi_t = circ.simulate(torch.from_numpy(Xt_host).cuda())
    Kt = qdev.gram(psi_t, psi)
    torch.cuda.synchronize()
    assert tuple(Kt.shape) == (T, N) and float(Kt.min()) >= 0.0 and float(Kt.max()) <= 1.0 + 1e-12
    trows = _sample_rows(T)
    ref_t = O.statevectors(O.zz_featuremap(n, 1), n, Xt_host[trows])
    want = np.abs(ref_t.conj() @ ref_psi.T) ** 2
    got = Kt[torch.from_numpy(trows).cuda()][:, idx].cpu().numpy()
    assert np.abs(got - want).max() < 1e-12
    # every row of the block sums to what a second pass over the same operands gives (no tile skipped or doubled):
    # sum_j K[t, j] recomputed from a transposed job (train x test) on a band of columns
    Ktt = qdev.gram(psi[N - 2048:], psi_t)                       # [2048, T]
    assert float((Ktt.T - Kt[:, N - 2048:]).abs().max()) < 1e-13
    del Kt, Ktt, psi_t, psi
    torch.cuda.empty_cache()


def test_host_entry_point_bands_and_mirror():
    """HipFidelityKernel.evaluate (numpy in, page-locked numpy out) at 32768 rows: the host path computes K in
    16 row bands, copies each to the host while the next is computed, and fills the mirror entries of later
    bands from transposed copies.  Properties on the whole matrix, oracle parity on sampled rows."""
    from quake_amd.kernels import HipFidelityKernel
    N, n = 32768, 10
    X = np.random.default_rng(1010).uniform(LO, HI, size=(N, n))
    X[N - 5] = X[7]                                              # a duplicate sample: entry pinned to exactly 1
    K = HipFidelityKernel(fm.zz_featuremap(ParameterVector("x", n), n, 2)).evaluate(X)
    assert K.shape == (N, N) and K.flags["C_CONTIGUOUS"]
    assert (np.diagonal(K) == 1.0).all() and K[N - 5, 7] == 1.0 and K[7, N - 5] == 1.0
    for r0 in range(0, N, 4096):
        assert np.array_equal(K[r0:r0 + 4096], K[:, r0:r0 + 4096].T), r0
    rows = _sample_rows(N)
    ref_psi = O.statevectors(O.zz_featuremap(n, 2), n, X[rows])
    ref = np.abs(ref_psi.conj() @ ref_psi.T) ** 2
    np.fill_diagonal(ref, 1.0)
    assert np.abs(K[np.ix_(rows, rows)] - ref).max() < 1e-12
    # a cross call (test x train) through the same banded path
    Y = np.random.default_rng(1011).uniform(LO, HI, size=(3000, n))
    Kc = HipFidelityKernel(fm.zz_featuremap(ParameterVector("x", n), n, 2)).evaluate(Y, X)
    ref_y = O.statevectors(O.zz_featuremap(n, 2), n, Y[:40])
    assert np.abs(Kc[:40][:, rows] - np.abs(ref_y.conj() @ ref_psi.T) ** 2).max() < 1e-12


def test_config4_projected_kernel_full_size():
    """configs[3]: projected kernel, N = 32768, 12 qubits (K 8 GiB), fused simulate + RDM."""
    import torch
    from quake_amd import device as qdev
    N, n = 32768, 12
    X_host = np.random.default_rng(1004).uniform(LO, HI, size=(N, n))
    X = torch.from_numpy(X_host).cuda()
    circ = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 1))
    rdm = qdev.simulate_rdm1(circ, X)
    assert float((rdm[..., 0] + rdm[..., 1] - 1.0).abs().max()) < 1e-13        # unit trace per qubit
    K = qdev.pqk(rdm, None, 1.0)
    torch.cuda.synchronize()
    _symmetric_and_unit_diagonal(K, torch)
    assert float(K.min()) > 0.0 and float(K.max()) <= 1.0
    rows = _sample_rows(N)
    ops = O.zz_featuremap(n, 1)
    ref = O.projected_kernel(ops, n, X_host[rows], 1.0)
    idx = torch.from_numpy(rows).cuda()
    assert np.abs(K[idx][:, idx].cpu().numpy() - ref).max() < 1e-12
    Kc = qdev.pqk(rdm[N - 900:N - 100], rdm, 1.0)                               # cross path, no mirroring
    d = (Kc - K[N - 900:N - 100]).abs()
    d[torch.arange(800, device="cuda"), torch.arange(N - 900, N - 100, device="cuda")] = 0.0
    assert float(d.max()) < 1e-14
    del K, Kc, d
    torch.cuda.empty_cache()


def test_config2_and_config3_full_size():
    """configs[1] (N = 4096, 8 qubits, ZZ r = 2) against the oracle entry by entry, and one genome of
    configs[2] (N = 16384, 10 qubits) through properties + sampled rows."""
    import torch
    from quake_amd import device as qdev, genetic
    N, n = 4096, 8
    X_host = np.random.default_rng(1002).uniform(LO, HI, size=(N, n))
    circ = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 2))
    K = qdev.gram(circ.simulate(torch.from_numpy(X_host).cuda()))
    assert np.abs(K.cpu().numpy() - O.evaluate(O.zz_featuremap(n, 2), n, X_host)).max() < 1e-12
    N, Q, F, G = 16384, 10, 18, 8
    gd = {"single_non_parametric": ["Id", "H", "X", "SX"], "single_parametric": ["RX", "RY", "RZ"],
          "two_non_parametric": ["CX"], "two_parametric": []}
    rng = np.random.default_rng(1003)
    Xall = rng.uniform(1e-6, 1 - 1e-6, size=(N, F))
    genes = genetic.initial_population(4, Q, G, gd, F, rng=rng)[1]
    cmap = [[q, q + 1] for q in range(Q - 1)] + [[q + 1, q] for q in range(Q - 1)]
    fmap, xi = genetic.to_quantum(genes, gd, F, G, Q, coupling_map=cmap, variant="v5", physical_qubits=list(range(Q)), verbose=False)
    oops, oxi = O.to_quantum(genes, gd, F, G, Q, coupling_map=cmap, variant="v5", physical_qubits=list(range(Q)))
    assert list(xi) == list(oxi)
    circ = qdev.DeviceCircuit(fmap)
    psi = circ.simulate(torch.from_numpy(np.ascontiguousarray(Xall[:, xi])).cuda())
    K = qdev.gram(psi)
    torch.cuda.synchronize()
    _symmetric_and_unit_diagonal(K, torch)
    rows = _sample_rows(N)
    ref_psi = O.statevectors(oops, Q, Xall[rows][:, oxi])
    ref = np.abs(ref_psi.conj() @ ref_psi.T) ** 2
    np.fill_diagonal(ref, 1.0)
    idx = torch.from_numpy(rows).cuda()
    assert np.abs(K[idx][:, idx].cpu().numpy() - ref).max() < 1e-12
    del K, psi
    torch.cuda.empty_cache()


def test_config3_population_batch_full_size():
    """configs[2] as the GA runs it: 32 ten-qubit genomes (v5 decoder) x N = 16384 in ONE qk_evaluate_batch call,
    statistics only (no K copied back).  Two of the genomes are checked against statistics the oracle computes
    from its own statevectors (K formed block by block on the host, never held whole)."""
    from quake_amd import genetic
    from quake_amd.kernels import evaluate_population
    N, Q, F, G, count = 16384, 10, 18, 8, 32
    gd = {"single_non_parametric": ["Id", "H", "X", "SX"], "single_parametric": ["RX", "RY", "RZ"],
          "two_non_parametric": ["CX"], "two_parametric": []}
    rng = np.random.default_rng(1003)
    Xall = rng.uniform(1e-6, 1 - 1e-6, size=(N, F))
    y = rng.integers(0, 2, size=N)
    pop = genetic.initial_population(count, Q, G, gd, F, rng=rng)
    cmap = [[q, q + 1] for q in range(Q - 1)] + [[q + 1, q] for q in range(Q - 1)]
    kw = dict(coupling_map=cmap, variant="v5", physical_qubits=list(range(Q)))
    decoded = [genetic.to_quantum(g, gd, F, G, Q, verbose=False, **kw) for g in pop]
    none, stats = evaluate_population([m for m, _ in decoded], [np.ascontiguousarray(Xall[:, xi]) for _, xi in decoded],
                                      labels=y, return_kernels=False)
    assert none is None and len(stats) == count
    for g in (0, count - 1):
        oops, oxi = O.to_quantum(pop[g], gd, F, G, Q, **kw)
        psi = O.statevectors(oops, Q, Xall[:, oxi])
        s = ss = s00 = s11 = s01 = 0.0
        y0, y1 = (y == 0), (y == 1)
        B = 2048
        for r0 in range(0, N, B):
            Kb = np.abs(psi[r0:r0 + B].conj() @ psi[r0:].T) ** 2          # rows r0.., columns r0.. (upper part)
            Kb[np.tril_indices(B, 0, Kb.shape[1])] = 0.0                  # strict upper triangle only
            a0, a1 = y0[r0:r0 + B], y1[r0:r0 + B]
            c0, c1 = y0[r0:], y1[r0:]
            s += Kb.sum(); ss += (Kb ** 2).sum()
            s00 += Kb[a0][:, c0].sum(); s11 += Kb[a1][:, c1].sum()
            s01 += Kb[a0][:, c1].sum() + Kb[a1][:, c0].sum()
        n0, n1 = y0.sum(), y1.sum()
        mean = s / (N * N)
        want = {"sparsity": (s00 / (n0 * (n0 - 1) / 2) + s11 / (n1 * (n1 - 1) / 2)) / 2 - s01 / (n0 * n1),
                "offdiagonal_mean": mean, "offdiagonal_std": np.sqrt(ss / (N * N) - mean * mean)}
        for key, v in want.items():
            assert abs(stats[g][key] - v) < 1e-10, (g, key, stats[g][key], v)
