"""FP32 mode (optional precision of BASELINE.json north_star): complex64 statevectors + FP32-MFMA
Gram stage against the FP64 oracle.  Tolerance of the north star: 1e-5 absolute on K; the two-level
accumulation keeps the kernel two orders of magnitude inside it."""
import numpy as np
import pytest
import torch

from oracle import qk_oracle as O
from quake_amd import device as qdev, featuremaps as fm
from quake_amd.circuit import ParameterVector
from quake_amd.kernels import HipFidelityKernel
from tests._helpers import map_pairs, synth_X

pytestmark = pytest.mark.gpu
TOL_K32 = 1e-5          # north_star
TOL_PSI32 = 2e-7        # one rounding of an amplitude of modulus <= 1


@pytest.mark.parametrize("n,N", [(1, 7), (3, 40), (4, 33), (5, 70), (8, 130), (10, 65), (12, 40)])
def test_kernel_object_single_precision(n, N):
    X = synth_X(N, n, "scaled", 500 + n)
    for name, (circ, oops, nq) in map_pairs(n, 2).items():
        k32 = HipFidelityKernel(circ, precision="single")
        K = k32.evaluate(X)
        ref = O.evaluate(oops, nq, X)
        assert K.dtype == np.float64 and np.array_equal(K, K.T) and (np.diag(K) == 1.0).all()
        assert np.abs(K - ref).max() < TOL_K32, (name, n)
        Y = synth_X(N // 2 + 1, n, "scaled", 600 + n)
        assert np.abs(k32.evaluate(X, Y) - O.evaluate(oops, nq, X, Y)).max() < TOL_K32, (name, n)


@pytest.mark.parametrize("n,builder,repeats,N", [(9, "zz", 2, 50), (14, "zz", 2, 5), (16, "zz", 1, 4), (15, "custom1", 2, 3)])
def test_complex64_statevectors(n, builder, repeats, N):
    """Evolved in FP64, rounded once at the store (also through the multi-pass workspace path)."""
    X = synth_X(N, n, "scaled", 700 + n)
    circ, oops, _ = map_pairs(n, repeats)[builder]
    dc = qdev.DeviceCircuit(circ)
    psi32 = dc.simulate(torch.from_numpy(X).cuda(), dtype=torch.complex64)
    assert psi32.dtype == torch.complex64
    ref = O.statevectors(oops, n, X)
    assert np.abs(psi32.cpu().numpy() - ref).max() < TOL_PSI32
    assert np.array_equal(psi32.cpu().numpy(), ref.astype(np.complex64))     # exactly the rounded FP64 result


def test_fp32_gram_accumulation_at_full_width():
    """D = 2^16 amplitudes per row: the regime where plain FP32 accumulation drifts.  Rows with
    large overlaps (worst case for absolute error) and random rows."""
    n, N = 16, 192
    rng = np.random.default_rng(0)
    base = rng.normal(size=(1 << n)) + 1j * rng.normal(size=(1 << n))
    psi = np.empty((N, 1 << n), dtype=np.complex128)
    for i in range(N):
        v = base + (0.02 * i) * (rng.normal(size=(1 << n)) + 1j * rng.normal(size=(1 << n)))
        psi[i] = v / np.linalg.norm(v)
    ref = np.abs(psi.conj() @ psi.T) ** 2
    p32 = torch.from_numpy(psi.astype(np.complex64)).cuda()
    K = qdev.gram(p32).cpu().numpy()
    err = np.abs(K - ref)
    np.fill_diagonal(err, 0.0)           # the diagonal is set to exactly 1
    assert err.max() < 2e-6, err.max()
    Kx = qdev.gram(p32[:100], p32[100:]).cpu().numpy()
    assert np.abs(Kx - ref[:100, 100:]).max() < 2e-6


def test_fp32_multi_job_and_transposed_copy():
    n, N = 7, 300
    X = synth_X(N, n, "scaled", 9)
    dc = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 1))
    psi = dc.simulate(torch.from_numpy(X).cuda(), dtype=torch.complex64)
    ref = O.evaluate(O.zz_featuremap(n, 1), n, X)
    K = torch.full((N, N + 5), -1.0, dtype=torch.float64, device="cuda")
    KT = torch.full((130, 170), -1.0, dtype=torch.float64, device="cuda")
    qdev.gram_jobs([dict(A=psi[:170], B=psi[:170], K=K[:170, :170], symmetric=True),
                    dict(A=psi[:170], B=psi[170:], K=K[:170, 170:N], KT=KT),
                    dict(A=psi[170:], B=psi[170:], K=K[170:, 170:N], symmetric=True)], 1 << n)
    Kh = K.cpu().numpy()
    assert np.abs(Kh[:170, :N] - ref[:170]).max() < TOL_K32
    assert np.abs(Kh[170:, 170:N] - ref[170:, 170:]).max() < TOL_K32
    assert np.array_equal(KT.cpu().numpy(), Kh[:170, 170:N].T)
    assert (Kh[:, N:] == -1.0).all()
