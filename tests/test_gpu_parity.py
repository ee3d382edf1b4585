"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Tolerances (BASELINE.json north_star): |K_hip - K_oracle| <= 1e-10 absolute in FP64.
The tests hold the kernels to 1e-12 on K and on every statevector amplitude.
"""
import numpy as np
import pytest
from sklearn.model_selection import cross_val_score
from sklearn.svm import SVC

from oracle import qk_oracle as O
from quake_amd import circuit as qc
from quake_amd import featuremaps as fm
from quake_amd import genetic
from quake_amd.kernels import HipFidelityKernel, make_kernels, projected_quantum_kernel
from tests._helpers import map_pairs, psi_from_compiled, synth_X, two_qubit_pairs
from tests.test_circuit_ir import GATE_DICT_FULL

pytestmark = pytest.mark.gpu
TOL_K = 1e-12
TOL_PSI = 1e-12


@pytest.mark.parametrize("n,repeats", [(1, 1), (2, 2), (3, 1), (5, 2), (8, 2), (10, 1)])
def test_statevectors_all_maps(n, repeats):
    X = synth_X(37, n, "scaled", 100 + n)
    for name, (circ, oops, nq) in map_pairs(n, repeats).items():
        got = HipFidelityKernel(circ).statevectors(X)
        ref = O.statevectors(oops, nq, X)
        assert got.shape == ref.shape and got.dtype == np.complex128
        assert np.abs(got - ref).max() < TOL_PSI, (name, n, repeats)


@pytest.mark.parametrize("repeats", [1, 3])
def test_statevectors_two_qubit_maps(repeats):
    X = synth_X(20, 2, "scaled", 7)
    for name, (circ, oops, nq) in two_qubit_pairs(repeats).items():
        Xn = X[:, :circ.num_parameters]
        got = HipFidelityKernel(circ).statevectors(Xn)
        assert np.abs(got - O.statevectors(oops, nq, Xn)).max() < TOL_PSI, name


@pytest.mark.parametrize("n,builder,repeats,N", [
    (12, "zz", 1, 5), (13, "custom1", 1, 3), (14, "zz", 2, 3), (15, "custom1", 1, 2), (16, "zz", 1, 2), (16, "custom1", 2, 2),
    (13, "zz_full", 1, 3), (15, "zz_full", 2, 2),
])
def test_statevectors_large_and_multipass(n, builder, repeats, N):
    """n > 13 does not fit one workgroup's LDS: the simulator runs several HBM passes."""
    X = synth_X(N, n, "scaled", 200 + n)
    circ, oops, _ = map_pairs(n, repeats)[builder]
    k = HipFidelityKernel(circ)
    got = k.statevectors(X)
    if n > 13 and repeats > 1:
        assert k._circuit().num_passes >= 2
    assert np.abs(got - O.statevectors(oops, n, X)).max() < TOL_PSI


@pytest.mark.parametrize("n,builder,repeats,N", [
    (11, "zz", 2, 5), (11, "z", 3, 3), (12, "zz", 2, 5), (12, "z", 3, 3), (12, "zz_full", 2, 3), (13, "zz", 2, 3), (13, "z", 3, 2),
    (12, "custom1", 2, 3), (13, "custom2", 2, 2),
])
def test_statevectors_hadamard_layers(n, builder, repeats, N):
    """Layers of Hadamards between diagonal stages in ONE pass (11, 12 and 13 local qubits: 128, 256 and 512 threads):
    the simulator takes them four to a sweep (radix 16), the sweep over qubits 0-3 through lane-rotated positions with
    sign repair; the custom maps mix other gates in and keep the general sweeps."""
    X = synth_X(N, n, "scaled", 300 + n)
    circ, oops, _ = map_pairs(n, repeats)[builder]
    k = HipFidelityKernel(circ)
    got = k.statevectors(X)
    assert k._circuit().num_passes == 1
    assert np.abs(got - O.statevectors(oops, n, X)).max() < TOL_PSI


@pytest.mark.parametrize("n,builder,repeats", [(6, "zz", 1), (8, "z", 1), (8, "zz", 2), (9, "zz", 3), (10, "zz", 2), (10, "custom1", 1)])
def test_statevectors_small_width_large_batch(n, builder, repeats):
    """Up to 10 qubits a batch of 2^22 amplitudes or more sends the maps (product state + diagonal stages + layers of
    Hadamards) to the LDS kernel instead of the wave-resident one: rows sampled from such a batch against the oracle."""
    N = (1 << 22) >> n
    X = synth_X(N, n, "scaled", 400 + n)
    circ, oops, _ = map_pairs(n, repeats)[builder]
    got = HipFidelityKernel(circ).statevectors(X)
    rows = np.unique(np.concatenate([np.arange(3), np.random.default_rng(n).integers(0, N, 40), [N - 2, N - 1]]))
    assert np.abs(got[rows] - O.statevectors(oops, n, X[rows])).max() < TOL_PSI
    assert np.abs(np.sum(np.abs(got) ** 2, axis=1) - 1.0).max() < 1e-12


@pytest.mark.parametrize("n,builder,repeats,N", [
    (11, "zz", 2, 8192), (12, "zz", 2, 4096), (12, "z", 3, 4096), (13, "zz", 2, 4096), (12, "custom1", 2, 4096),
    (14, "zz", 2, 1024), (15, "zz_full", 2, 384), (16, "zz", 2, 192), (16, "custom1", 2, 192), (16, "zz", 1, 192),
])
def test_statevectors_many_items_per_workgroup(n, builder, repeats, N):
    """the persistent workgroups of the LDS simulator at 11-16 qubits with SEVERAL items each -- phase
    tables kept from one item to the next (DevStage::reuse), the angle prefetch, the per-workgroup deposit tables,
    the h_load / h_store sweeps of two-pass circuits -- whereas the other tests of this file give a workgroup one
    item at most.  The batch is simulated on the device (up to 512 MiB of Psi); first, last, middle and random rows
    go to the host and against the oracle."""
    import torch
    from quake_amd import device as qdev
    X = synth_X(N, n, "scaled", 600 + n + repeats)
    circ, oops, _ = map_pairs(n, repeats)[builder]
    dc = qdev.DeviceCircuit(circ)
    items = N << max(0, n - 12)
    assert items >= 2 * 1024, "every workgroup (at most 4 per CU x 256 CUs) must get two items or more"
    psi = dc.simulate(torch.from_numpy(X).cuda())
    rows = np.unique(np.concatenate([[0, 1, N // 2 - 1, N // 2, N - 2, N - 1], np.random.default_rng(n).integers(0, N, 10)]))
    got = psi[torch.from_numpy(rows).cuda()].cpu().numpy()
    assert np.abs(got - O.statevectors(oops, n, X[rows])).max() < TOL_PSI
    norms = (psi.real ** 2 + psi.imag ** 2).sum(dim=1)
    assert float((norms - 1.0).abs().max()) < 1e-12          # every row of the batch, not only the sampled ones
    # the same rows simulated alone (one item per workgroup) give the same bits: nothing leaks between items
    alone = dc.simulate(torch.from_numpy(np.ascontiguousarray(X[rows])).cuda()).cpu().numpy()
    assert np.array_equal(alone, got)


def test_statevectors_many_items_per_workgroup_genomes():
    """... and the general sweeps (composite 2x2 matrices, CX, 512-thread passes) on 12-qubit GA genomes, 4096 rows"""
    import torch
    from quake_amd import device as qdev
    gate_dict = {"single_non_parametric": ["Id", "H", "X", "SX"], "single_parametric": ["RX", "RY", "RZ"],
                 "two_non_parametric": ["CX"], "two_parametric": []}
    n, F, G, N = 12, 18, 8, 4096
    pop = genetic.initial_population(2, n, G, gate_dict, F, rng=np.random.default_rng(77))
    for genes in pop:
        fmap, xi = genetic.to_quantum(genes, gate_dict, F, G, n, physical_qubits=list(range(n)), verbose=False)
        oops, oxi = O.to_quantum(genes, gate_dict, F, G, n, physical_qubits=list(range(n)))
        assert xi == oxi
        X = synth_X(N, F, "unit", 612)[:, xi]
        psi = qdev.DeviceCircuit(fmap).simulate(torch.from_numpy(np.ascontiguousarray(X)).cuda())
        rows = np.unique(np.concatenate([[0, N // 2, N - 1], np.random.default_rng(5).integers(0, N, 9)]))
        got = psi[torch.from_numpy(rows).cuda()].cpu().numpy()
        assert np.abs(got - O.statevectors(oops, n, X[rows])).max() < TOL_PSI


@pytest.mark.parametrize("n", [10, 11, 12, 13])
def test_statevectors_hadamard_layers_without_cross_terms(n):
    """Layers of Hadamards between diagonal stages none of whose two-qubit terms straddles local bits 5 | 6 (no cross
    table): the fused diagonal stage + first radix-16 sweep path in its plain form, three layers deep."""
    x = qc.ParameterVector("x", n)
    c = qc.QuantumCircuit(n)
    for q in range(n):
        c.h(q)
    for rep in range(3):
        for q in range(5):
            c.cz(q, q + 1)                                  # two-qubit terms inside the low half (every qubit in one,
        for q in range(6, n - 1):                           # or the planner multiplies its h p h p ... into one matrix)
            c.cx(q, q + 1); c.p(x[q] * x[q + 1], q + 1); c.cx(q, q + 1)   # ... and inside the high half
        for q in range(n):
            c.p(x[q] * (1.0 + 0.25 * rep), q)
        for q in range(n):
            c.h(q)
    c.cz(n - 1, n - 2)
    X = synth_X(4, n, "scaled", 500 + n)
    k = HipFidelityKernel(c)
    got = k.statevectors(X)
    assert k._circuit().num_passes == 1
    assert np.abs(got - psi_from_compiled(c, X)).max() < TOL_PSI


def _fresh_qubit_circuit(n):
    """First gates on high (non-local) qubits in every flavour the planner distinguishes."""
    x = qc.ParameterVector("x", 4)
    c = qc.QuantumCircuit(n)
    c.h(0); c.ry(x[0], 1)
    c.rz(x[1], n - 1)            # diagonal gate BEFORE the first gate of the top qubit: no scalar shortcut
    c.h(n - 1)
    c.cx(n - 2, 0)               # control on a qubit that is still |0>
    c.ry(x[2], n - 2)            # parametric first gate on a high qubit
    c.sx(n - 3)                  # constant first gate on a high qubit
    c.cz(n - 2, 2); c.cx(n - 2, 1); c.cx(n - 3, n - 2)
    c.p(x[3], n - 3); c.rz(x[0] * x[1], n - 2)
    for q in range(2, n - 4):
        c.h(q)
    for q in range(n - 4):
        c.cx(q, q + 1); c.p(x[q % 4], q + 1); c.cx(q, q + 1)
    c.rx(x[1], n - 2); c.h(n - 3); c.cz(0, n - 1)
    return c                      # qubit n-4 is never touched


@pytest.mark.parametrize("n", [9, 13, 14, 15, 17])
def test_statevectors_fresh_high_qubits(n):
    circ = _fresh_qubit_circuit(n)
    X = synth_X(3, 4, "scaled", 300 + n)
    got = HipFidelityKernel(circ).statevectors(X)
    assert np.abs(got - psi_from_compiled(circ, X)).max() < TOL_PSI


@pytest.mark.parametrize("n,seed", [(11, 0), (12, 4), (13, 5), (14, 1), (15, 2), (16, 3)])
def test_statevectors_random_circuits(n, seed):
    """Random gate soups: every stage kind, many cross-half diagonal terms, several passes."""
    rng = np.random.default_rng(seed)
    x = qc.ParameterVector("x", 5)
    c = qc.QuantumCircuit(n)
    for _ in range(12 * n):
        g = rng.integers(0, 11)
        q, r = (int(v) for v in rng.choice(n, size=2, replace=False))
        a = x[int(rng.integers(0, 5))] * float(rng.uniform(0.5, 2.0))
        if g == 0: c.h(q)
        elif g == 1: c.sx(q)
        elif g == 2: c.x(q)
        elif g == 3: c.rx(a, q)
        elif g == 4: c.ry(a, q)
        elif g == 5: c.rz(a, q)
        elif g == 6: c.p(a, q)
        elif g == 7: c.cx(q, r)
        elif g == 8: c.cz(q, r)
        elif g == 9: c.ecr(q, r)
        else:
            c.cx(q, r); c.p(a, r); c.cx(q, r)
    X = synth_X(2, 5, "scaled", 400 + n)
    got = HipFidelityKernel(c).statevectors(X)
    assert np.abs(got - psi_from_compiled(c, X)).max() < TOL_PSI


def _gate_soup(n, seed, gates, F=5):
    rng = np.random.default_rng(seed)
    x = qc.ParameterVector("x", F)
    c = qc.QuantumCircuit(n)
    for k in range(F):
        c.ry(x[k] * 0.7, int(rng.integers(0, n)))
    for _ in range(gates):
        g = rng.integers(0, 11)
        q, r = (int(v) for v in rng.choice(n, size=2, replace=False))
        a = x[int(rng.integers(0, F))] * float(rng.uniform(0.5, 2.0))
        if g == 0: c.h(q)
        elif g == 1: c.sx(q)
        elif g == 2: c.x(q)
        elif g == 3: c.rx(a, q)
        elif g == 4: c.ry(a, q)
        elif g == 5: c.rz(a, q)
        elif g == 6: c.p(a, q)
        elif g == 7: c.cx(q, r)
        elif g == 8: c.cz(q, r)
        elif g == 9: c.ecr(q, r)
        else:
            c.cx(q, r); c.p(a, r); c.cx(q, r)
    return c


@pytest.mark.parametrize("n,seed,N", [(11, 20, 1500), (12, 21, 1100), (13, 22, 600)])
def test_register_resident_simulator_many_samples(n, seed, N):
    """The register-resident kernel (qk_reg.hip: 11-13 qubits, one pass) on gate soups -- register-register and
    thread-controlled cx, ecr, composite matrices, diagonal stages with terms between register and thread bits in
    every layout -- with several samples per persistent workgroup (angle double buffer, counted memory waits) and in
    the FP32-output mode; rows sampled against the oracle's simulator, every row's norm checked."""
    import torch
    from quake_amd import device as qdev
    c = _gate_soup(n, seed, 9 * n)
    h = HipFidelityKernel(c)._circuit()
    assert h.num_passes == 1 and "register-resident plan" in h.plan_summary
    X = synth_X(N, 5, "scaled", 700 + n)
    dc = qdev.DeviceCircuit(c)
    Xd = torch.from_numpy(X).cuda()
    psi = dc.simulate(Xd)
    rows = np.unique(np.concatenate([[0, 1, N // 2, N - 2, N - 1], np.random.default_rng(n).integers(0, N, 7)]))
    ref = psi_from_compiled(c, X[rows])
    assert np.abs(psi[torch.from_numpy(rows).cuda()].cpu().numpy() - ref).max() < TOL_PSI
    assert float(((psi.real ** 2 + psi.imag ** 2).sum(dim=1) - 1.0).abs().max()) < 1e-12
    psi32 = dc.simulate(Xd, dtype=torch.complex64)
    assert np.abs(psi32[torch.from_numpy(rows).cuda()].cpu().numpy() - ref).max() < 2e-7
    del psi, psi32


@pytest.mark.parametrize("n,seed,N,gates", [(4, 30, 131072 + 37, 8), (5, 31, 65536 + 5, 12), (6, 32, 40000 + 1, 16), (7, 33, 20000 + 3, 63),
                                            (8, 34, 9000 + 7, 72), (9, 35, 4096 + 1, 81)])
def test_register_resident_simulator_packed_samples(n, seed, N, gates):
    """4 to 9 qubits: several samples share a workgroup of the register-resident kernel (index bits above the circuit's
    qubits number the sample).  Gate soups as above; N is not a multiple of the group size, so the last group is
    short; every row is compared with the oracle's simulator (narrow states: cheap), in FP64 and in FP32-output mode."""
    import torch
    from quake_amd import device as qdev
    c = _gate_soup(n, seed, gates)
    h = HipFidelityKernel(c)._circuit()
    assert h.num_passes == 1 and "register-resident plan" in h.plan_summary and " 1 samples per workgroup" not in h.plan_summary
    X = synth_X(N, 5, "scaled", 900 + n)
    dc = qdev.DeviceCircuit(c)
    Xd = torch.from_numpy(X).cuda()
    guard = torch.full((N + 64, 1 << n), 7.0, dtype=torch.complex128, device="cuda")      # rows past N must stay untouched
    psi = dc.simulate(Xd, out=guard[:N])
    ref = psi_from_compiled(c, X)
    assert np.abs(psi.cpu().numpy() - ref).max() < TOL_PSI
    assert bool((guard[N:] == 7.0).all())
    psi32 = dc.simulate(Xd, dtype=torch.complex64)
    assert np.abs(psi32.cpu().numpy() - ref).max() < 2e-7
    # a batch smaller than one group, and one sample
    for small in (1, 3):
        assert np.abs(dc.simulate(Xd[:small].contiguous()).cpu().numpy() - ref[:small]).max() < TOL_PSI


@pytest.mark.parametrize("n,N", [(8, 16384 + 9), (9, 8192 + 5), (10, 8192 + 3), (11, 900), (12, 700), (13, 300)])
def test_register_resident_ratio_tables(n, N):
    """Diagonal stages with many terms between register and thread qubits (ZZ map with full entanglement, r = 2, plus a
    CZ between every qubit and the last one): the planner turns those terms into ratio tables per register position
    (qk_circuit.cpp, section 4; 13 qubits: the high-bits tables come in two slots), the kernel multiplies them into its
    Gray-code walk.  Batches large enough for the default policy to take the register path; rows sampled against the
    oracle, every row's norm."""
    import torch
    from quake_amd import device as qdev, featuremaps as fm
    x = qc.ParameterVector("x", n)
    c = fm.zz_featuremap(x, n, 2, entanglement="full")
    for q in range(n - 1):
        c.cz(q, n - 1)
    dc = qdev.DeviceCircuit(c)
    assert "register-resident plan" in dc.handle.plan_summary, dc.handle.plan_summary
    # (10 qubits and below: where the tables -- of all the samples of a workgroup, when samples are packed -- would cost
    # too many resident workgroups the planner keeps those terms per thread: the same circuit through the other code path)
    if n >= 11:
        assert "ratio tables" in dc.handle.plan_summary, dc.handle.plan_summary
    X = synth_X(N, n, "scaled", 1300 + n)
    psi = dc.simulate(torch.from_numpy(X).cuda())
    rows = np.unique(np.concatenate([[0, 1, N - 2, N - 1], np.random.default_rng(n).integers(0, N, 6)]))
    assert np.abs(psi[torch.from_numpy(rows).cuda()].cpu().numpy() - psi_from_compiled(c, X[rows])).max() < TOL_PSI
    assert float(((psi.real ** 2 + psi.imag ** 2).sum(dim=1) - 1.0).abs().max()) < 1e-12
    del psi


@pytest.mark.parametrize("n,N,seed", [(6, 70000, 100), (7, 40000, 101), (8, 16384 + 3, 101), (9, 8192 + 5, 100), (10, 8192 + 1, 102),
                                      (12, 1024 + 3, 103), (13, 512 + 1, 100)])
def test_hardware_native_genomes_large_batch(n, N, seed):
    """GA genomes of the hardware-native alphabet of the reference's examples ({id, x, sx, rz, ecr}: every rotation is
    rz sx rz) at batch sizes that take the register-resident kernel: the planner multiplies each run of rz around one
    sx / x into one matrix per sample (qk_circuit.cpp section 1b, end of round 3) and the pre-pass kernels write those
    matrices through LDS in whole lines.  Rows sampled against the oracle, every row's norm."""
    import torch
    from quake_amd import device as qdev
    hw = {"single_non_parametric": ["Id", "X", "SX"], "single_parametric": ["RZ", "RZ"], "two_non_parametric": ["ECR"], "two_parametric": []}
    genes = genetic.initial_population(1, n, 8, hw, n, rng=np.random.default_rng(seed))[0]
    c, cols = genetic.to_quantum(genes, hw, n, 8, n, physical_qubits=list(range(n)), verbose=False)
    dc = qdev.DeviceCircuit(c)
    assert "register-resident plan" in dc.handle.plan_summary, dc.handle.plan_summary
    X = np.random.default_rng(seed + n).uniform(0.02, 0.98, size=(N, n))[:, cols]      # (get_subsample's range: asin stays real)
    psi = dc.simulate(torch.from_numpy(np.ascontiguousarray(X)).cuda())
    rows = np.unique(np.concatenate([[0, 1, N - 2, N - 1], np.random.default_rng(n).integers(0, N, 8)]))
    assert np.abs(psi[torch.from_numpy(rows).cuda()].cpu().numpy() - psi_from_compiled(c, X[rows])).max() < TOL_PSI
    assert float(((psi.real ** 2 + psi.imag ** 2).sum(dim=1) - 1.0).abs().max()) < 1e-12
    del psi


@pytest.mark.parametrize("n,N,seed", [(3, 7, 0), (5, 33, 1), (8, 16384 + 1, 2), (10, 8192 + 3, 3), (12, 1024 + 1, 4), (14, 5, 5)])
def test_cx_and_cz_on_basis_state_controls(n, N, seed):
    """Circuits that open with id / x / rz / p / cz / cx / ecr on every qubit -- controls that are still |0> or |1> when
    their cx or cz comes, which the planner drops or turns into an x, and ecr gates on such a first qubit, which it writes
    as rz(pi/2), sx and x (qk_circuit.cpp section 1) -- followed by gates of every kind: the oracle applies every gate as
    written."""
    rng = np.random.default_rng(500 + seed)
    F = 3
    x = qc.ParameterVector("x", F)
    c = qc.QuantumCircuit(n)
    for k in range(F):
        c.rz(x[k], k % n)
    for _ in range(4 * n):      # the opening: basis states and phases, and ecr gates whose first qubit may still be one
        g = int(rng.integers(0, 7))
        q = int(rng.integers(0, n))
        r = int((q + 1 + rng.integers(0, n - 1)) % n)
        a = x[int(rng.integers(0, F))] * float(rng.uniform(0.5, 2.0))
        if g == 0: c.id(q)
        elif g == 1: c.x(q)
        elif g == 2: c.rz(a, q)
        elif g == 3: c.p(a, q)
        elif g == 4: c.cz(q, r)
        elif g == 5: c.cx(q, r)
        else: c.ecr(q, r)
    for _ in range(3 * n):      # then everything
        g = int(rng.integers(0, 9))
        q = int(rng.integers(0, n))
        r = int((q + 1 + rng.integers(0, n - 1)) % n)
        a = x[int(rng.integers(0, F))] * float(rng.uniform(0.5, 2.0))
        if g == 0: c.h(q)
        elif g == 1: c.sx(q)
        elif g == 2: c.ry(a, q)
        elif g == 3: c.rx(a, q)
        elif g == 4: c.cx(q, r)
        elif g == 5: c.cz(q, r)
        elif g == 6: c.ecr(q, r)
        elif g == 7: c.x(q)
        else: c.p(a, q)
    X = synth_X(N, F, "scaled", 700 + seed)
    got = HipFidelityKernel(c).statevectors(X) if N <= 64 else None
    if got is None:
        import torch
        from quake_amd import device as qdev
        psi = qdev.DeviceCircuit(c).simulate(torch.from_numpy(X).cuda())
        rows = np.unique(np.concatenate([[0, 1, N - 1], rng.integers(0, N, 6)]))
        assert np.abs(psi[torch.from_numpy(rows).cuda()].cpu().numpy() - psi_from_compiled(c, X[rows])).max() < TOL_PSI
        del psi
    else:
        assert np.abs(got - psi_from_compiled(c, X)).max() < TOL_PSI


def test_register_resident_simulator_in_chunks():
    """A batch whose pre-pass tables would exceed 1 GiB goes through the register-resident path in chunks of whole sample
    groups (6 qubits, ZZ r = 2: 1376 bytes of angles and tables per sample, 800 003 samples = a chunk of 2^30 / 1376 samples rounded down to whole groups and the rest, the second one
    ending in a short group); rows sampled from both chunks and around the seam against the oracle, every row's norm."""
    import torch
    from quake_amd import device as qdev
    n, N = 6, 800003
    circ, oops, _ = map_pairs(n, 2)["zz"]
    assert "register-resident plan" in HipFidelityKernel(circ)._circuit().plan_summary
    X = synth_X(N, n, "scaled", 77)
    psi = qdev.DeviceCircuit(circ).simulate(torch.from_numpy(X).cuda())
    # (the seams: one chunk of 2^30 / 1376 samples and the rest -- or, when the pre-pass runs ahead on its own stream, see
    # the next test, four chunks of a quarter of the batch rounded down to whole groups)
    seams = [np.arange(780200, 780400)] + [np.arange(k * N // 4 - 300, k * N // 4 + 50) for k in (1, 2, 3)]
    rows = np.unique(np.concatenate([[0, 1, N - 2, N - 1], *seams, np.random.default_rng(5).integers(0, N, 200)]))
    got = psi[torch.from_numpy(rows).cuda()].cpu().numpy()
    assert np.abs(got - O.statevectors(oops, n, X[rows])).max() < TOL_PSI
    assert float(((psi.real ** 2 + psi.imag ** 2).sum(dim=1) - 1.0).abs().max()) < 1e-12
    del psi


def test_register_resident_prepass_runs_ahead():
    """Where a sample's angle and stage tables are three eighths of its statevector or more (5 qubits, custom1 map r = 2:
    3.75 times), a call of 2^25 amplitudes or more goes in four chunks and the pre-pass kernels of chunk k + 1 run on the
    library's side stream while the simulator works on chunk k, two sets of tables alternating (qk_sim.hip, the register
    branch of simulate_impl).  Two calls back to back on a stream that is not the default one, no synchronisation between
    them -- the second call's pre-pass must not write a set that the first call's simulator is still reading -- then a
    call on the default stream; rows around every seam against the oracle, every row's norm."""
    import torch
    from quake_amd import device as qdev, featuremaps as fm
    n, N = 5, 1200003
    fmap = fm.custom1_featuremap(qc.ParameterVector("x", n), n, 2)
    dc = qdev.DeviceCircuit(fmap)
    assert "register-resident plan" in dc.handle.plan_summary
    Xs = [synth_X(N, n, "scaled", 900 + k) for k in range(2)]
    Xd = [torch.from_numpy(x).cuda() for x in Xs]
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out = [dc.simulate(xd) for xd in Xd]
    s.synchronize()
    out.append(dc.simulate(Xd[0]))
    torch.cuda.synchronize()
    seams = [np.arange(k * N // 4 - 300, k * N // 4 + 50) for k in (1, 2, 3)]
    rows = np.unique(np.concatenate([[0, 1, N - 2, N - 1], *seams, np.random.default_rng(6).integers(0, N, 100)]))
    ridx = torch.from_numpy(rows).cuda()
    for k, psi in enumerate(out):
        assert np.abs(psi[ridx].cpu().numpy() - psi_from_compiled(fmap, Xs[k % 2][rows])).max() < TOL_PSI, k
        assert float(((psi.real ** 2 + psi.imag ** 2).sum(dim=1) - 1.0).abs().max()) < 1e-12, k
    assert torch.equal(out[0], out[2])
    del out


def test_register_resident_prepass_runs_ahead_two_threads():
    """Two host threads, each with its own stream, call the simulator at the same time on batches that take the side
    stream (the calls share the library's one side stream per GPU; events are per thread): both results equal the
    single-threaded ones bit for bit."""
    import threading
    import torch
    from quake_amd import device as qdev, featuremaps as fm
    n, N = 5, 1100000
    dcs = [qdev.DeviceCircuit(fm.custom1_featuremap(qc.ParameterVector("x", n), n, 2)),
           qdev.DeviceCircuit(fm.zz_featuremap(qc.ParameterVector("x", n), n, 2))]
    Xd = [torch.from_numpy(synth_X(N, n, "scaled", 950 + k)).cuda() for k in range(2)]
    want = [dc.simulate(xd) for dc, xd in zip(dcs, Xd)]
    torch.cuda.synchronize()
    got, errors = [None, None], []

    def work(k):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(3):
                    got[k] = dcs[k].simulate(Xd[k])
            s.synchronize()
        except Exception as e:      # noqa: BLE001 (reported below, in the main thread)
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(got[k], want[k]), k
    del got, want


@pytest.mark.parametrize("n,seed", [(1, 0), (3, 1), (7, 2), (10, 3), (11, 4), (12, 5), (14, 6)])
def test_statevectors_single_qubit_runs(n, seed):
    """Long runs of single-qubit gates on one qubit between rare two-qubit gates (the shape of a GA genome
    circuit, reference genetic_v3.py:326-448): the planner multiplies each run into one 2x2 matrix per sample.
    Runs with rz / p inside, at their head and at their tail, first on a qubit (product state) and after a cx,
    on the wave-resident simulator (n <= 10) and the LDS one (one pass and several)."""
    rng = np.random.default_rng(100 + seed)
    F = 4
    x = qc.ParameterVector("x", F)
    c = qc.QuantumCircuit(n)
    for k in range(F):          # every feature is used at least once
        c.ry(x[k], k % n)
    for layer in range(10):
        for q in range(n):
            g = int(rng.integers(0, 8))
            a = x[int(rng.integers(0, F))] * float(rng.uniform(0.5, 2.0))
            if g == 0: c.h(q)
            elif g == 1: c.sx(q)
            elif g == 2: c.x(q)
            elif g == 3: c.rx(a, q)
            elif g == 4: c.ry(a, q)
            elif g == 5: c.rz(a, q)
            elif g == 6: c.p(a, q)
            else: c.id(q)
        if n > 1 and layer % 3 == 2:
            q, r = (int(v) for v in rng.choice(n, size=2, replace=False))
            if layer % 2: c.cx(q, r)
            else: c.ecr(q, r)
    X = synth_X(3, F, "scaled", 900 + n)
    got = HipFidelityKernel(c).statevectors(X)
    ref = psi_from_compiled(c, X)
    assert np.abs(got - ref).max() < TOL_PSI
    assert np.abs(np.linalg.norm(got, axis=1) - 1.0).max() < 1e-13


@pytest.mark.parametrize("N,M", [(1, 1), (5, 3), (130, 57), (257, 129)])
def test_evaluate_symmetric_and_cross(N, M):
    n = 6
    x = qc.ParameterVector("x", n)
    circ = fm.zz_featuremap(x, n, 2)
    oops = O.zz_featuremap(n, 2)
    X, Y = synth_X(N, n, "scaled", 1), synth_X(M, n, "scaled", 2)
    k = HipFidelityKernel(circ)
    K = k.evaluate(x_vec=X)
    Kref = O.evaluate(oops, n, X)
    assert K.shape == (N, N) and K.dtype == np.float64 and K.flags["C_CONTIGUOUS"]
    assert np.abs(K - Kref).max() < TOL_K
    assert np.array_equal(np.diag(K), np.ones(N))          # exact unit diagonal
    assert np.array_equal(K, K.T)                          # mirrored, bit-symmetric
    Kc = k.evaluate(x_vec=X, y_vec=Y)
    assert Kc.shape == (N, M)
    assert np.abs(Kc - O.evaluate(oops, n, X, Y)).max() < TOL_K


def test_golden_pins_on_gpu(pins, data8, data_labels):
    p = pins["P1"]
    fmap, xi = genetic.to_quantum(p["genes"], p["gate_dict"], 8, 8, 4, verbose=False)
    K = HipFidelityKernel(feature_map=fmap).evaluate(x_vec=data8[:, xi])   # float32 input, widened exactly
    assert abs(K[25, 41] - p["expected"]) < 1e-14
    p = pins["P2"]
    fmap, xi = genetic.to_quantum(p["genes"], p["gate_dict"], 8, 8, 4, verbose=False)
    K = HipFidelityKernel(feature_map=fmap).evaluate(x_vec=data8[:, xi])
    acc = cross_val_score(SVC(kernel="precomputed", C=1), K, data_labels, cv=5, scoring="accuracy").mean()
    assert acc == pytest.approx(p["expected_cv_accuracy"], abs=1e-12)
    oops, oxi = O.to_quantum(p["genes"], p["gate_dict"], 8, 8, 4)
    Kref = O.evaluate(oops, 4, data8[:, oxi])
    assert np.abs(K - Kref).max() < TOL_K
    # identical SVC predictions from the two Gram matrices
    a = SVC(kernel="precomputed", C=1).fit(K, data_labels).predict(K)
    b = SVC(kernel="precomputed", C=1).fit(Kref, data_labels).predict(Kref)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("cols", ["torino8", "all18"])
def test_config1_bundled_dataset(cols, data_cv, data8, data_labels):
    """BASELINE configs[0], the reference's own CPU-runnable case: the bundled 100-row dataset through
    zz_featuremap r = 1 -- on the 8 columns of examples/torino_run.py:64 (8 qubits) and on all 18
    columns (18 qubits: two simulator passes, statevectors of 4 MiB)."""
    X = (data8 if cols == "torino8" else data_cv[:48]).astype(np.float64)   # 18 qubits: the oracle takes ~0.4 s per row
    y = data_labels[:X.shape[0]]
    n = X.shape[1]
    circ, oops, _ = map_pairs(n, 1)["zz"]
    K = HipFidelityKernel(circ).evaluate(X)
    Kref = O.evaluate(oops, n, X)
    assert np.abs(K - Kref).max() < TOL_K
    assert np.array_equal(K, K.T) and (np.diag(K) == 1.0).all()
    a = SVC(kernel="precomputed", C=1).fit(K, y).predict(K)
    b = SVC(kernel="precomputed", C=1).fit(Kref, y).predict(Kref)
    assert np.array_equal(a, b)
    sa = cross_val_score(SVC(kernel="precomputed", C=1), K, y, cv=5)
    sb = cross_val_score(SVC(kernel="precomputed", C=1), Kref, y, cv=5)
    assert np.array_equal(sa, sb)


@pytest.mark.parametrize("variant,Q", [("v3", 4), ("v4", 4), ("v5", 10)])
def test_genome_circuits(variant, Q):
    """Config 3 in miniature: heterogeneous genomes, every gate of the alphabet."""
    rng = np.random.default_rng(31)
    G, F, N = 8, 18, 48
    cmap = [[q, q + 1] for q in range(Q - 1)] + [[q + 1, q] for q in range(Q - 1)] if variant == "v5" else None
    phys = list(range(Q))
    if variant == "v4":
        # genetic_v4.py:280-287: tuples of PHYSICAL qubits; a map that lacks one pair, so some two-qubit
        # genes on qubit 1 are dropped while those on qubits 0 and 3 are kept
        phys = [12, 13, 14, 15]
        cmap = [(12, 13), (13, 12), (15, 14), (14, 15)]
    checked = 0
    for trial in range(6):
        genes = genetic.initial_population(1, Q, G, GATE_DICT_FULL, F, rng=rng)[0]
        kw = dict(coupling_map=cmap, variant=variant, physical_qubits=phys)
        fmap, xi = genetic.to_quantum(genes, GATE_DICT_FULL, F, G, Q, verbose=False, **kw)
        oops, oxi = O.to_quantum(genes, GATE_DICT_FULL, F, G, Q, **kw)
        X = synth_X(N, F, "unit", trial)[:, xi]
        k = HipFidelityKernel(fmap)
        assert np.abs(k.statevectors(X) - O.statevectors(oops, Q, X)).max() < TOL_PSI
        assert np.abs(k.evaluate(X) - O.evaluate(oops, Q, X)).max() < TOL_K
        checked += 1
    assert checked == 6


def test_projected_kernel(pins, data8):
    n, N = 5, 70
    x = qc.ParameterVector("x", n)
    X = synth_X(N, n, "scaled", 9)
    for circ, oops in ((fm.zz_featuremap(x, n, 1), O.zz_featuremap(n, 1)),
                       (fm.custom1_featuremap(x, n, 2), O.custom1_featuremap(n, 2))):
        K = projected_quantum_kernel(circ, X, 1.0)
        Kref = O.projected_kernel(oops, n, X, 1.0)
        assert np.abs(K - Kref).max() < TOL_K
        assert np.array_equal(np.diag(K), np.ones(N)) and np.array_equal(K, K.T)
    K2 = projected_quantum_kernel(fm.zz_featuremap(x, n, 1), X, 0.37)
    assert np.abs(K2 - O.projected_kernel(O.zz_featuremap(n, 1), n, X, 0.37)).max() < TOL_K
    # a genome circuit on the bundled data
    p = pins["P1"]
    fmap, xi = genetic.to_quantum(p["genes"], p["gate_dict"], 8, 8, 4, verbose=False)
    oops, _ = O.to_quantum(p["genes"], p["gate_dict"], 8, 8, 4)
    assert np.abs(projected_quantum_kernel(fmap, data8[:, xi], 1) - O.projected_kernel(oops, 4, data8[:, xi], 1)).max() < TOL_K
    # no parameters -> all ones (reference genetic_v3.py:789-791)
    c = qc.QuantumCircuit(2)
    c.h(0)
    assert np.array_equal(projected_quantum_kernel(c, np.zeros((4, 0)), 1.0), np.ones((4, 4)))


def test_api_shapes_and_errors():
    n = 3
    x = qc.ParameterVector("x", n)
    k = HipFidelityKernel(fm.z_featuremap(x, n))
    with pytest.raises(ValueError):
        k.evaluate(np.zeros((4, n + 1)))
    with pytest.raises(ValueError):
        k.evaluate(np.zeros((4, n)), np.zeros((4, n - 1)))
    one = k.evaluate(np.array([0.1, 0.2, 0.3]))            # 1-D input = one sample
    assert one.shape == (1, 1) and one[0, 0] == 1.0
    cross = k.evaluate(np.array([0.1, 0.2, 0.3]), np.array([[0.1, 0.2, 0.3], [0.3, 0.2, 0.1]]))
    assert cross.shape == (1, 2) and abs(cross[0, 0] - 1.0) < 1e-14
    # empty inputs: empty outputs of the right shape (also for the projected kernel and FP32 mode)
    assert k.evaluate(np.zeros((0, n))).shape == (0, 0)
    assert k.evaluate(np.zeros((4, n)), np.zeros((0, n))).shape == (4, 0)
    assert k.evaluate(np.zeros((0, n)), np.zeros((3, n))).shape == (0, 3)
    assert HipFidelityKernel(fm.z_featuremap(x, n), precision="single").evaluate(np.zeros((0, n))).shape == (0, 0)
    assert projected_quantum_kernel(fm.z_featuremap(x, n), np.zeros((0, n)), 1.0).shape == (0, 0)
    # float32 and integer inputs widen exactly
    Xf = synth_X(9, n, "scaled", 3).astype(np.float32)
    assert np.array_equal(k.evaluate(Xf), k.evaluate(Xf.astype(np.float64)))
    # a circuit without parameters: every state identical, K == 1
    c = qc.QuantumCircuit(2)
    c.h(0); c.cx(0, 1); c.sx(1)
    K = HipFidelityKernel(c).evaluate(np.zeros((5, 0)))
    assert np.abs(K - 1).max() < 1e-14
    # make_kernels keeps the reference's signature; align=True maps get training parameters
    ks = make_kernels([fm.zz_featuremap(x, n), fm.zz_featuremap(x, n, align=True)], backend=None)
    assert ks[0].training_parameters == [] and len(ks[1].training_parameters) == n
    X = synth_X(6, n, "scaled", 5)
    assert np.abs(ks[1].evaluate(X) - ks[0].evaluate(X)).max() < 1e-13     # theta = 0: rx(0) = identity
    ks[1].assign_training_parameters([0.3, -0.2, 0.9])
    th = np.tile([0.3, -0.2, 0.9], (6, 1))
    # oracle: rx(theta_q) prefix then the zz map
    oops = [("rx", (q,), (lambda P, q=q: P[:, n + q]), (n + q,)) for q in range(n)] + O.zz_featuremap(n, 1)
    ref = O.fidelity_kernel(O.simulate(oops, n, np.hstack([X, th])))
    assert np.abs(ks[1].evaluate(X) - ref).max() < TOL_K


def test_gram_raw_layout_asymmetric():
    """MFMA operand/result layout check on data that is neither normalised nor symmetric."""
    import torch
    from quake_amd import device as dev

    rng = np.random.default_rng(77)
    for (N, M, D) in [(200, 300, 64), (129, 17, 8), (16, 16, 4), (3, 2, 2), (140, 140, 256)]:
        A = rng.normal(size=(N, D)) + 1j * rng.normal(size=(N, D))
        B = rng.normal(size=(M, D)) + 1j * rng.normal(size=(M, D))
        ref = np.abs(A.conj() @ B.T) ** 2
        tA, tB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
        KT = torch.zeros((M, N), dtype=torch.float64, device="cuda")
        K = torch.zeros((N, M), dtype=torch.float64, device="cuda")
        dev.gram_jobs([dict(A=tA, B=tB, K=K, KT=KT)], D)
        torch.cuda.synchronize()
        scale = ref.max()
        assert np.abs(K.cpu().numpy() - ref).max() < 1e-13 * scale, (N, M, D)
        assert np.array_equal(KT.cpu().numpy(), K.cpu().numpy().T)
        # strided output view + symmetric job in the same launch
        big = torch.full((N + 3, N + 5), -1.0, dtype=torch.float64, device="cuda")
        dev.gram_jobs([dict(A=tA, B=tA, K=big[1:N + 1, 2:N + 2], symmetric=True),
                       dict(A=tA, B=tB, K=K)], D)
        torch.cuda.synchronize()
        S = big.cpu().numpy()
        refs = np.abs(A.conj() @ A.T) ** 2
        np.fill_diagonal(refs, 1.0)
        assert np.abs(S[1:N + 1, 2:N + 2] - refs).max() < 1e-13 * refs.max()
        assert (S[0] == -1).all() and (S[:, :2] == -1).all() and (S[N + 1:] == -1).all() and (S[:, N + 2:] == -1).all()


def test_gram_kernel_variants_agree_bitwise():
    """Every Gram kernel sums the amplitudes of an entry in the same order (16 x 16 x 4 MFMA steps in ascending
    k, three accumulators), so the double-buffered kernel that serves rows beyond 2^22 amplitudes and both
    LDS-ring kernels must give the SAME bits -- on ragged edges, cross jobs with the transposed copy, symmetric jobs, one-stage rows (D = 8) and rows
    whose stage count leaves every remainder of the ring's unrolling."""
    import ctypes as C
    import torch
    from quake_amd import _lib, device as dev

    lib = _lib.load()
    lib.qk_gram_set_variant.argtypes = [C.c_char_p]
    lib.qk_gram_set_variant.restype = C.c_int
    names = ["dma3m_64x64_w4_g64", "ring3_64x64_w4", "ring5_64x64_w4"]
    rng = np.random.default_rng(78)
    try:
        # the last case has more tiles (864 + 528) than the chip has slots and a ragged last round in both jobs
        for (N, M, D) in [(70, 130, 8), (65, 64, 16), (200, 131, 32), (129, 257, 64), (300, 77, 128), (140, 140, 2048),
                          (2000, 1700, 32)]:
            A = rng.normal(size=(N, D)) + 1j * rng.normal(size=(N, D))
            B = rng.normal(size=(M, D)) + 1j * rng.normal(size=(M, D))
            tA, tB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
            outs = []
            for nm in names:
                if D < 16 and not nm.startswith("ring"):
                    continue       # 16-amplitude k-tiles: these kernels start at D = 16
                assert lib.qk_gram_set_variant(nm.encode()) == 0
                K = torch.full((N, M), -1.0, dtype=torch.float64, device="cuda")
                KT = torch.full((M, N), -1.0, dtype=torch.float64, device="cuda")
                S = torch.full((N, N), -1.0, dtype=torch.float64, device="cuda")
                dev.gram_jobs([dict(A=tA, B=tB, K=K, KT=KT), dict(A=tA, B=tA, K=S, symmetric=True)], D)
                torch.cuda.synchronize()
                outs.append((nm, K.cpu().numpy(), KT.cpu().numpy(), S.cpu().numpy()))
            ref = np.abs(A.conj() @ B.T) ** 2
            assert np.abs(outs[0][1] - ref).max() < 1e-13 * ref.max(), (N, M, D)
            for nm, K, KT, S in outs:
                assert np.array_equal(K, outs[0][1]), (nm, N, M, D)
                assert np.array_equal(KT, K.T), (nm, N, M, D)
                assert np.array_equal(S, outs[0][3]) and np.array_equal(S, S.T), (nm, N, M, D)
    finally:
        lib.qk_gram_set_variant(b"auto")


def test_equal_rows_stay_exactly_one():
    """FidelityStatevectorKernel skips pairs whose statevectors compare equal, so those entries are
    exactly 1.0 -- duplicate samples in the symmetric matrix, a test row that is also a train row in
    a cross block, every entry for a circuit without parameters."""
    n = 5
    circ, oops, _ = map_pairs(n, 2)["zz"]
    X = synth_X(40, n, "scaled", 9)
    X[7] = X[3]; X[21] = X[3]; X[30] = X[12]
    Y = synth_X(11, n, "scaled", 10)
    Y[4] = X[3]; Y[9] = X[0]
    k = HipFidelityKernel(circ)
    K, Kref = k.evaluate(X), O.evaluate(oops, n, X)
    assert np.array_equal(K == 1.0, Kref == 1.0) and (K == 1.0).sum() == 40 + 6 + 2
    assert np.abs(K - Kref).max() < TOL_K
    Kc, Kcref = k.evaluate(X, Y), O.evaluate(oops, n, X, Y)
    assert np.array_equal(Kc == 1.0, Kcref == 1.0) and (Kc == 1.0).sum() == 3 + 1
    assert np.abs(Kc - Kcref).max() < TOL_K
    const = qc.QuantumCircuit(3)
    const.h(0); const.cx(0, 1); const.sx(2)
    Kone = HipFidelityKernel(const).evaluate(np.zeros((6, 0)))
    assert Kone.shape == (6, 6) and (Kone == 1.0).all()


def test_pinned_output_arrays():
    """Large outputs of the host entry points live in page-locked memory (qk_host_alloc) so that K comes
    back at the PCIe rate; to the caller they are ordinary fresh numpy arrays."""
    import ctypes as C
    import gc
    from quake_amd import _lib
    lib = _lib.load()
    p = C.c_void_p()
    assert lib.qk_host_alloc(1 << 16, C.byref(p)) == 0 and p.value
    C.memset(p.value, 0xAB, 1 << 16)
    assert lib.qk_host_free(p) == 0 and lib.qk_host_free(None) == 0
    assert lib.qk_host_alloc(16, None) != 0
    n, N = 4, 600                                   # 600 x 600 doubles > PINNED_MIN_BYTES
    circ, oops, _ = map_pairs(n, 1)["custom1"]
    X = synth_X(N, n, "scaled", 3)
    K = HipFidelityKernel(circ).evaluate(X)
    assert K.nbytes >= _lib.PINNED_MIN_BYTES and isinstance(K.base, _lib._PinnedBlock)
    assert K.flags.writeable and K.flags.c_contiguous and K.dtype == np.float64
    assert np.abs(K - O.evaluate(oops, n, X)).max() < TOL_K
    view = K[10:20]
    del K
    gc.collect()
    assert view[0, 10] == 1.0                       # a view keeps the block alive
    K2 = np.array(view)                              # ordinary copy semantics
    del view
    gc.collect()
    assert K2[0, 10] == 1.0
    small = HipFidelityKernel(circ).evaluate(X[:8])
    assert small.base is None                        # small outputs stay pageable


def test_projected_kernel_distance_scales():
    """The pairwise kernel's square root is hand-rolled (hardware rsq seed + refinement): rows that
    differ by 1 down to 1e-12 per component, against a long-double evaluation of the same formula
    (reference genetic_v3.py:793-805)."""
    import torch
    from quake_amd import device as dev
    rng = np.random.default_rng(5)
    N, n = 700, 6
    for scale in (1.0, 1e-3, 1e-6, 1e-9, 1e-12, 0.0):
        r = rng.uniform(-1, 1, size=(1, n, 4)) + scale * rng.uniform(-1, 1, size=(N, n, 4))
        r[..., 1] = 1.0 - r[..., 0]       # unit trace, as every density matrix has it: the kernel relies on d rho11 = -d rho00
        K = dev.pqk(torch.from_numpy(r).cuda(), None, 1.0).cpu().numpy()
        d = (r[:, None] - r[None]).astype(np.longdouble)
        fro = np.sqrt(d[..., 0] ** 2 + d[..., 1] ** 2 + 2 * (d[..., 2] ** 2 + d[..., 3] ** 2))
        ref = np.exp(-fro.sum(-1)).astype(np.float64)
        np.fill_diagonal(ref, 1.0)
        assert np.abs(K - ref).max() < 5e-15, scale
        assert np.array_equal(K, K.T)


@pytest.mark.parametrize("precision,tol,n,N", [("double", TOL_K, 5, 3300), ("single", 1e-5, 5, 3300), ("double", TOL_K, 11, 5000)])
def test_host_path_row_bands(precision, tol, n, N):
    """Above 2048 rows the host entry point computes K in row bands and copies each band back while
    the next one is computed: symmetric (mirror entries come from earlier bands) and cross blocks.  Short rows: equal
    bands (3300 rows = 3 bands); from 11 qubits on the bands halve (5000 rows = 2560 + 1280 + 1160)."""
    M = 700
    circ, oops, _ = map_pairs(n, 1)["zz"]
    X, Y = synth_X(N, n, "scaled", 77), synth_X(M, n, "scaled", 78)
    k = HipFidelityKernel(circ, precision=precision)
    K = k.evaluate(X)
    ref = O.evaluate(oops, n, X)
    assert np.array_equal(K, K.T) and (np.diag(K) == 1.0).all()
    assert np.abs(K - ref).max() < tol
    assert np.abs(k.evaluate(X, Y) - O.evaluate(oops, n, X, Y)).max() < tol


def test_one_expression_feeding_half_and_full_angle_gates():
    """A C-ABI program whose rz and p read ONE expression slot (compile() never emits that): the planner keeps it off
    the wave-resident format (one scale per expression) and the result still equals the oracle."""
    from quake_amd import _lib
    from tests._helpers import compiled_to_oracle_ops
    from tests.test_planner import _shared_angle_program
    compiled = _shared_angle_program()
    h = _lib.CircuitHandle(compiled)
    assert "wave_program no" in h.plan_summary
    X = np.random.default_rng(8).uniform(-3, 3, size=(333, 1))
    psi = _lib.host_empty((333, 4), np.complex128)
    _lib.check(_lib.load().qk_simulate(h.handle, _lib.ptr(X), 333, 1, _lib.ptr(psi)))
    ref = O.simulate(compiled_to_oracle_ops(compiled), 2, X)
    assert np.abs(psi - ref).max() < 1e-12
