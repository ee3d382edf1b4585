"""Device-resident entry points (torch tensors in HBM -> *_dev C ABI) against the oracle."""
import numpy as np
import pytest

from oracle import qk_oracle as O
from tests._helpers import synth_X
from tests.test_sharded_cpu import _rdm_table

pytestmark = pytest.mark.gpu


def _setup(n, N, builder, repeats=1, seed=3):
    import torch
    from quake_amd import device as dev, featuremaps as fm
    from quake_amd.circuit import ParameterVector
    X = synth_X(N, n, "scaled", seed)
    circ = dev.DeviceCircuit(getattr(fm, builder)(ParameterVector("x", n), n, repeats))
    oops = getattr(O, builder)(n, repeats)
    return torch, dev, circ, oops, X


@pytest.mark.parametrize("n,N", [(3, 50), (9, 70), (12, 40), (13, 9), (14, 5)])
def test_rdm1_and_pqk_blocks(n, N):
    torch, dev, circ, oops, X = _setup(n, N, "custom1_featuremap", 1)
    psi = circ.simulate(torch.from_numpy(X).cuda())
    rd = dev.rdm1(psi, n)
    ref = _rdm_table(O.statevectors(oops, n, X), n)
    assert np.abs(rd.cpu().numpy() - ref).max() < 1e-13
    Kref = O.projected_kernel(oops, n, X, 0.6)
    K = dev.pqk(rd, None, 0.6)
    assert np.abs(K.cpu().numpy() - Kref).max() < 1e-12
    # cross block into a strided view
    big = torch.full((N, N + 7), -2.0, dtype=torch.float64, device="cuda")
    m = N // 2
    dev.pqk(rd[:m], rd[m:], 0.6, out=big[:m, 3:3 + N - m])
    got = big.cpu().numpy()
    assert np.abs(got[:m, 3:3 + N - m] - Kref[:m, m:]).max() < 1e-12
    assert (got[:, :3] == -2).all() and (got[m:] == -2).all() and (got[:, 3 + N - m:] == -2).all()


def test_sharded_single_rank_paths():
    from quake_amd import sharded
    n, N = 8, 300
    torch, dev, circ, oops, X = _setup(n, N, "zz_featuremap", 2)
    psi = circ.simulate(torch.from_numpy(X).cuda())
    K = torch.empty((N, N), dtype=torch.float64, device="cuda")
    sharded.ShardedGram(N, 1 << n, 0, 1).compute(psi, K)
    assert np.abs(K.cpu().numpy() - O.evaluate(oops, n, X)).max() < 1e-12
    Kp = sharded.sharded_projected_kernel(dev.rdm1(psi, n), 1.0)
    assert np.abs(Kp.cpu().numpy() - O.projected_kernel(oops, n, X, 1.0)).max() < 1e-12


def test_multi_job_launch_matches_plan():
    """The 4-GPU plan of rank 1 executed on ONE GPU: every job of the rank in one launch, mirrored
    blocks land transposed in their staging buffers."""
    from quake_amd import sharded
    n, N, world, rank = 7, 4 * 96, 4, 1
    torch, dev, circ, oops, X = _setup(n, N, "zz_featuremap", 1)
    psi = circ.simulate(torch.from_numpy(X).cuda())
    Kref = O.evaluate(oops, n, X)
    tasks = sharded.plan_tasks(rank, world, N)
    g0, g1 = sharded.row_range(rank, world, N)
    Kb = torch.full((g1 - g0, N), -1.0, dtype=torch.float64, device="cuda")
    jobs, sends = [], []
    for t in tasks:
        j = dict(A=psi[t.a0:t.a1], B=psi[t.b0:t.b1], K=Kb[t.a0 - g0:t.a1 - g0, t.b0:t.b1], symmetric=t.symmetric)
        if not t.symmetric:
            j["KT"] = torch.empty((t.b1 - t.b0, t.a1 - t.a0), dtype=torch.float64, device="cuda")
            sends.append((t, j["KT"]))
        jobs.append(j)
    dev.gram_jobs(jobs, 1 << n)
    torch.cuda.synchronize()
    got = Kb.cpu().numpy()
    for t in tasks:
        assert np.abs(got[t.a0 - g0:t.a1 - g0, t.b0:t.b1] - Kref[t.a0:t.a1, t.b0:t.b1]).max() < 1e-12
    for t, buf in sends:
        assert np.abs(buf.cpu().numpy() - Kref[t.b0:t.b1, t.a0:t.a1]).max() < 1e-12


@pytest.mark.parametrize("n,N,fused", [(1, 5, True), (3, 70, True), (9, 33, True), (10, 17, True), (11, 13, True), (12, 9, True), (13, 5, True),
                                       (14, 4, False)])
def test_fused_simulate_rdm1(n, N, fused):
    """qk_simulate_rdm1_dev: the RDMs come straight from the LDS-resident state (no Psi in HBM) when
    the circuit is one chunk per sample; otherwise through the workspace path.  Same numbers."""
    import torch
    from quake_amd import _lib, device as qdev, featuremaps as fm
    from quake_amd.circuit import ParameterVector
    X = np.random.default_rng(40 + n).uniform(-1.5, 1.5, size=(N, n))
    circ = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 2))
    assert bool(_lib.load().qk_simulate_rdm1_is_fused(circ.handle.handle)) == fused
    Xd = torch.from_numpy(X).cuda()
    got = qdev.simulate_rdm1(circ, Xd).cpu().numpy()
    r = O.rdm1(O.statevectors(O.zz_featuremap(n, 2), n, X), n)
    want = np.stack([r[..., 0, 0].real, r[..., 1, 1].real, r[..., 0, 1].real, r[..., 0, 1].imag], axis=-1)
    assert np.abs(got - want).max() < 1e-13
    two_step = qdev.rdm1(circ.simulate(Xd), n).cpu().numpy()
    assert np.abs(got - two_step).max() < 1e-14      # two simulators (LDS-resident fused path, wave-resident kernel): different rounding


@pytest.mark.parametrize("n,N,workspace", [(8, 16384 + 5, 1), (10, 8192 + 3, 0), (12, 1024 + 1, 0), (13, 512 + 1, 0)])
def test_simulate_rdm1_routes(n, N, workspace):
    """The three routes of qk_simulate_rdm1_dev on a circuit with a layer of rotations between entangling diagonal stages
    (custom1 map, r = 2) at batch sizes where the library leaves the LDS simulator's fused mode: 10 qubits and
    up, the register-resident kernel's own reduced-density-matrix mode (no workspace); 8 qubits (packed samples), simulate
    + reduce through a workspace (qk_simulate_rdm1_wants_workspace).  What the library picks, the call without a workspace,
    simulate + rdm1 done by hand and the oracle agree; small batches of the same circuit ask for no workspace."""
    import torch
    from quake_amd import _lib, device as qdev, featuremaps as fm
    from quake_amd.circuit import ParameterVector
    from tests._helpers import psi_from_compiled
    lib = _lib.load()
    fmap = fm.custom1_featuremap(ParameterVector("x", n), n, 2)
    circ = qdev.DeviceCircuit(fmap)
    assert lib.qk_simulate_rdm1_is_fused(circ.handle.handle) == 1
    assert lib.qk_simulate_rdm1_wants_workspace(circ.handle.handle, N) == workspace
    assert lib.qk_simulate_rdm1_wants_workspace(circ.handle.handle, 64) == 0
    X = np.random.default_rng(60 + n).uniform(-1.0, 1.0, size=(N, circ.n_params))
    Xd = torch.from_numpy(X).cuda()
    got = qdev.simulate_rdm1(circ, Xd)
    no_ws = torch.empty_like(got)
    _lib.check(lib.qk_simulate_rdm1_dev(circ.handle.handle, Xd.data_ptr(), N, Xd.stride(0), no_ws.data_ptr(), None, None))
    by_hand = qdev.rdm1(circ.simulate(Xd), n)
    torch.cuda.synchronize()
    assert float((got - no_ws).abs().max()) < 1e-13 and float((got - by_hand).abs().max()) < 1e-13
    rows = np.unique(np.concatenate([[0, 1, N - 1], np.random.default_rng(n).integers(0, N, 5)]))
    r = O.rdm1(psi_from_compiled(fmap, X[rows]), n)
    want = np.stack([r[..., 0, 0].real, r[..., 1, 1].real, r[..., 0, 1].real, r[..., 0, 1].imag], axis=-1)
    assert np.abs(got[torch.from_numpy(rows).cuda()].cpu().numpy() - want).max() < 1e-13


@pytest.mark.parametrize("n,seed,N", [(10, 3, 4096 + 7), (11, 20, 2048 + 3), (12, 21, 1024 + 5)])
def test_simulate_rdm1_busy_circuit_uses_the_sixteen_amplitude_plan(n, seed, N):
    """A circuit full of matrix butterflies and swaps simulates with 8 amplitudes per thread, but the register-resident
    kernel's reduced-density-matrix mode needs 16: such circuits carry a second plan for that mode (qk_circuit::rdm_plan,
    "rdm_plan yes").  Large batch: the library's route (no workspace asked), simulate + rdm1 by hand (the 8-amplitude
    plan) and the oracle agree."""
    import torch
    from quake_amd import _lib, device as qdev
    from tests._helpers import psi_from_compiled
    from tests.test_gpu_parity import _gate_soup
    lib = _lib.load()
    c = _gate_soup(n, seed, 4 * n)
    circ = qdev.DeviceCircuit(c)
    s = circ.handle.plan_summary
    assert "rdm_plan yes" in s and f"workgroup of {(1 << n) // 8} threads" in s, s
    assert lib.qk_simulate_rdm1_wants_workspace(circ.handle.handle, N) == 0
    X = np.random.default_rng(80 + n).uniform(-1.0, 1.0, size=(N, circ.n_params))
    Xd = torch.from_numpy(X).cuda()
    got = qdev.simulate_rdm1(circ, Xd)
    by_hand = qdev.rdm1(circ.simulate(Xd), n)
    torch.cuda.synchronize()
    assert float((got - by_hand).abs().max()) < 1e-13
    rows = np.unique(np.concatenate([[0, 1, N - 1], np.random.default_rng(n).integers(0, N, 5)]))
    r = O.rdm1(psi_from_compiled(c, X[rows]), n)
    want = np.stack([r[..., 0, 0].real, r[..., 1, 1].real, r[..., 0, 1].real, r[..., 0, 1].imag], axis=-1)
    assert np.abs(got[torch.from_numpy(rows).cuda()].cpu().numpy() - want).max() < 1e-13


def test_k_block_to_host_leg():
    """ShardedGram.compute(K_host=...): the row block also lands in page-locked host memory through
    qk_block_to_host_async on a side stream (strided 2-D copies, no temporaries)."""
    from quake_amd import _lib, sharded
    n, N = 7, 520
    torch, dev, circ, oops, X = _setup(n, N, "zz_featuremap", 1)
    psi = circ.simulate(torch.from_numpy(X).cuda())
    K = torch.empty((N, N), dtype=torch.float64, device="cuda")
    K_host = torch.from_numpy(_lib.host_empty((N, N)))
    K_host.fill_(-3.0)
    sharded.ShardedGram(N, 1 << n, 0, 1).compute(psi, K, K_host=K_host)
    assert torch.equal(K_host, K.cpu())
    assert np.abs(K_host.numpy() - O.evaluate(oops, n, X)).max() < 1e-12
    # a strided sub-block into a strided host view; everything around it untouched
    host = torch.full((40, 90), -1.0, dtype=torch.float64)
    dev.block_to_host(K[100:130, 7:57], host[5:35, 20:70])
    torch.cuda.synchronize()
    assert torch.equal(host[5:35, 20:70], K[100:130, 7:57].cpu())
    assert (host[:5] == -1).all() and (host[35:] == -1).all() and (host[:, :20] == -1).all() and (host[:, 70:] == -1).all()


class _ForeignArray:
    """A non-torch DLPack producer: only the two protocol methods, as a cupy / jax / numba array would offer them."""
    def __init__(self, t):
        self._t = t

    def __dlpack__(self, stream=None, **kw):
        return self._t.__dlpack__(stream=stream) if stream is not None else self._t.__dlpack__()

    def __dlpack_device__(self):
        return self._t.__dlpack_device__()


def test_dlpack_hand_off_in_and_out():
    """Device entry points take any __dlpack__ producer or a raw capsule (zero copy) and hand back tensors that
    are DLPack producers themselves."""
    import torch
    from quake_amd import device as qdev
    n, N = 6, 90
    torch_, dev, circ, oops, X = _setup(n, N, "zz_featuremap", 2)
    Xd = torch.from_numpy(X).cuda()
    psi = circ.simulate(_ForeignArray(Xd))                                  # a foreign object in
    psi_caps = circ.simulate(torch.utils.dlpack.to_dlpack(Xd))              # a raw capsule in
    assert torch.equal(psi, psi_caps) and torch.equal(psi, circ.simulate(Xd))
    out = torch.empty((N, N), dtype=torch.float64, device="cuda")
    K = qdev.gram(_ForeignArray(psi), None, out=_ForeignArray(out))         # written through the foreign view: zero copy
    assert K.data_ptr() == out.data_ptr()
    assert np.abs(out.cpu().numpy() - O.evaluate(oops, n, X)).max() < 1e-12
    back = torch.from_dlpack(qdev.to_dlpack(K))                             # and out again as a capsule
    assert back.data_ptr() == K.data_ptr() and hasattr(K, "__dlpack__")
    rd = qdev.rdm1(_ForeignArray(psi), n)
    assert np.abs(qdev.pqk(_ForeignArray(rd), None, 0.5).cpu().numpy() - O.projected_kernel(oops, n, X, 0.5)).max() < 1e-12
    with pytest.raises(ValueError):
        circ.simulate(X)                                                    # numpy is a CPU DLPack producer: refused, no silent copy
    with pytest.raises(TypeError):
        qdev.gram([1, 2, 3])


def test_rccl_smoke_world_size_one():
    """The collectives of the N > 1 path, in the forms bench.py and sharded.py use them, over RCCL itself (backend
    "nccl") with as many ranks as this box allows: one.  A child process (this one has initialised the GPU)."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "tools", "rccl_smoke.py")],
                       capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["backend"] == "nccl" and out["all_gather_in_place"] and out["p2p"] and out["all_reduce_max"]


@pytest.mark.gpu
@pytest.mark.parametrize("n,N,seed", [(8, 16384 + 5, 101), (10, 4096 + 3, 100), (11, 2048 + 1, 103), (12, 1024 + 5, 102), (12, 7, 102)])
def test_simulate_rdm1_hardware_native_genome(n, N, seed):
    """The reduced density matrices of GA genomes of the hardware-native alphabet: their ecr gates on qubits that are still
    basis states become single-qubit gates and a constant angle program of the planner's own (qk_circuit.cpp section 1),
    and a circuit that simulates with 8 amplitudes per thread carries a second plan for the fused mode, made from the
    caller's tables.  The library's route, simulate + rdm1 by hand and the oracle agree."""
    import torch
    from quake_amd import device as qdev, genetic
    from tests._helpers import psi_from_compiled
    hw = {"single_non_parametric": ["Id", "X", "SX"], "single_parametric": ["RZ", "RZ"], "two_non_parametric": ["ECR"], "two_parametric": []}
    genes = genetic.initial_population(1, n, 8, hw, n, rng=np.random.default_rng(seed))[0]
    fmap, cols = genetic.to_quantum(genes, hw, n, 8, n, physical_qubits=list(range(n)), verbose=False)
    circ = qdev.DeviceCircuit(fmap)
    X = np.ascontiguousarray(np.random.default_rng(seed + n).uniform(0.02, 0.98, size=(N, n))[:, cols])
    Xd = torch.from_numpy(X).cuda()
    got = qdev.simulate_rdm1(circ, Xd)
    by_hand = qdev.rdm1(circ.simulate(Xd), n)
    torch.cuda.synchronize()
    assert float((got - by_hand).abs().max()) < 1e-13
    rows = np.unique(np.concatenate([[0, 1, N - 1], np.random.default_rng(n).integers(0, N, 4)]))
    r = O.rdm1(psi_from_compiled(fmap, X[rows]), n)
    want = np.stack([r[..., 0, 0].real, r[..., 1, 1].real, r[..., 0, 1].real, r[..., 0, 1].imag], axis=-1)
    assert np.abs(got[torch.from_numpy(rows).cuda()].cpu().numpy() - want).max() < 1e-13
