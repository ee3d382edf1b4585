"""Multi-GPU row-block sharding, rehearsed on CPU: the plan is checked for coverage
and balance for 1..8 ranks, and the exchange (all-gather of Psi shards + point-to-point
hand-over of the mirrored blocks) runs for real over gloo with 2 and 4 processes.
The per-block arithmetic is the numpy oracle here (test stand-in); on the GPU box the
same class launches the HIP kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import qk_oracle as O
from quake_amd import sharded


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 8])
def test_plan_covers_every_entry_once(world):
    N = 16 * world
    cover = np.zeros((N, N), dtype=int)
    per_rank = []
    for r in range(world):
        for t in sharded.plan_tasks(r, world, N):
            if t.symmetric:
                iu = np.triu_indices(t.a1 - t.a0)
                cover[t.a0 + iu[0], t.b0 + iu[1]] += 1
                il = np.tril_indices(t.a1 - t.a0, -1)
                cover[t.a0 + il[0], t.b0 + il[1]] += 1       # mirrored inside the kernel
            else:
                cover[t.a0:t.a1, t.b0:t.b1] += 1
                cover[t.b0:t.b1, t.a0:t.a1] += 1             # transposed block sent to the peer
                assert sharded.row_range(t.peer, world, N)[0] <= t.b0 < t.b1 <= sharded.row_range(t.peer, world, N)[1]
        per_rank.append(sharded.computed_entries(r, world, N))
    assert (cover == 1).all()
    assert sum(per_rank) == N * (N + 1) // 2
    assert max(per_rank) - min(per_rank) == 0          # perfectly balanced
    for r in range(world):                              # at most one message per ordered pair of ranks
        senders = [s for s, _ in sharded.incoming(r, world, N)]
        assert len(senders) == len(set(senders))


def test_shape_checks():
    with pytest.raises(ValueError):
        sharded.check_shape(10, 4)
    with pytest.raises(ValueError):
        sharded.check_shape(6, 2)     # 3 rows per block cannot be halved


def _cpu_gram_jobs(jobs, D):
    for j in jobs:
        A, B = j["A"].numpy(), j["B"].numpy()
        K = np.abs(A.conj() @ B.T) ** 2
        if j.get("symmetric"):
            np.fill_diagonal(K, 1.0)
        j["K"].copy_(torch.from_numpy(K))
        if j.get("KT") is not None:
            j["KT"].copy_(torch.from_numpy(np.ascontiguousarray(K.T)))


class _CpuCopier:
    """Stand-in for sharded.HipHostCopier (a side stream on the GPU): copies at once, counts the calls."""
    def __init__(self):
        self.fences, self.copies, self.finished = 0, 0, 0

    def fence(self):
        self.fences += 1

    def copy(self, dev_view, host_view):
        host_view.copy_(dev_view)
        self.copies += 1

    def finish(self):
        self.finished += 1


def _worker(rank, world, port, N, n, replicate=False, host_staging=False, k_host=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = np.random.default_rng(5).uniform(-1.5, 1.5, size=(N, n))
        ops = O.zz_featuremap(n, 1)
        r0, r1 = sharded.row_range(rank, world, N)
        D = 1 << n
        if replicate:   # bench.py's default: every rank simulates all rows, no collective before the Gram
            psi_full = torch.from_numpy(O.statevectors(ops, n, X))
            work = None
        else:
            psi_full = torch.zeros((N, D), dtype=torch.complex128)
            local = torch.from_numpy(O.statevectors(ops, n, X[r0:r1]))
            psi_full[r0:r1] = local
            work = dist.all_gather_into_tensor(psi_full, local, async_op=True)
        sg = sharded.ShardedGram(N, D, rank, world, gram_jobs=_cpu_gram_jobs, host_staging=host_staging)
        K_block = torch.full((r1 - r0, N), -1.0, dtype=torch.float64)
        K_host = torch.full((r1 - r0, N), -2.0, dtype=torch.float64) if k_host else None
        copier = _CpuCopier() if k_host else None
        sg.compute(psi_full, K_block, gather_work=work, K_host=K_host, copier=copier)
        ref = O.evaluate(ops, n, X)[r0:r1]
        err = np.abs(K_block.numpy() - ref).max()
        assert err < 1e-13, err
        if k_host:   # the K block -> host leg delivers every entry of the row block exactly once
            assert torch.equal(K_host, K_block)
            assert copier.copies == len(sg.tasks) + len(sg.recv) and copier.finished == 1
            assert copier.fences == (2 if world > 1 else 1)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_gram_over_gloo(world):
    mp.spawn(_worker, args=(world, _free_port(), 8 * world, 4), nprocs=world, join=True)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gram_replicated_psi_over_gloo(world):
    mp.spawn(_worker, args=(world, _free_port(), 8 * world, 4, True), nprocs=world, join=True)


def test_sharded_gram_host_staged_exchange_over_gloo():
    """bench.py's one-GPU rehearsal mode: the mirrored blocks travel through host buffers."""
    mp.spawn(_worker, args=(4, _free_port(), 32, 4, True, True), nprocs=4, join=True)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_gram_k_block_to_host_over_gloo(world):
    mp.spawn(_worker, args=(world, _free_port(), 8 * world, 4, True, False, True), nprocs=world, join=True)


def test_k_block_to_host_single_rank():
    _worker(0, 1, _free_port(), 12, 3, True, False, True)


def _cpu_pqk(rdm_x, rdm_y, gamma, out):
    x = rdm_x.numpy()
    y = x if rdm_y is None else rdm_y.numpy()
    d = x[:, None] - y[None]                                   # (Nx, Ny, n, 4)
    fro = np.sqrt(d[..., 0] ** 2 + d[..., 1] ** 2 + 2 * (d[..., 2] ** 2 + d[..., 3] ** 2))
    K = np.exp(-gamma * fro.sum(axis=2))
    if rdm_y is None:
        np.fill_diagonal(K, 1.0)
    out.copy_(torch.from_numpy(K))
    return out


def _rdm_table(psi, n):
    r = O.rdm1(psi, n)
    return np.stack([r[..., 0, 0].real, r[..., 1, 1].real, r[..., 0, 1].real, r[..., 0, 1].imag], axis=-1)


def _pqk_worker(rank, world, port, N, n):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = np.random.default_rng(9).uniform(-1.5, 1.5, size=(N, n))
        ops = O.zz_featuremap(n, 1)
        r0, r1 = sharded.row_range(rank, world, N)
        local = torch.from_numpy(np.ascontiguousarray(_rdm_table(O.statevectors(ops, n, X[r0:r1]), n)))
        K = sharded.sharded_projected_kernel(local, 0.8, rank, world, pqk_fn=_cpu_pqk)
        ref = O.projected_kernel(ops, n, X, 0.8)[r0:r1]
        assert np.abs(K.numpy() - ref).max() < 1e-13
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_projected_kernel_over_gloo(world):
    mp.spawn(_pqk_worker, args=(world, _free_port(), 6 * world, 3), nprocs=world, join=True)


def _cross_worker(rank, world, port, N, T, n):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(21)
        X, Xt = rng.uniform(-1.5, 1.5, size=(N, n)), rng.uniform(-1.5, 1.5, size=(T, n))
        ops = O.zz_featuremap(n, 1)
        r0, r1 = sharded.row_range(rank, world, N)
        t0, t1 = sharded.row_range(rank, world, T)
        psi_full = torch.zeros((N, 1 << n), dtype=torch.complex128)
        local = torch.from_numpy(O.statevectors(ops, n, X[r0:r1]))
        psi_full[r0:r1] = local
        work = dist.all_gather_into_tensor(psi_full, local, async_op=True)
        psi_test = torch.from_numpy(O.statevectors(ops, n, Xt[t0:t1]))
        Kt = sharded.sharded_cross_gram(psi_test, psi_full, gather_work=work, gram_jobs=_cpu_gram_jobs)
        assert np.abs(Kt.numpy() - O.evaluate(ops, n, Xt, X)[t0:t1]).max() < 1e-13
    finally:
        dist.destroy_process_group()


def test_sharded_test_block_over_gloo():
    mp.spawn(_cross_worker, args=(2, _free_port(), 12, 6, 3), nprocs=2, join=True)
