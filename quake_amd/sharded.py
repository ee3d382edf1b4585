"""Row-block sharding of the train x train Gram matrix over the GPUs of one node.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI).
Rank g owns rows ``[g*N/G, (g+1)*N/G)``:

1./2. every rank needs all N statevectors.  The feature matrix X is tiny and the
   simulator is deterministic, so the default is that every rank simulates ALL rows
   itself (16 qubits, N = 65536: ~30 ms, against >= 160 ms to move the other ranks'
   56 GiB over xGMI) -- no collective on the data path.  For circuits deep enough
   that simulation costs more than the transfer, a rank simulates its own rows only
   and an all-gather of the Psi shards replicates them, overlapped with the diagonal
   block (``gather_work``);
3. it computes its share of the symmetric matrix.  Computing the full row block
   would do every off-diagonal entry twice across the node, so the blocks are dealt
   out circulantly instead: rank g computes its diagonal block (upper triangle),
   the blocks ``(g, g+d)`` for ``d = 1 .. ceil(G/2)-1`` and, for even G, half of
   block ``(g, g+G/2)``.  Every rank gets exactly 1/G of the N(N+1)/2 entries;
4. the kernel writes each off-diagonal entry twice -- into the rank's own row block
   and, transposed, into a staging buffer that is sent point-to-point to the rank
   owning the mirror rows (each ordered pair of ranks exchanges at most one block).

The result on rank g is the complete row block ``K[g*N/G:(g+1)*N/G, :]``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple


@dataclass(frozen=True)
class BlockTask:
    """Compute ``|Psi[a0:a1] Psi[b0:b1]^H|^2`` on this rank.

    ``peer`` is the rank that owns rows b0:b1 (== own rank for the diagonal block);
    for ``peer != rank`` the transposed block is sent there."""
    a0: int
    a1: int
    b0: int
    b1: int
    peer: int
    symmetric: bool


def row_range(rank: int, world: int, N: int) -> Tuple[int, int]:
    return (N // world) * rank, (N // world) * (rank + 1)


def check_shape(N: int, world: int) -> None:
    if N % world != 0:
        raise ValueError(f"N={N} must be divisible by the number of GPUs ({world})")
    if world % 2 == 0 and world > 1 and (N // world) % 2 != 0:
        raise ValueError("with an even number of GPUs each row block must have an even number of rows")


def plan_tasks(rank: int, world: int, N: int) -> List[BlockTask]:
    """Block tasks of one rank (see module docstring)."""
    check_shape(N, world)
    g0, g1 = row_range(rank, world, N)
    tasks = [BlockTask(g0, g1, g0, g1, rank, True)]
    for d in range(1, (world + 1) // 2):
        h = (rank + d) % world
        h0, h1 = row_range(h, world, N)
        tasks.append(BlockTask(g0, g1, h0, h1, h, False))
    if world % 2 == 0 and world > 1:
        h = (rank + world // 2) % world
        h0, h1 = row_range(h, world, N)
        if rank < world // 2:     # first half of my rows against all of the partner's rows
            tasks.append(BlockTask(g0, (g0 + g1) // 2, h0, h1, h, False))
        else:                     # all of my rows against the second half of the partner's rows
            tasks.append(BlockTask(g0, g1, (h0 + h1) // 2, h1, h, False))
    return tasks


def incoming(rank: int, world: int, N: int) -> List[Tuple[int, BlockTask]]:
    """(sender, the sender's task) for every mirror block this rank receives."""
    out = []
    for s in range(world):
        if s == rank:
            continue
        for t in plan_tasks(s, world, N):
            if t.peer == rank and not t.symmetric:
                out.append((s, t))
    return out


def computed_entries(rank: int, world: int, N: int) -> int:
    """Gram entries this rank computes (upper triangle incl. diagonal for its own block)."""
    n = 0
    for t in plan_tasks(rank, world, N):
        r, c = t.a1 - t.a0, t.b1 - t.b0
        n += r * (r + 1) // 2 if t.symmetric else r * c
    return n


class HipHostCopier:
    """The K-block -> host leg on a side stream (``qk_block_to_host_async``): ``fence()`` orders the side
    stream behind everything enqueued so far on torch's current stream, ``copy`` enqueues one strided
    block copy there, ``finish`` waits for the copies.  CPU rehearsals pass an object with the same three
    methods."""

    def __init__(self):
        import torch
        self._torch = torch
        self.stream = torch.cuda.Stream()

    def fence(self):
        ev = self._torch.cuda.Event()
        ev.record()
        self.stream.wait_event(ev)

    def copy(self, dev_view, host_view):
        from .device import block_to_host
        block_to_host(dev_view, host_view, self.stream)

    def finish(self):
        self.stream.synchronize()


class ShardedGram:
    """Executes the plan on torch tensors.  ``gram_jobs(jobs, D)`` defaults to the HIP
    launcher :func:`quake_amd.device.gram_jobs`."""

    def __init__(self, N: int, D: int, rank: int = 0, world: int = 1, group=None,
                 gram_jobs: Optional[Callable] = None, host_staging: bool = False):
        """``host_staging``: exchange the mirrored blocks through host memory (for a backend that cannot
        send device tensors point to point, e.g. gloo in a one-GPU rehearsal); RCCL sends them directly."""
        check_shape(N, world)
        self.N, self.D, self.rank, self.world, self.group = N, D, rank, world, group
        self.host_staging = host_staging
        if gram_jobs is None:
            from .device import gram_jobs as hip_gram_jobs
            gram_jobs = hip_gram_jobs
        self._gram_jobs = gram_jobs
        self.tasks = plan_tasks(rank, world, N)
        self.recv = incoming(rank, world, N)
        self._stage = None

    def allocate(self, device, torch):
        """Staging buffers for the transposed blocks (send) and the incoming ones (recv)."""
        f64 = torch.float64
        self._stage = {
            "send": [torch.empty((t.b1 - t.b0, t.a1 - t.a0), dtype=f64, device=device)
                     for t in self.tasks if not t.symmetric],
            "recv": [torch.empty((t.b1 - t.b0, t.a1 - t.a0), dtype=f64, device=device)
                     for _, t in self.recv],
        }

    def _jobs(self, psi_full, K_block, tasks, sends):
        g0, _ = row_range(self.rank, self.world, self.N)
        jobs = []
        for t in tasks:
            job = dict(A=psi_full[t.a0:t.a1], B=psi_full[t.b0:t.b1],
                       K=K_block[t.a0 - g0:t.a1 - g0, t.b0:t.b1], symmetric=t.symmetric)
            if not t.symmetric:
                job["KT"] = sends.pop(0)
            jobs.append(job)
        return jobs

    def compute(self, psi_full, K_block, gather_work=None, timer=None, K_host=None, copier=None):
        """psi_full: complex128 [N, D] (own shard valid; the rest valid once ``gather_work``
        completes).  K_block: float64 [N/G, N], fully written on return.
        K_host: optional host ``[N/G, N]`` float64 (page-locked): the rank's row block is also delivered
        there -- the blocks this rank computed itself are copied while the mirrored blocks are still
        travelling between the GPUs, the received ones once they have landed.  Complete on return."""
        import torch
        import torch.distributed as dist

        if self._stage is None:
            self.allocate(K_block.device, torch)
        if K_host is not None:
            if tuple(K_host.shape) != tuple(K_block.shape):
                raise ValueError("K_host must have the shape of K_block")
            if copier is None:
                copier = self._copier = getattr(self, "_copier", None) or HipHostCopier()
        sends = list(self._stage["send"])
        diag = [t for t in self.tasks if t.symmetric]
        rest = [t for t in self.tasks if not t.symmetric]
        if timer: timer("gram_begin")
        if gather_work is None:
            # every row of psi_full is already valid (each rank simulated all of them, or world == 1):
            # one launch over all blocks, so the tail of one block overlaps the start of the next
            self._gram_jobs(self._jobs(psi_full, K_block, diag + rest, sends), self.D)
        else:
            # the diagonal block only needs the local shard: it overlaps the all-gather
            self._gram_jobs(self._jobs(psi_full, K_block, diag, sends), self.D)
            gather_work.wait()
            if rest:
                self._gram_jobs(self._jobs(psi_full, K_block, rest, sends), self.D)
        if timer: timer("gram_end")
        g0, _ = row_range(self.rank, self.world, self.N)
        if K_host is not None:
            copier.fence()          # behind the Gram launches
            for t in self.tasks:    # what this rank computed itself (the diagonal block holds its own mirror)
                copier.copy(K_block[t.a0 - g0:t.a1 - g0, t.b0:t.b1], K_host[t.a0 - g0:t.a1 - g0, t.b0:t.b1])
        if self.world == 1:
            if K_host is not None:
                copier.finish()
            return K_block
        ops = []
        send_bufs, recv_bufs = self._stage["send"], self._stage["recv"]
        if self.host_staging:
            send_bufs = [b.cpu() for b in send_bufs]
            recv_bufs = [torch.empty(b.shape, dtype=b.dtype) for b in recv_bufs]
        for buf, t in zip(send_bufs, rest):
            ops.append(dist.P2POp(dist.isend, buf, t.peer, group=self.group))
        for buf, (src, _) in zip(recv_bufs, self.recv):
            ops.append(dist.P2POp(dist.irecv, buf, src, group=self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if self.host_staging:
            for dst, src in zip(self._stage["recv"], recv_bufs):
                dst.copy_(src)
        for buf, (_, t) in zip(self._stage["recv"], self.recv):
            # sender computed rows a0:a1 x my rows b0:b1 and sent the transpose [b, a]
            K_block[t.b0 - g0:t.b1 - g0, t.a0:t.a1].copy_(buf)
        if K_host is not None:
            copier.fence()          # behind the placement of the received blocks
            for _, t in self.recv:
                copier.copy(K_block[t.b0 - g0:t.b1 - g0, t.a0:t.a1], K_host[t.b0 - g0:t.b1 - g0, t.a0:t.a1])
            copier.finish()
        return K_block


def sharded_projected_kernel(rdm_local, gamma: float, rank: int = 0, world: int = 1, group=None, pqk_fn=None,
                             host_staging: bool = False):
    """Projected-kernel row block on one rank (config 4 of BASELINE.json).

    Each rank has simulated its row shard and reduced it to the per-qubit RDM table
    ``rdm_local`` (float64 ``[N/G, n, 4]``, a few hundred bytes per sample).  The ONE exchange is
    an all-gather of that table -- the statevectors never leave their GPU -- after which the rank
    evaluates its ``[N/G, N]`` block of ``exp(-gamma * sum_q ||rho_i - rho_j||_F)``.  The block on
    the diagonal is computed as a symmetric block so ``K[i, i]`` is exactly 1 as in the reference
    (genetic_v3.py:805).  ``pqk_fn(rdm_x, rdm_y_or_None, gamma, out)`` defaults to the HIP launcher."""
    import torch
    import torch.distributed as dist

    if pqk_fn is None:
        from .device import pqk as pqk_fn
    nloc = rdm_local.shape[0]
    rdm_local = rdm_local.contiguous()
    if world == 1:
        out = torch.empty((nloc, nloc), dtype=torch.float64, device=rdm_local.device)
        pqk_fn(rdm_local, None, gamma, out)
        return out
    table = torch.empty((world * nloc,) + tuple(rdm_local.shape[1:]), dtype=rdm_local.dtype, device=rdm_local.device)
    if host_staging:   # a backend without device collectives (one-GPU rehearsal over gloo)
        table_host = torch.empty(table.shape, dtype=table.dtype)
        dist.all_gather_into_tensor(table_host, rdm_local.cpu(), group=group)
        table.copy_(table_host)
    else:
        dist.all_gather_into_tensor(table, rdm_local, group=group)
    out = torch.empty((nloc, world * nloc), dtype=torch.float64, device=rdm_local.device)
    r0 = rank * nloc
    if r0 > 0:
        pqk_fn(rdm_local, table[:r0], gamma, out[:, :r0])
    pqk_fn(rdm_local, None, gamma, out[:, r0:r0 + nloc])
    if r0 + nloc < world * nloc:
        pqk_fn(rdm_local, table[r0 + nloc:], gamma, out[:, r0 + nloc:])
    return out


def sharded_cross_gram(psi_test_local, psi_train_full, gather_work=None, gram_jobs=None, out=None):
    """Test x train block of one rank (config 5's 16384 test rows): the test rows are sharded like the
    train rows, the replicated train statevectors (the all-gather of :class:`ShardedGram`, reused) are
    the other operand, so the rank's ``[T/G, N]`` block needs no further exchange.
    ``gather_work`` is the pending all-gather of ``psi_train_full``, if it has not been waited on."""
    import torch

    if gram_jobs is None:
        from .device import gram_jobs
    if gather_work is not None:
        gather_work.wait()
    T, D = psi_test_local.shape
    N = psi_train_full.shape[0]
    if out is None:
        out = torch.empty((T, N), dtype=torch.float64, device=psi_test_local.device)
    gram_jobs([dict(A=psi_test_local, B=psi_train_full, K=out, symmetric=False)], D)
    return out
