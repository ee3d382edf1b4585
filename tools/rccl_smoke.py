#!/usr/bin/env python3
"""RCCL smoke on however many GPUs the box has (one rank per GPU; a one-GPU box gives world_size 1): the calls of
the N > 1 data path in the forms bench.py / sharded.py use them -- init with device_id, the in-place all-gather of
statevector shards through a float64 view, point-to-point hand-over of a transposed K block, MAX all-reduce of the
step time, barrier.  usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1
--master-port 29511 tools/rccl_smoke.py"""
import json, os, sys, time
import torch, torch.distributed as dist

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0))
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
N, D = 64 * world, 256
psi = torch.zeros((N, D), dtype=torch.complex128, device=dev)
r0, r1 = rank * 64, (rank + 1) * 64
psi[r0:r1] = torch.full((64, D), complex(rank + 1, -(rank + 1)), dtype=torch.complex128, device=dev)
psi_real = torch.view_as_real(psi).view(N, 2 * D)
work = dist.all_gather_into_tensor(psi_real, psi_real[r0:r1], async_op=True)
work.wait()
ok_gather = all(bool((psi[k * 64:(k + 1) * 64] == complex(k + 1, -(k + 1))).all()) for k in range(world))
# hand-over of a block to the next rank (and from the previous one), as the mirror exchange does
send = torch.full((64, 32), float(rank), dtype=torch.float64, device=dev)
recv = torch.full((64, 32), -1.0, dtype=torch.float64, device=dev)
ok_p2p = True
if world > 1:
    ops = [dist.P2POp(dist.isend, send, (rank + 1) % world), dist.P2POp(dist.irecv, recv, (rank - 1) % world)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    ok_p2p = bool((recv == float((rank - 1) % world)).all())
torch.cuda.synchronize()
dist.barrier()
torch.cuda.synchronize()
t = torch.tensor([time.perf_counter() % 1.0 + rank], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
ok_max = float(t.item()) >= world - 1
if rank == 0:
    print(json.dumps({"backend": dist.get_backend(), "world": world, "all_gather_in_place": ok_gather, "p2p": ok_p2p,
                      "all_reduce_max": ok_max, "nccl_version": list(torch.cuda.nccl.version())}), flush=True)
dist.destroy_process_group()
sys.exit(0 if (ok_gather and ok_p2p and ok_max) else 1)
