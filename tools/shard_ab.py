#!/usr/bin/env python3
"""Time one rank's share of the sharded N x N Gram (world ranks, rank r) on ONE GPU, per kernel variant."""
import argparse, ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import _lib, device as qdev, featuremaps as fm, sharded
from quake_amd.circuit import ParameterVector
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=65536); ap.add_argument("--qubits", type=int, default=16)
ap.add_argument("--world", type=int, default=8); ap.add_argument("--rank", type=int, default=0)
ap.add_argument("variants", nargs="*")
a = ap.parse_args()
lib = _lib.load(); lib.qk_gram_set_variant.argtypes = [C.c_char_p]; lib.qk_gram_set_variant.restype = C.c_int
N, n, D = a.rows, a.qubits, 1 << a.qubits
X = torch.from_numpy(np.random.default_rng(1005).uniform(-1.5, 1.5, size=(N, n))).cuda()
psi = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 1)).simulate(X)
r0, r1 = sharded.row_range(a.rank, a.world, N)
K = torch.empty((r1 - r0, N), dtype=torch.float64, device="cuda")
sg = sharded.ShardedGram(N, D, a.rank, a.world); sg.allocate(K.device, torch)
flops = 8.0 * D * sharded.computed_entries(a.rank, a.world, N)
for v in a.variants * 2:
    assert lib.qk_gram_set_variant(v.encode()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    jobs = sg._jobs(psi, K, sg.tasks, list(sg._stage["send"]))
    qdev.gram_jobs(jobs, D)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(json.dumps({"variant": v, "world": a.world, "ms": round(ms, 2), "tflops_alg": round(flops / ms / 1e9, 2)}), flush=True)
