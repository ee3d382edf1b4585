#!/usr/bin/env python3
"""Gram stage alone at small sizes: time vs N at D = 2^n (tile-count quantisation, fixed costs)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import _lib
from quake_amd import device as qdev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
D = 1 << n
for N in [int(a) for a in sys.argv[2:]] or [1024, 2048, 3968, 4032, 4096, 4160, 8192]:
    psi = torch.randn((N, D), dtype=torch.complex128, device="cuda")
    psi /= psi.norm(dim=1, keepdim=True)
    K = torch.empty((N, N), dtype=torch.float64, device="cuda")
    qdev.gram(psi, None, out=K); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): qdev.gram(psi, None, out=K)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tiles = (N + 63) // 64
    live = tiles * (tiles + 1) // 2
    print(json.dumps({"N": N, "D": D, "ms": round(ms, 4), "live_tiles": live, "rounds_of_512": round(live / 512, 3),
                      "alg_TFLOPs": round(8.0 * D * N * (N + 1) / 2 / ms / 1e9, 2), "us_per_round": round(ms * 1e3 / np.ceil(live / 512), 2)}), flush=True)
