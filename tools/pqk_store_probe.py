#!/usr/bin/env python3
"""Does the symmetric projected-kernel launch wait for its mirrored stores?  The same rows as a symmetric block (half the
pairs computed, the other half written as mirrored 32-byte segments) and as a cross block (every pair computed, every store a
128-byte run): time per computed pair."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import device as qdev
N, n = 32768, 12
rng = np.random.default_rng(0)
r = rng.uniform(-0.5, 0.5, size=(N, n, 4))
rd = torch.from_numpy(r).cuda()
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
K = torch.empty((N, N), dtype=torch.float64, device="cuda")
sym = timed(lambda: qdev.pqk(rd, None, 1.0, out=K))
cross = timed(lambda: qdev.pqk(rd, rd, 1.0, out=K))
print(json.dumps({"N": N, "qubits": n, "sym_ms": sym, "cross_ms": cross, "sym_ns_per_computed_pair": sym * 1e6 / (N * (N + 1) / 2),
                  "cross_ns_per_computed_pair": cross * 1e6 / (N * N)}))
