#!/usr/bin/env python3
"""Time the FP32 Gram kernel for the tile shape given by QK_GRAM32_TILE (one process per shape)."""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import device as qdev, featuremaps as fm
from quake_amd.circuit import ParameterVector
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=32768); ap.add_argument("--qubits", type=int, default=16)
a = ap.parse_args()
N, n, D = a.rows, a.qubits, 1 << a.qubits
X = torch.from_numpy(np.random.default_rng(1005).uniform(-1.5, 1.5, size=(N, n))).cuda()
psi = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 1)).simulate(X, dtype=torch.complex64)
K = torch.empty((N, N), dtype=torch.float64, device="cuda")
qdev.gram(psi, out=K); torch.cuda.synchronize()
ts = []
for _ in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); qdev.gram(psi, out=K); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = min(ts)
print(json.dumps({"tile": os.environ.get("QK_GRAM32_TILE", "auto"), "rows": N, "n": n, "ms": round(ms, 2),
                  "tflops_alg": round(8.0 * D * N * (N + 1) / 2 / ms / 1e9, 1)}), flush=True)
