#!/usr/bin/env python3
"""Time the ablation builds of the Gram kernel (tools/gram_ablate.sh), one child process per build
(each loads its own library).  usage: python tools/gram_ablate.py --rows 16384 --qubits 12 0 1 2 3 7"""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=16384)
ap.add_argument("--qubits", type=int, default=12)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--child", type=int, default=-1)
ap.add_argument("masks", nargs="*", type=int)
a = ap.parse_args()

if a.child < 0:
    for m in a.masks:
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--rows", str(a.rows), "--qubits", str(a.qubits),
                              "--rounds", str(a.rounds), "--child", str(m)])
        if rc != 0:
            sys.exit(rc)
    sys.exit(0)

sys.path.insert(0, ROOT)
from pathlib import Path
import numpy as np, torch
from quake_amd import _lib
if a.child > 0:
    _lib.LIB_PATH = Path(ROOT) / "tools" / "_abl" / f"libquake_hip_abl{a.child}.so"
from quake_amd import device as qdev, featuremaps as fm
from quake_amd.circuit import ParameterVector
N, n = a.rows, a.qubits
X = torch.from_numpy(np.random.default_rng(1005).uniform(-1.5, 1.5, size=(N, n))).cuda()
psi = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 1)).simulate(X)
K = torch.empty((N, N), dtype=torch.float64, device="cuda")
ts = []
for r in range(a.rounds + 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); qdev.gram(psi, None, out=K); e1.record(); torch.cuda.synchronize()
    if r:
        ts.append(e0.elapsed_time(e1))
ms = float(np.median(ts))
hw = 6.0 * (1 << n) * N * (N + 1) / 2
print(json.dumps({"ablate_mask": a.child, "rows": N, "qubits": n, "ms_med": ms, "ms_min": float(min(ts)),
                  "hw_tflops": hw / ms / 1e9, "mfma_frac": hw / ms / 1e9 / 78.6}), flush=True)
