#!/usr/bin/env python3
"""Gram time against the head start given to the second workgroup of each CU (QK_GRAM_STAGGER, read once per
process: one child per value).  usage: python tools/gram_stagger.py --rows 4096 --qubits 8 -- -1 0 3 6 9 12"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=4096)
ap.add_argument("--qubits", type=int, default=8)
ap.add_argument("--rounds", type=int, default=15)
ap.add_argument("--child", action="store_true")
ap.add_argument("values", nargs="*", type=int)
a = ap.parse_args()
if not a.child:
    for v in a.values:
        env = dict(os.environ, QK_GRAM_STAGGER=str(v))
        if v < 0:
            env.pop("QK_GRAM_STAGGER")
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--rows", str(a.rows), "--qubits", str(a.qubits),
                              "--rounds", str(a.rounds), "--child"], env=env)
        if rc:
            sys.exit(rc)
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np, torch
from quake_amd import device as qdev, featuremaps as fm
from quake_amd.circuit import ParameterVector
N, n = a.rows, a.qubits
X = torch.from_numpy(np.random.default_rng(1005).uniform(-1.5, 1.5, size=(N, n))).cuda()
psi = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 2)).simulate(X)
K = torch.empty((N, N), dtype=torch.float64, device="cuda")
ts = []
for r in range(a.rounds + 3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); qdev.gram(psi, None, out=K); e1.record(); torch.cuda.synchronize()
    if r >= 3:
        ts.append(e0.elapsed_time(e1))
ms = float(np.median(ts))
hw = 6.0 * (1 << n) * N * (N + 1) / 2
print(json.dumps({"stagger": os.environ.get("QK_GRAM_STAGGER", "auto"), "rows": N, "qubits": n, "ms_med": ms,
                  "ms_min": float(min(ts)), "mfma_frac": hw / ms / 1e9 / 78.6}), flush=True)
