#!/usr/bin/env python3
"""Time qk_rdm1_dev alone (one read of Psi: 16 * 2^n bytes per sample)."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import device as qdev
ap = argparse.ArgumentParser()
ap.add_argument("--qubits", type=int, nargs="+", default=[12])
ap.add_argument("--gib", type=float, default=2.0)
a = ap.parse_args()
for n in a.qubits:
    N = int(a.gib * 2**30 / (16 << n))
    psi = torch.randn((N, 1 << n), dtype=torch.complex128, device="cuda")
    qdev.rdm1(psi, n); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): qdev.rdm1(psi, n)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(json.dumps({"n": n, "N": N, "ms": round(ms, 3), "read_GBps": round(N * (16 << n) / ms / 1e6, 1)}), flush=True)
