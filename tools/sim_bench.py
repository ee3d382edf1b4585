#!/usr/bin/env python3
"""Time the batched statevector simulator: algorithmic bytes = 16 * 2^n per sample (one write of Psi)."""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import device as qdev, featuremaps as fm
from quake_amd.circuit import ParameterVector, QuantumCircuit

def circuits(n):
    x = ParameterVector("x", n)
    h = QuantumCircuit(n)
    for q in range(n): h.h(q)
    hp = fm.z_featuremap(x, n, 1)
    return {"h_only": h, "z_r1": hp, "zz_r1": fm.zz_featuremap(x, n, 1), "zz_r2": fm.zz_featuremap(x, n, 2),
            "custom1_r1": fm.custom1_featuremap(x, n, 1), "custom2_r1": fm.custom2_featuremap(x, n, 1)}

ap = argparse.ArgumentParser()
ap.add_argument("--qubits", type=int, nargs="+", default=[16, 13, 12, 10, 8])
ap.add_argument("--gib", type=float, default=4.0, help="Psi bytes per run")
ap.add_argument("--maps", nargs="*", default=None)
a = ap.parse_args()
for n in a.qubits:
    N = int(a.gib * 2**30 / (16 << n))
    X = torch.from_numpy(np.random.default_rng(1).uniform(-1.5, 1.5, size=(N, n))).cuda()
    psi = torch.empty((N, 1 << n), dtype=torch.complex128, device="cuda")
    for name, c in circuits(n).items():
        if a.maps and name not in a.maps: continue
        dc = qdev.DeviceCircuit(c)
        Xn = X[:, :dc.n_params].contiguous() if dc.n_params else X[:, :0].contiguous()
        dc.simulate(Xn, out=psi); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): dc.simulate(Xn, out=psi)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(json.dumps({"n": n, "map": name, "N": N, "passes": dc.handle.num_passes, "ms": round(ms, 3),
                          "alg_GBps": round(N * (16 << n) / ms / 1e6, 1), "frac_of_8TBps": round(N * (16 << n) / ms / 1e6 / 8000, 4)}), flush=True)
