#!/usr/bin/env python3
"""Cost of the pieces of the wave-resident simulator: a ladder of small circuits at one width, one launch each
(run under rocprofv3 --pmc SQ_INSTS_VALU ... --kernel-trace to get instructions per sample per piece).
usage: python tools/wave_cost.py [--qubits 10] [--samples 65536]"""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import circuit as qc, device as qdev

ap = argparse.ArgumentParser()
ap.add_argument("--qubits", type=int, default=10)
ap.add_argument("--samples", type=int, default=65536)
a = ap.parse_args()
n, N = a.qubits, a.samples
x = qc.ParameterVector("x", n)


def base(param=False, asin=False):
    c = qc.QuantumCircuit(n)
    for q in range(n):
        if asin: c.ry(2 * np.arcsin(x[q] * 0.3), q)
        elif param: c.ry(x[q], q)
        else: c.h(q)
    if not param and not asin:
        c.p(x[0], 0)       # keeps the parameter count at one or more
    return c


def with_gates(kind, q, r, reps=4):
    c = base()
    for _ in range(reps):
        c.cx(q, r)
        if kind == "comp": c.rx(0.3, r); c.ry(0.4, r)
        elif kind == "h": c.h(r)
        elif kind == "rx": c.rx(0.3, r)
    return c


ladder = {"init_h": base(), "init_ry_param": base(param=True), "init_ry_asin": base(asin=True)}
for name, (q, r) in {"lane01": (0, 1), "lane45": (4, 5), "reg89": (n - 2, n - 1), "lane0_reg9": (0, n - 1)}.items():
    for kind in ("cx", "h", "rx", "comp"):
        ladder[f"{kind}_{name}"] = with_gates(kind, q, r)
for name, c in ladder.items():
    npar = c.num_parameters
    X = torch.from_numpy(np.random.default_rng(1).uniform(-1.0, 1.0, size=(N, max(npar, 1)))).cuda()
    dc = qdev.DeviceCircuit(c)
    psi = torch.empty((N, 1 << n), dtype=torch.complex128, device="cuda")
    dc.simulate(X, out=psi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dc.simulate(X, out=psi); e1.record(); torch.cuda.synchronize()
    print(json.dumps({"circuit": name, "ms": round(e0.elapsed_time(e1), 4), "plan": dc.handle.plan_summary[:0]}), flush=True)
