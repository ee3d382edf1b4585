#!/usr/bin/env python3
"""Rate of the FP64 Gram kernel by job shape at 16 qubits: ns per 1000 computed entries for symmetric and cross
jobs of different sizes, alone and combined in one launch (the shapes of a rank's share and of a host-path band)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import device as qdev, featuremaps as fm
from quake_amd.circuit import ParameterVector

n = int(os.environ.get("QUBITS", "16")); D = 1 << n
N = int(os.environ.get("ROWS", "32768"))
X = torch.from_numpy(np.random.default_rng(1).uniform(-1.5, 1.5, size=(N, n))).cuda()
circ = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 1))
psi = circ.simulate(X)
K = torch.empty((N, N), dtype=torch.float64, device="cuda")
KT = torch.empty((N, N), dtype=torch.float64, device="cuda")

def run(jobs, reps=2):
    def go(): qdev.gram_jobs(jobs, D)
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def sym(a0, a1): return dict(A=psi[a0:a1], B=psi[a0:a1], K=K[a0:a1, a0:a1], symmetric=True)
def cross(a0, a1, b0, b1, kt=False):
    j = dict(A=psi[a0:a1], B=psi[b0:b1], K=K[a0:a1, b0:b1], symmetric=False)
    if kt: j["KT"] = KT[b0:b1, a0:a1]
    return j
def entries(jobs): return sum((j["A"].shape[0] * (j["A"].shape[0] + 1) // 2) if j["symmetric"] else j["A"].shape[0] * j["B"].shape[0] for j in jobs)

h, q, e = N // 2, N // 4, N // 8
cases = {
    f"sym {N}": [sym(0, N)],
    f"sym {h}": [sym(0, h)],
    f"sym {q}": [sym(0, q)],
    f"sym {e}": [sym(0, e)],
    f"cross {h}x{h}": [cross(0, h, h, N)],
    f"cross {q}x{q}": [cross(0, q, q, h)],
    f"cross {e}x{N-e} (band)": [cross(0, e, e, N)],
    f"cross {e}x{N-e} (band) +KT": [cross(0, e, e, N, True)],
    f"band launch: sym {e} + cross {e}x{N-e} +KT": [sym(0, e), cross(0, e, e, N, True)],
    f"rank share G=4 of {N}: sym {q} + cross {q}x{q} + cross {e}x{q}": [sym(0, q), cross(0, q, q, h, True), cross(0, e, h, h + q, True)],
    f"4 x sym {q} in one launch": [sym(k * q, (k + 1) * q) for k in range(4)],
    f"4 x cross {q}x{q} in one launch": [cross(0, q, q, h), cross(q, h, h, h + q), cross(h, h + q, h + q, N), cross(h + q, N, 0, q)],
}
for name, jobs in cases.items():
    ms = run(jobs)
    ent = entries(jobs)
    print(json.dumps({"case": name, "ms": round(ms, 2), "ns_per_kentry": round(ms * 1e6 / (ent / 1e3), 3),
                      "alg_tflops": round(8.0 * D * ent / ms / 1e9, 1)}), flush=True)
