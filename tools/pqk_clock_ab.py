#!/usr/bin/env python3
"""Why does the projected kernel's pairwise launch run at ~2.0 GHz?  Same process, same
inputs (configs[3]: N = 32768, 12 qubits): the launch timed (a) cold, after the GPU idled for a second,
(b) as the k-th of 40 back-to-back launches (a clock ramp would show as a falling series), (c) right after
300 ms of FP64 matrix load (clocks up, chip warm).  If (b)'s tail and (c) equal (a), it is not a ramp: the
chip holds that clock under this instruction mix (FP64 vector ALU on all 1024 SIMDs)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import device as qdev, featuremaps as fm
from quake_amd.circuit import ParameterVector

N, n = 32768, 12
X = torch.from_numpy(np.random.default_rng(1004).uniform(-1.5, 1.5, size=(N, n))).cuda()
circ = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 1))
rdm = qdev.simulate_rdm1(circ, X)
K = torch.empty((N, N), dtype=torch.float64, device="cuda")

def one():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); qdev.pqk(rdm, None, 1.0, out=K); e1.record()
    return e0, e1

qdev.pqk(rdm, None, 1.0, out=K); torch.cuda.synchronize()
out = {}
cold = []
for _ in range(3):
    time.sleep(1.0)
    e = one(); torch.cuda.synchronize(); cold.append(e[0].elapsed_time(e[1]))
out["cold_after_1s_idle_ms"] = cold
ev = [one() for _ in range(40)]
torch.cuda.synchronize()
out["back_to_back_ms"] = [round(a.elapsed_time(b), 3) for a, b in ev]
A = torch.randn(8192, 8192, dtype=torch.float64, device="cuda")
warm = []
for _ in range(3):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        (A @ A); torch.cuda.synchronize()
    (A @ A)
    e = one(); torch.cuda.synchronize(); warm.append(e[0].elapsed_time(e[1]))
out["right_after_fp64_matmul_load_ms"] = warm
print(json.dumps(out))
