#!/usr/bin/env python3
"""A/B timing of Gram-kernel variants in ONE process, interleaved rounds (guide rule 24).
usage: python tools/gram_ab.py [--rows 16384] [--qubits 16] [--rounds 3] [variant ...]"""
import argparse, ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import _lib
from quake_amd import device as qdev, featuremaps as fm
from quake_amd.circuit import ParameterVector

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=16384)
ap.add_argument("--qubits", type=int, default=16)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--cross", action="store_true", help="non-symmetric N x N block instead of the symmetric one")
ap.add_argument("variants", nargs="*")
a = ap.parse_args()
lib = _lib.load()
lib.qk_gram_set_variant.argtypes = [C.c_char_p]
lib.qk_gram_set_variant.restype = C.c_int
N, n = a.rows, a.qubits
D = 1 << n
X = torch.from_numpy(np.random.default_rng(1005).uniform(-1.5, 1.5, size=(N, n))).cuda()
psi = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 1)).simulate(X)
psi2 = psi.clone() if a.cross else None
K = torch.empty((N, N), dtype=torch.float64, device="cuda")
entries = N * N if a.cross else N * (N + 1) // 2
flops = 8.0 * D * entries
ref = None
times = {v: [] for v in a.variants}
for rnd in range(a.rounds + 1):
    for v in a.variants:
        assert lib.qk_gram_set_variant(v.encode()) == 0, v
        K.fill_(-1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        qdev.gram(psi, psi2, out=K)
        e1.record()
        torch.cuda.synchronize()
        if rnd == 0:
            if ref is None:
                ref = K.clone()
                print(f"{v}: reference, min {float(K.min()):.3e} max {float(K.max()):.3e}", flush=True)
            else:
                print(f"{v}: max |K - K_ref| = {float((K - ref).abs().max()):.3e}", flush=True)
        else:
            times[v].append(e0.elapsed_time(e1))
for v in a.variants:
    t = np.array(times[v])
    print(json.dumps({"variant": v, "ms_med": float(np.median(t)), "ms_min": float(t.min()),
                      "tflops_med": flops / np.median(t) / 1e9, "frac_peak": flops / np.median(t) / 1e9 / 78.6}), flush=True)
