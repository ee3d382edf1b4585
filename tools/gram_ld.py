#!/usr/bin/env python3
"""Gram time against the row stride of Psi (experiment: QK_GRAM_LD makes the lean / ring kernels read rows
`ld` amplitudes apart).  usage: python tools/gram_ld.py --rows 16384 --qubits 16 --variant lean3m_64x64_w4_s0 -- 0 16 272"""
import argparse, ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=16384)
ap.add_argument("--qubits", type=int, default=16)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--variant", default="lean3m_64x64_w4_s0")
ap.add_argument("--child", type=int, default=-1)
ap.add_argument("pads", nargs="*", type=int)
a = ap.parse_args()
D = 1 << a.qubits
if a.child < 0:
    for pad in a.pads:
        env = dict(os.environ, QK_GRAM_LD=str(D + pad), QK_GRAM_VARIANT=a.variant)
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--rows", str(a.rows), "--qubits", str(a.qubits),
                              "--rounds", str(a.rounds), "--variant", a.variant, "--child", str(pad)], env=env)
        if rc:
            sys.exit(rc)
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np, torch
from quake_amd import _lib, device as qdev, featuremaps as fm
from quake_amd.circuit import ParameterVector
lib = _lib.load()
N, n, pad = a.rows, a.qubits, a.child
X = torch.from_numpy(np.random.default_rng(1005).uniform(-1.5, 1.5, size=(N, n))).cuda()
psi = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 1)).simulate(X)     # [N, D] complex128
big = torch.zeros((N, D + pad), dtype=torch.complex128, device="cuda")
big[:, :D] = psi
del psi
K = torch.empty((N, N), dtype=torch.float64, device="cuda")
ts = []
for r in range(a.rounds + 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.qk_gram_dev(C.c_void_p(big.data_ptr()), N, None, 0, D, C.c_void_p(K.data_ptr()), N, None))
    e1.record(); torch.cuda.synchronize()
    if r:
        ts.append(e0.elapsed_time(e1))
# spot check against a direct product on a few rows
i = torch.tensor([0, 1, 77, N - 1], device="cuda")
ref = (big[i, :D].conj() @ big[:, :D].T).abs() ** 2
err = float((K[i] - ref).abs().max())
ms = float(np.median(ts))
hw = 6.0 * D * N * (N + 1) / 2
print(json.dumps({"variant": a.variant, "pad": pad, "ms_med": ms, "mfma_frac": hw / ms / 1e9 / 78.6, "max_err_rows": err}), flush=True)
