#!/usr/bin/env python3
"""Same-box A/B of two builds of the library on the symmetric Gram stage at several sizes: QK_OLD=1 loads
quake_amd/libquake_hip_before.so (a build of another commit, copied there by hand -- *.so files travel to the GPU box but
stay out of git), otherwise the tree's library.  Run both in ONE gpurun call, alternating."""
import sys, os, json
sys.path.insert(0, os.getcwd())
from pathlib import Path
from quake_amd import _lib
if os.environ.get("QK_OLD"): _lib.LIB_PATH = Path(os.getcwd()) / "quake_amd" / "libquake_hip_before.so"
import numpy as np, torch
from quake_amd import device as qdev
for n, Ns in ((8, [4096, 8192, 16384]), (10, [4096, 16384]), (12, [16384]), (16, [16384])):
    D = 1 << n
    for N in Ns:
        psi = torch.randn((N, D), dtype=torch.complex128, device="cuda"); psi /= psi.norm(dim=1, keepdim=True)
        K = torch.empty((N, N), dtype=torch.float64, device="cuda")
        for _ in range(2): qdev.gram(psi, None, out=K)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); qdev.gram(psi, None, out=K); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[len(ts) // 2]
        print(json.dumps({"lib": "old" if os.environ.get("QK_OLD") else "new", "n": n, "N": N, "ms": round(ms, 4), "mfma_frac": round(6.0 * D * N * (N + 1) / 2 / ms / 1e9 / 78.6, 3)}), flush=True)
        del psi, K
