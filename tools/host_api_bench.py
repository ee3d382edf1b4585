#!/usr/bin/env python3
"""Wall time of the HOST entry point (numpy in, numpy out: H2D of X, simulate, Gram, D2H of K) next to
the device-resident stages -- the PCIe-inclusive rate DESIGN.md quotes."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import featuremaps as fm
from quake_amd.circuit import ParameterVector
from quake_amd.kernels import HipFidelityKernel
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=16384)
ap.add_argument("--qubits", type=int, default=12)
a = ap.parse_args()
N, n = a.rows, a.qubits
X = np.random.default_rng(0).uniform(-1.5, 1.5, size=(N, n))
k = HipFidelityKernel(fm.zz_featuremap(ParameterVector("x", n), n, 1))
k.evaluate(X[:256])
for _ in range(2):
    t0 = time.perf_counter(); K = k.evaluate(X); dt = time.perf_counter() - t0
print(json.dumps({"rows": N, "qubits": n, "host_call_s": dt, "entries_per_s": N * N / dt, "K_GiB": N * N * 8 / 2**30,
                  "effective_D2H_GBps_if_all_copy": N * N * 8 / dt / 1e9}))
