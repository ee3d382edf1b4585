#!/usr/bin/env python3
"""Wall time of one host call (HipFidelityKernel.evaluate, numpy in / numpy out) at the sizes the
reference's own runs use (N = 100 ... 2000 rows, 4 ... 10 qubits): launch / allocation overhead, not
throughput."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import featuremaps as fm
from quake_amd.circuit import ParameterVector
from quake_amd.kernels import HipFidelityKernel, projected_quantum_kernel
for N, n in ((100, 4), (100, 8), (500, 8), (1000, 8), (2000, 10)):
    X = np.random.default_rng(0).uniform(-1.5, 1.5, size=(N, n))
    circ = fm.zz_featuremap(ParameterVector("x", n), n, 2)
    k = HipFidelityKernel(circ)
    k.evaluate(X)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps): k.evaluate(X)
    sym = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): k.evaluate(X[: N // 4], X)
    cross = (time.perf_counter() - t0) / reps
    projected_quantum_kernel(circ, X, 1.0)
    t0 = time.perf_counter()
    for _ in range(reps): projected_quantum_kernel(circ, X, 1.0)
    pq = (time.perf_counter() - t0) / reps
    print(json.dumps({"rows": N, "qubits": n, "evaluate_sym_ms": round(sym * 1e3, 3), "evaluate_cross_ms": round(cross * 1e3, 3),
                      "projected_ms": round(pq * 1e3, 3)}), flush=True)
