#!/usr/bin/env python3
"""The reference's own GA workload (examples/genetic_featuremap_v3.py: 30 chromosomes, N = 100 rows, 4 qubits,
8 gates per qubit): one generation through qk_evaluate_batch vs one qk_evaluate call per chromosome."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import genetic
from quake_amd.kernels import HipFidelityKernel, evaluate_population
gd = {"single_non_parametric": ["Id", "X", "SX"], "single_parametric": ["RZ", "RZ"], "two_non_parametric": ["ECR"], "two_parametric": []}
X = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "data_cv.npy"))[:, [0, 5, 6, 8, 10, 12, 14, 15]].astype(np.float64)
y = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "data_labels.npy")).astype(np.int32)
pop = genetic.initial_population(30, 4, 8, gd, 8, rng=np.random.default_rng(0))
dec = [genetic.to_quantum(s, gd, 8, 8, 4, verbose=False) for s in pop]
maps, xs = [m for m, _ in dec], [X[:, i] for _, i in dec]
evaluate_population(maps, xs, labels=y)
t0 = time.perf_counter()
for _ in range(20): evaluate_population(maps, xs, labels=y, noise=("gaussian", 0.044))
tb = (time.perf_counter() - t0) / 20
[HipFidelityKernel(m).evaluate(x) for m, x in zip(maps, xs)]
t0 = time.perf_counter()
for _ in range(20): [HipFidelityKernel(m).evaluate(x) for m, x in zip(maps, xs)]
tl = (time.perf_counter() - t0) / 20
print(json.dumps({"chromosomes": 30, "rows": 100, "qubits": 4, "batch_ms_per_generation": tb * 1e3,
                  "loop_ms_per_generation": tl * 1e3, "batch_us_per_chromosome": tb / 30 * 1e6}))
# the library call alone (circuits already lowered: what is left once Python's share is taken out)
import ctypes as C
from quake_amd import _lib
lib = _lib.load()
handles = [_lib.CircuitHandle(m.compile()) for m in maps]
xs64 = [np.ascontiguousarray(x, dtype=np.float64) for x in xs]
P = C.c_void_p
G = len(maps)
circ_arr = (P * G)(*[h.handle for h in handles])
x_arr = (P * G)(*[x.ctypes.data for x in xs64])
n_arr = (C.c_int64 * G)(*[x.shape[0] for x in xs64])
lab_arr = (P * G)(*[y.ctypes.data] * G)
sums = np.zeros((G, 8))
opts = C.pointer(_lib.qk_batch_opts(_lib.NOISE_GAUSSIAN, 0, 0.044, 1))
call = lambda: _lib.check(lib.qk_evaluate_batch(G, circ_arr, x_arr, n_arr, lab_arr, opts, None, sums.ctypes.data))
call()
t0 = time.perf_counter()
for _ in range(50): call()
tc = (time.perf_counter() - t0) / 50
print(json.dumps({"library_call_ms_per_generation": tc * 1e3, "library_us_per_chromosome": tc / 30 * 1e6,
                  "what": "qk_evaluate_batch, statistics only, Gaussian noise, circuits already lowered"}))
