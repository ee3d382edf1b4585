#!/usr/bin/env python3
"""Device-resident timings of the other BASELINE.json configs (2: N=4096 n=8 zz r=2; 3: 32 genomes,
N=16384, Q=10; 4: projected kernel N=32768 n=12).  Inputs are generated as SURVEY.md 8d says."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import device as qdev, featuremaps as fm, genetic
from quake_amd.circuit import ParameterVector

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def stage_times(circ, X, projected=False, gamma=1.0):
    N = X.shape[0]
    psi = torch.empty((N, circ.dim), dtype=torch.complex128, device="cuda")
    K = torch.empty((N, N), dtype=torch.float64, device="cuda")
    if projected:
        # product path: fused simulate + RDM (no Psi in HBM), then the pairwise kernel
        rd = qdev.simulate_rdm1(circ, X)
        out = {"simulate_rdm1_ms": timed(lambda: qdev.simulate_rdm1(circ, X))}
        out["pqk_ms"] = timed(lambda: qdev.pqk(rd, None, gamma, out=K))
        out["pqk_GBps_written"] = N * N * 8 / out["pqk_ms"] / 1e6
        out["total_ms"] = out["simulate_rdm1_ms"] + out["pqk_ms"]
        out["entries_per_s"] = N * N / out["total_ms"] * 1e3
        # the unfused stages, for reference
        out["unfused"] = {"simulate_ms": timed(lambda: circ.simulate(X, out=psi)),
                          "rdm1_ms": timed(lambda: qdev.rdm1(psi, circ.n_qubits))}
        out["unfused"]["rdm1_GBps"] = N * circ.dim * 16 / out["unfused"]["rdm1_ms"] / 1e6
        return out
    out = {"simulate_ms": timed(lambda: circ.simulate(X, out=psi))}
    out["gram_ms"] = timed(lambda: qdev.gram(psi, out=K))
    out["gram_TFLOPs_alg"] = 8.0 * circ.dim * N * (N + 1) / 2 / out["gram_ms"] / 1e9
    out["total_ms"] = sum(v for k, v in out.items() if k.endswith("_ms"))
    out["entries_per_s"] = N * N / out["total_ms"] * 1e3
    return out

ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="*", type=int, default=[2, 3, 4])
a = ap.parse_args()
lo, hi = -np.pi / 2 + 1e-5, np.pi / 2 - 1e-5
if 2 in a.configs:
    N, n = 4096, 8
    X = torch.from_numpy(np.random.default_rng(1002).uniform(lo, hi, size=(N, n))).cuda()
    c = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 2))
    print(json.dumps({"config": 2, "N": N, "n": n, **stage_times(c, X)}), flush=True)
    # the Qiskit library default (every pair entangled), reported separately as SURVEY 8d asks
    c = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 2, entanglement="full"))
    print(json.dumps({"config": "2 (entanglement=full)", "N": N, "n": n, **stage_times(c, X)}), flush=True)
if 3 in a.configs:
    N, Q, F, G = 16384, 10, 18, 8
    gd = {"single_non_parametric": ["Id", "H", "X", "SX"], "single_parametric": ["RX", "RY", "RZ"],
          "two_non_parametric": ["CX"], "two_parametric": []}
    rng = np.random.default_rng(1003)
    Xall = rng.uniform(1e-6, 1 - 1e-6, size=(N, F))
    pop = genetic.initial_population(32, Q, G, gd, F, rng=rng)
    cmap = [[q, q + 1] for q in range(Q - 1)] + [[q + 1, q] for q in range(Q - 1)]
    tot = {"simulate_ms": 0.0, "gram_ms": 0.0}
    t0 = time.perf_counter()
    for genes in pop:
        fmap, xi = genetic.to_quantum(genes, gd, F, G, Q, coupling_map=cmap, variant="v5", physical_qubits=list(range(Q)), verbose=False)
        if fmap.num_parameters != len(xi): continue
        c = qdev.DeviceCircuit(fmap)
        X = torch.from_numpy(np.ascontiguousarray(Xall[:, xi])).cuda()
        r = stage_times(c, X)
        tot["simulate_ms"] += r["simulate_ms"]; tot["gram_ms"] += r["gram_ms"]
    print(json.dumps({"config": 3, "N": N, "Q": Q, "genomes": len(pop), **tot,
                      "entries_per_s": len(pop) * N * N / (tot["simulate_ms"] + tot["gram_ms"]) * 1e3,
                      "gram_TFLOPs_alg": len(pop) * 8.0 * 1024 * N * (N + 1) / 2 / tot["gram_ms"] / 1e9}), flush=True)
if 4 in a.configs:
    N, n = 32768, 12
    X = torch.from_numpy(np.random.default_rng(1004).uniform(lo, hi, size=(N, n))).cuda()
    c = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 1))
    print(json.dumps({"config": 4, "N": N, "n": n, **stage_times(c, X, projected=True)}), flush=True)
