#!/usr/bin/env python3
"""Per-qubit reduced density matrices of a batch: the fused simulate + RDM mode of the LDS kernel (Psi never written)
against simulate (default path: register-resident kernel where it applies) + rdm1 over the written Psi."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import _lib, device as qdev, featuremaps as fm, genetic
from quake_amd.circuit import ParameterVector

def genome(n, seed=7):
    gd = {"single_non_parametric": ["Id", "H", "X", "SX"], "single_parametric": ["RX", "RY", "RZ"], "two_non_parametric": ["CX"], "two_parametric": []}
    genes = genetic.initial_population(1, n, 8, gd, n, rng=np.random.default_rng(seed))[0]
    return genetic.to_quantum(genes, gd, n, 8, n, physical_qubits=list(range(n)), verbose=False)[0]

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for n, N in ((13, 16384), (12, 32768), (11, 65536), (10, 131072), (8, 262144), (8, 4096)):
    x = ParameterVector("x", n)
    maps = {"zz_r1": fm.zz_featuremap(x, n, 1), "zz_r2": fm.zz_featuremap(x, n, 2), "custom1_r2": fm.custom1_featuremap(x, n, 2),
            "zz_full_r2": fm.zz_featuremap(x, n, 2, entanglement="full"), "genome": genome(n), "genome_b": genome(n, 11)}
    X = torch.from_numpy(np.random.default_rng(1).uniform(-1.0, 1.0, size=(N, n))).cuda()      # (arcsin in the genomes' angle programs)
    psi = torch.empty((N, 1 << n), dtype=torch.complex128, device="cuda")
    for name, c in maps.items():
        dc = qdev.DeviceCircuit(c)
        if dc.n_params > n: continue
        Xn = X[:, :dc.n_params].contiguous()
        auto = timed(lambda: qdev.simulate_rdm1(dc, Xn))
        lib = _lib.load()
        out_f = torch.empty((N, n, 4), dtype=torch.float64, device="cuda")
        def fused_call():      # no workspace: a circuit that can run fused does
            _lib.check(lib.qk_simulate_rdm1_dev(dc.handle.handle, Xn.data_ptr(), N, Xn.stride(0), out_f.data_ptr(), None, None))
        fused = timed(fused_call) if lib.qk_simulate_rdm1_is_fused(dc.handle.handle) else float("nan")
        def unf():
            dc.simulate(Xn, out=psi)
            return qdev.rdm1(psi, n)
        unfused = timed(unf)
        a, b = qdev.simulate_rdm1(dc, Xn), unf()
        print(json.dumps({"n": n, "N": N, "map": name, "fused_ms": round(fused, 3), "sim_plus_rdm_ms": round(unfused, 3), "library_choice_ms": round(auto, 3),
                          "wants_workspace": int(lib.qk_simulate_rdm1_wants_workspace(dc.handle.handle, N)),
                          "max_diff": float((a - b).abs().max())}), flush=True)
