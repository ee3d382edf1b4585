#!/usr/bin/env python3
"""Time the batched statevector simulator: algorithmic bytes = 16 * 2^n per sample (one write of Psi)."""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import _lib
if os.environ.get("QK_OLD"):      # same-box A/B against another build of the library, copied next to it by hand: QK_OLD=1 -> libquake_hip_before.so, QK_OLD=name -> libquake_hip_name.so
    from pathlib import Path
    _lib.LIB_PATH = Path(_lib.LIB_PATH).with_name("libquake_hip_%s.so" % ("before" if os.environ["QK_OLD"] == "1" else os.environ["QK_OLD"]))
from quake_amd import device as qdev, featuremaps as fm
from quake_amd.circuit import ParameterVector, QuantumCircuit

def circuits(n):
    x = ParameterVector("x", n)
    h = QuantumCircuit(n)
    for q in range(n): h.h(q)
    hp = fm.z_featuremap(x, n, 1)
    def ph(qs, reps=4):       # p then h: the h is a real sweep, not a hoisted first gate
        c = QuantumCircuit(n)
        for _ in range(reps):
            for q in qs: c.p(x[q], q)
            for q in qs: c.h(q)
        return c
    extra = {}
    if n >= 9:
        extra = {"ph_low3x4": ph([0, 1, 2]), "ph_mid3x4": ph([3, 4, 5]), "ph_high3x4": ph([n - 4, n - 5, n - 6])}
    if n <= 13:      # a GA chromosome (v3 decoder, every gate of the alphabet): the config-3 circuit shape
        from quake_amd import genetic
        gd = {"single_non_parametric": ["Id", "H", "X", "SX"], "single_parametric": ["RX", "RY", "RZ"],
              "two_non_parametric": ["CX"], "two_parametric": []}
        genes = genetic.initial_population(1, n, 8, gd, n, rng=np.random.default_rng(7))[0]
        g, xi = genetic.to_quantum(genes, gd, n, 8, n, physical_qubits=list(range(n)), verbose=False)
        if g.num_parameters == n:
            extra["genome_g8"] = g
        # --genomes K: K more chromosomes per gate alphabet (the full one above and the hardware-native one of the
        # reference's examples), for rules that must hold for genomes in general and not for one of them
        hw = {"single_non_parametric": ["Id", "X", "SX"], "single_parametric": ["RZ", "RZ"], "two_non_parametric": ["ECR"], "two_parametric": []}
        for k in range(N_GENOMES):
            for tag, alphabet in (("full", gd), ("hw", hw)):
                genes = genetic.initial_population(1, n, 8, alphabet, n, rng=np.random.default_rng(100 + k))[0]
                g, xi = genetic.to_quantum(genes, alphabet, n, 8, n, physical_qubits=list(range(n)), verbose=False)
                if g.num_parameters == n:
                    extra["genome_%s%d" % (tag, k)] = g
    return {"h_only": h, "z_r1": hp, "z_r2": fm.z_featuremap(x, n, 2), "zz_r1": fm.zz_featuremap(x, n, 1), "zz_r2": fm.zz_featuremap(x, n, 2),
            "custom1_r1": fm.custom1_featuremap(x, n, 1), "custom2_r1": fm.custom2_featuremap(x, n, 1),
            "custom1_r2": fm.custom1_featuremap(x, n, 2), "custom2_r2": fm.custom2_featuremap(x, n, 2),
            "zz_full_r2": fm.zz_featuremap(x, n, 2, entanglement="full"), **extra}

ap = argparse.ArgumentParser()
ap.add_argument("--qubits", type=int, nargs="+", default=[16, 13, 12, 10, 8])
ap.add_argument("--gib", type=float, default=4.0, help="Psi bytes per run")
ap.add_argument("--maps", nargs="*", default=None)
ap.add_argument("--genomes", type=int, default=0, help="extra random chromosomes per gate alphabet (maps genome_full<k>, genome_hw<k>)")
ap.add_argument("--fill", action="store_true", help="also time a plain device fill / copy of the same size")
a = ap.parse_args()
N_GENOMES = a.genomes
if a.fill:
    buf = torch.empty(int(a.gib * 2**30) // 8, dtype=torch.float64, device="cuda")
    src = torch.ones_like(buf)
    for name, fn in (("fill", lambda: buf.fill_(1.0)), ("copy", lambda: buf.copy_(src))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(json.dumps({"op": name, "ms": round(ms, 3), "written_GBps": round(buf.numel() * 8 / ms / 1e6, 1)}), flush=True)
    del buf, src
for n in a.qubits:
    N = int(a.gib * 2**30 / (16 << n))
    X = torch.from_numpy(np.random.default_rng(1).uniform(-1.5, 1.5, size=(N, n))).cuda()
    psi = torch.empty((N, 1 << n), dtype=torch.complex128, device="cuda")
    for name, c in circuits(n).items():
        if a.maps and name not in a.maps and not (a.genomes and name.startswith("genome_") and "genome" in a.maps): continue
        dc = qdev.DeviceCircuit(c)
        Xn = X[:, :dc.n_params].contiguous() if dc.n_params else X[:, :0].contiguous()
        dc.simulate(Xn, out=psi); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): dc.simulate(Xn, out=psi)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(json.dumps({"n": n, "map": name, "N": N, "passes": dc.handle.num_passes, "ms": round(ms, 3),
                          "alg_GBps": round(N * (16 << n) / ms / 1e6, 1), "frac_of_8TBps": round(N * (16 << n) / ms / 1e6 / 8000, 4)}), flush=True)
