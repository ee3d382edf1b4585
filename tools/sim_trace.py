#!/usr/bin/env python3
"""Where the LDS simulator spends its time, per phase (phase-timing build: tools/sim_trace.sh).
usage: python tools/sim_trace.py --qubits 12 --maps zz_r2 genome_g8 [--samples 16384]"""
import argparse, ctypes as C, json, os, sys
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--qubits", type=int, nargs="+", default=[12])
ap.add_argument("--samples", type=int, default=16384)
ap.add_argument("--maps", nargs="*", default=["zz_r1", "zz_r2"])
a = ap.parse_args()
os.environ["QK_SIM_PATH"] = "lds"
import numpy as np, torch
from quake_amd import _lib
_lib.LIB_PATH = Path(ROOT) / "tools" / "_abl" / "libquake_hip_simtrace.so"
from quake_amd import device as qdev, featuremaps as fm, genetic
from quake_amd.circuit import ParameterVector
lib = _lib.load()
lib.qk_sim_set_trace.argtypes = [C.c_void_p]
lib.qk_sim_set_trace.restype = None
names = ["load+angles+tables", "product state", "diag tables", "diag sweep", "u1 sweeps", "cx/ecr", "store+barrier"]
for n in a.qubits:
    x = ParameterVector("x", n)
    circs = {"zz_r1": fm.zz_featuremap(x, n, 1), "zz_r2": fm.zz_featuremap(x, n, 2)}
    gd = {"single_non_parametric": ["Id", "H", "X", "SX"], "single_parametric": ["RX", "RY", "RZ"],
          "two_non_parametric": ["CX"], "two_parametric": []}
    genes = genetic.initial_population(1, n, 8, gd, n, rng=np.random.default_rng(7))[0]
    g, _ = genetic.to_quantum(genes, gd, n, 8, n, physical_qubits=list(range(n)), verbose=False)
    if g.num_parameters == n:
        circs["genome_g8"] = g
    N = a.samples
    X = torch.from_numpy(np.random.default_rng(1).uniform(-1.5, 1.5, size=(N, n))).cuda()
    psi = torch.empty((N, 1 << n), dtype=torch.complex128, device="cuda")
    for m in a.maps:
        if m not in circs:
            continue
        dc = qdev.DeviceCircuit(circs[m])
        trace = torch.zeros((8192, 16), dtype=torch.float64, device="cuda")
        lib.qk_sim_set_trace(C.c_void_p(trace.data_ptr()))
        dc.simulate(X, out=psi); torch.cuda.synchronize()
        trace.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dc.simulate(X, out=psi); e1.record(); torch.cuda.synchronize()
        t = trace.cpu().numpy()
        t = t[t[:, 7] > 0]
        items = t[:, 7].sum()
        per_item_us = t[:, :7].sum(0) / items / 100.0      # 100 MHz ticks -> us, per (sample, chunk) item of the LAST pass traced
        print(json.dumps({"qubits": n, "map": m, "ms": round(e0.elapsed_time(e1), 3), "passes": dc.handle.num_passes,
                          "workgroups": int(len(t)), "items": int(items), "plan": dc.handle.plan_summary,
                          "us_per_item": {k: round(float(v), 3) for k, v in zip(names, per_item_us)},
                          "us_per_item_total": round(float(per_item_us.sum()), 3),
                          "of_slot0_until_angle_programs_start_us": round(float(t[:, 11].sum() / items / 100.0), 3),
                          "of_slot0_until_own_angle_done_us": round(float(t[:, 8].sum() / items / 100.0), 3),
                          "diag_tables_split_us(stage record, tables, barrier)": [round(float(x), 3) for x in t[:, 12:15].sum(0) / items / 100.0],
                          "workgroup_loop_us_med": round(float(np.median(t[:, 9]) / 100.0), 1),
                          "workgroup_start_spread_us": round(float((t[:, 10].max() - t[:, 10].min()) / 100.0), 1)}), flush=True)
