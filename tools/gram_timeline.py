#!/usr/bin/env python3
"""Per-workgroup timeline of one Gram launch from the timeline build (tools/gram_ablate.sh 128).
usage: python tools/gram_timeline.py --rows 4096 --qubits 8 [--out gpurun_out/timeline.npy]"""
import argparse, ctypes as C, json, os, sys
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=4096)
ap.add_argument("--qubits", type=int, default=8)
ap.add_argument("--out", default="")
a = ap.parse_args()
import numpy as np, torch
from quake_amd import _lib
_lib.LIB_PATH = Path(ROOT) / "tools" / "_abl" / "libquake_hip_abl128.so"
from quake_amd import device as qdev, featuremaps as fm
from quake_amd.circuit import ParameterVector
lib = _lib.load()
lib.qk_gram_set_trace.argtypes = [C.c_void_p]
lib.qk_gram_set_trace.restype = None
N, n = a.rows, a.qubits
X = torch.from_numpy(np.random.default_rng(1005).uniform(-1.5, 1.5, size=(N, n))).cuda()
psi = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 2)).simulate(X)
K = torch.empty((N, N), dtype=torch.float64, device="cuda")
T = N // 64
blocks_max = (T * (T + 1) // 2 // 512 + 2) * 512
trace = torch.zeros((blocks_max, 8), dtype=torch.float64, device="cuda")
for r in range(4):   # warm, then the traced launch is the last one
    trace.zero_()
    lib.qk_gram_set_trace(C.c_void_p(trace.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); qdev.gram(psi, None, out=K); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
t = trace.cpu().numpy()
t = np.concatenate([t, np.arange(len(t), dtype=np.float64)[:, None]], axis=1)    # column 8: the workgroup's block index
t = t[t[:, 0] > 0]
if a.out:
    np.save(a.out, t)
t0 = t[:, 0].min()
us = (t[:, :4] - t0) / 100.0     # 100 MHz -> microseconds
hw = t[:, 4].astype(np.int64); xcc = t[:, 5].astype(np.int64) & 15
cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
slot = xcc * 1000 + se * 100 + sh * 10 + cu
dur = us[:, 3] - us[:, 0]
out = {"rows": N, "qubits": n, "event_ms": ms, "workgroups": int(len(t)), "span_us": float(us[:, 3].max()),
       "distinct_cu": int(len(np.unique(slot))),
       "prologue_us_med": float(np.median(us[:, 1] - us[:, 0])), "prologue_us_p90": float(np.percentile(us[:, 1] - us[:, 0], 90)),
       "loop_us_med": float(np.median(us[:, 2] - us[:, 1])), "loop_us_p10": float(np.percentile(us[:, 2] - us[:, 1], 10)),
       "loop_us_p90": float(np.percentile(us[:, 2] - us[:, 1], 90)),
       "epilogue_us_med": float(np.median(us[:, 3] - us[:, 2])), "epilogue_us_p90": float(np.percentile(us[:, 3] - us[:, 2], 90)),
       "tile_us_med": float(np.median(dur))}
if t[:, 6].min() > 1e6:   # ring kernel: two more stamps inside the prologue
    out["locate_us_med"] = float(np.median(t[:, 6] - t[:, 0]) / 100.0)
    out["aim_and_issue_us_med"] = float(np.median(t[:, 7] - t[:, 6]) / 100.0)
    out["first_data_us_med"] = float(np.median(t[:, 1] - t[:, 7]) / 100.0)
# per CU: busy = union of workgroup intervals; gaps between a workgroup's end and the next start on the same CU
gaps = []
per_cu_tiles = []
for s in np.unique(slot):
    m = slot == s
    per_cu_tiles.append(int(m.sum()))
    st = np.sort(us[m, 0]); en = np.sort(us[m, 3])
    # with two slots per CU: the k-th start (k >= 2) follows the (k-2)-th end
    for k in range(2, len(st)):
        gaps.append(st[k] - en[k - 2])
out["tiles_per_cu_min_med_max"] = [int(np.min(per_cu_tiles)), float(np.median(per_cu_tiles)), int(np.max(per_cu_tiles))]
out["restart_gap_us_med"] = float(np.median(gaps)) if gaps else None
out["restart_gap_us_p90"] = float(np.percentile(gaps, 90)) if gaps else None
out["first_start_spread_us"] = float(np.percentile(us[:, 0], 20))
order = np.argsort(us[:, 0])
out["start_us_of_workgroup_512"] = float(us[order[min(512, len(order) - 1)], 0])
out["last_start_us"] = float(us[:, 0].max())
print(json.dumps(out), flush=True)
