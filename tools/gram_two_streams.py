#!/usr/bin/env python3
"""Two Gram launches in flight on one device: a large cross job on one stream while a second
stream keeps issuing small symmetric jobs -- each of which clears its pacing counters before it starts.  With one
shared counter slice (QK_GRAM_PACE_SHARED=1, the behaviour before round 3) the small launches wipe the large one's
counters; with a slice per launch they cannot.  Prints the time of the large job alone, of the small jobs alone, and
of both together; run once per setting (the switch is read once per process)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd import device as qdev, featuremaps as fm
from quake_amd.circuit import ParameterVector

n = 16; D = 1 << n; N = 16384
X = torch.from_numpy(np.random.default_rng(1).uniform(-1.5, 1.5, size=(N, n))).cuda()
psi = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 1)).simulate(X)
K = torch.empty((N // 2, N // 2), dtype=torch.float64, device="cuda")
Ks = torch.empty((1024, 1024), dtype=torch.float64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
SMALL = 24

def big():
    with torch.cuda.stream(s1):
        qdev.gram_jobs([dict(A=psi[:N // 2], B=psi[N // 2:], K=K, symmetric=False)], D)
def small():
    with torch.cuda.stream(s2):
        for k in range(SMALL):
            qdev.gram_jobs([dict(A=psi[:1024], B=psi[:1024], K=Ks, symmetric=True)], D)
def wall(fns):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in (s1, s2): s.wait_event(e0)
    for f in fns: f()
    for s in (s1, s2): torch.cuda.current_stream().wait_stream(s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
big(); small(); torch.cuda.synchronize()
out = {"shared_counters": os.environ.get("QK_GRAM_PACE_SHARED", "0"),
       "big_alone_ms": [round(wall([big]), 2) for _ in range(3)],
       "small_alone_ms": [round(wall([small]), 2) for _ in range(3)],
       "together_ms": [round(wall([big, small]), 2) for _ in range(5)]}
print(json.dumps(out))
