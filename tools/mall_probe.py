#!/usr/bin/env python3
"""Does the Infinity Cache absorb a write -> read -> overwrite of a slice?  4 GiB buffer swept slice by slice:
fill_ (write) then mul_ (read + write in place) per slice; whole-buffer passes for comparison."""
import torch, json
tot = 4 << 30
buf = torch.empty(tot // 8, dtype=torch.float64, device="cuda")
def run(slice_bytes):
    n = slice_bytes // 8
    views = [buf[i:i + n] for i in range(0, buf.numel(), n)]
    def sweep():
        for v in views:
            v.fill_(1.0)
            v.mul_(1.0000001)
    sweep(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): sweep()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3
for mb in (4096, 1024, 512, 256, 192, 128, 96, 64, 32):
    ms = run(mb << 20)
    print(json.dumps({"slice_MiB": mb, "ms_per_4GiB_sweep": round(ms, 3), "launches": 2 * (4096 // mb)}), flush=True)
# launch overhead alone: the same number of launches on tiny views
for mb in (64, 32):
    n = 4096 // mb
    tiny = buf[:1024]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        tiny.fill_(1.0); tiny.mul_(1.0000001)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"launches": 2 * n, "empty_ms": round(e0.elapsed_time(e1), 3)}), flush=True)
