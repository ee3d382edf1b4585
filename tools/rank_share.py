#!/usr/bin/env python3
"""What ONE rank of a G-rank run costs, measured on one GPU.

For G in {2, 4, 8} and rank in {0, G/2} (the two shapes of the half block, sharded.py plan_tasks) this runs
that rank's whole job set at full size exactly as ShardedGram.compute builds it (one multi-job launch, the
transposed KT staging stores included), then the same jobs with KT = NULL (what the mirror stores cost), then
qk_block_to_host_async of the rank's [N/G, N] row block.  It also times the two ways of getting Psi onto every
rank -- simulate all N rows (`--psi replicate`) against simulate N/G rows (`--psi gather`, plus an all-gather
whose cost is ARITHMETIC here: bytes / 153 GB/s per xGMI link, one link per peer) -- for ZZ r = 1 and r = 2.

    python tools/rank_share.py [--rows 65536] [--qubits 16] [--out gpurun_out/rank_share.json]

No collective runs here (one GPU): everything that involves a second GPU is labelled a prediction.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

XGMI_LINK_GBS = 153.0     # per peer pair and direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s)


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()                      # warm-up (first launch of a shape pays table uploads)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--qubits", type=int, default=16)
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rank_share.json"))
    a = ap.parse_args()

    from quake_amd import _lib, device as qdev, featuremaps as fm, sharded
    from quake_amd.circuit import ParameterVector

    N, n = a.rows, a.qubits
    D = 1 << n
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1005)
    lo, hi = -np.pi / 2 + 1e-5, np.pi / 2 - 1e-5
    X = torch.from_numpy(rng.uniform(lo, hi, size=(N, n))).to(dev)
    psi = torch.empty((N, D), dtype=torch.complex128, device=dev)
    out = {"rows": N, "qubits": n, "note": "one MI355X; a rank's job set run alone on it. Nothing here crossed xGMI."}

    # ---- how Psi gets onto every rank: replicate (simulate all rows) vs gather (simulate N/G rows + all-gather)
    psi_plan = []
    for r in (1, 2):
        circ = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, r))
        full_ms = timed(lambda: circ.simulate(X, out=psi), 2)
        row = {"circuit": f"zz r={r}", "simulate_all_rows_ms": full_ms, "per_world": {}}
        for G in (2, 4, 8):
            if N % G:
                continue
            own = N // G
            own_ms = timed(lambda: circ.simulate(X[:own], out=psi[:own]), 2)
            recv_bytes = 16.0 * D * (N - own)
            # full mesh: (G-1) peers send their shard at the same time, each over its own link
            mesh_ms = 16.0 * D * own / (XGMI_LINK_GBS * 1e9) * 1e3
            ring_ms = recv_bytes / (XGMI_LINK_GBS * 1e9) * 1e3
            row["per_world"][str(G)] = {
                "simulate_own_rows_ms": own_ms, "allgather_bytes_received": recv_bytes,
                "allgather_ms_predicted_mesh": mesh_ms, "allgather_ms_predicted_ring": ring_ms,
                "gather_total_ms_predicted_mesh": own_ms + mesh_ms,
                "replicate_total_ms_measured": full_ms,
                "cheaper": "gather" if own_ms + mesh_ms < full_ms else "replicate",
                "cheaper_if_ring": "gather" if own_ms + ring_ms < full_ms else "replicate"}
        psi_plan.append(row)
    out["psi_on_every_rank"] = psi_plan

    # leave the headline circuit's statevectors in psi
    circ = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, 1))
    circ.simulate(X, out=psi)
    torch.cuda.synchronize()

    # ---- the single-GPU reference point
    K1 = torch.empty((N, N), dtype=torch.float64, device=dev)
    one_ms = timed(lambda: qdev.gram(psi, out=K1), a.reps)
    out["single_gpu_gram_ms"] = one_ms
    del K1
    torch.cuda.empty_cache()

    shares = []
    for G in [int(g) for g in a.worlds.split(",")]:
        if N % G:
            continue
        for rank in sorted({0, G // 2}):
            sg = sharded.ShardedGram(N, D, rank, G)
            sg.allocate(dev, torch)
            r0, r1 = sharded.row_range(rank, G, N)
            K = torch.empty((r1 - r0, N), dtype=torch.float64, device=dev)
            diag = [t for t in sg.tasks if t.symmetric]
            rest = [t for t in sg.tasks if not t.symmetric]

            def with_kt():
                qdev.gram_jobs(sg._jobs(psi, K, diag + rest, list(sg._stage["send"])), D)

            def without_kt():
                jobs = sg._jobs(psi, K, diag + rest, list(sg._stage["send"]))
                for j in jobs:
                    j.pop("KT", None)
                qdev.gram_jobs(jobs, D)

            kt_ms = timed(with_kt, a.reps)
            nokt_ms = timed(without_kt, a.reps)
            entries = sharded.computed_entries(rank, G, N)
            send_bytes = [8.0 * b.numel() for b in sg._stage["send"]]
            # K block -> host
            K_host = torch.from_numpy(_lib.host_empty((r1 - r0, N)))
            cp = sharded.HipHostCopier()

            def d2h():
                cp.fence()
                cp.copy(K, K_host)
                cp.finish()

            d2h()
            t0 = time.perf_counter()
            d2h()
            d2h_s = time.perf_counter() - t0
            shares.append({
                "world": G, "rank": rank, "jobs": [[t.a1 - t.a0, t.b1 - t.b0, bool(t.symmetric)] for t in diag + rest],
                "entries_computed": entries, "share_of_triangle": entries / (N * (N + 1) / 2),
                "gram_ms_with_KT": kt_ms, "gram_ms_without_KT": nokt_ms, "KT_store_cost_ms": kt_ms - nokt_ms,
                "ideal_ms_single_over_G": one_ms / G, "overhead_vs_ideal": kt_ms / (one_ms / G) - 1.0,
                "algorithmic_tflops": 8.0 * D * entries / (kt_ms * 1e-3) / 1e12,
                "mirror_bytes_sent": sum(send_bytes), "mirror_messages": len(send_bytes),
                "mirror_ms_predicted_largest_message_per_link": (max(send_bytes) if send_bytes else 0.0) / (XGMI_LINK_GBS * 1e9) * 1e3,
                "k_block_bytes": 8.0 * (r1 - r0) * N, "k_block_d2h_ms": d2h_s * 1e3,
                "k_block_d2h_GBps": 8.0 * (r1 - r0) * N / d2h_s / 1e9})
            print(json.dumps(shares[-1]), flush=True)
            del K, K_host, sg
            torch.cuda.empty_cache()
    out["rank_shares"] = shares
    # predicted step times (labelled as such): slowest measured rank share + its largest mirror message over one link
    pred = {}
    for G in sorted({s["world"] for s in shares}):
        rs = [s for s in shares if s["world"] == G]
        worst = max(rs, key=lambda s: s["gram_ms_with_KT"])
        sim = out["psi_on_every_rank"][0]["simulate_all_rows_ms"]
        pred[str(G)] = {"predicted_step_ms": sim + worst["gram_ms_with_KT"] + worst["mirror_ms_predicted_largest_message_per_link"],
                        "predicted_entries_per_s": N * N / ((sim + worst["gram_ms_with_KT"] + worst["mirror_ms_predicted_largest_message_per_link"]) * 1e-3),
                        "basis": "measured simulate(all rows) + measured slowest rank share (KT stores included) + "
                                 "PREDICTED mirror hand-over (largest message / 153 GB/s; the messages of a rank go to different peers)"}
    out["predicted_steps"] = pred
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
