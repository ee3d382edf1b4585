#!/usr/bin/env python3
"""Headline benchmark: Gram-kernel entries/s for the 16-qubit, N=65536 train x train
fidelity kernel (BASELINE.json metric; configs[4], which fits one MI355X: Psi 64 GiB +
K 32 GiB of 288 GB).

One step = the whole hot path over one batch of synthetic inputs that already sit in
HBM: simulate the statevectors (N>1: every rank simulates all rows itself -- ~30 ms, cheaper
than receiving them over xGMI; --psi gather shards the simulation and all-gathers Psi over
RCCL instead) -> FP64-MFMA Gram of the rank's share of the upper triangle -> (N>1: hand the
mirrored blocks to the ranks that own those rows, point to point).
N GPUs split the SAME N=65536 problem (strong scaling); `value` is delivered entries
(N^2, both triangles written) per second of max-over-ranks wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--qubits Q]

For N>1 launch with torch.distributed.run (one rank per GPU).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 (f32 in, f32 acc)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense FP64 matrix peak (v_mfma_f64_16x16x4_f64: 256 CU x 4 SIMD x 2048 flop / 64 clk x 2.4 GHz)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=65536, help="N (default: the north-star size)")
    ap.add_argument("--qubits", type=int, default=16)
    ap.add_argument("--repeats", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="skip the 1-second MFMA ceiling probe")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU-baseline sample (0 = auto)")
    ap.add_argument("--check", action="store_true", help="spot-check K against the oracle after the run")
    ap.add_argument("--precision", choices=["double", "single"], default="double",
                    help="single = complex64 statevectors + FP32-MFMA Gram stage (optional mode, 1e-5 tolerance)")
    ap.add_argument("--workload", choices=["fidelity", "pqk"], default="fidelity",
                    help="pqk = BASELINE configs[3]: projected kernel, N=32768, 12 qubits (use --rows/--qubits to change)")
    ap.add_argument("--psi", choices=["replicate", "gather"], default="replicate",
                    help="N>1: every rank simulates all rows (default, no collective on the data path), or "
                         "simulates its own rows and all-gathers the statevector shards over RCCL")
    ap.add_argument("--test-rows", type=int, default=-1,
                    help="also time the test x train block of configs[4] after the timed steps (default: N/4 at N=1, 0 = off)")
    return ap.parse_args()


def cpu_baseline(X_host, n, repeats, rows):
    """The oracle's C twin (oracle/qk_oracle.c) on one host core: gate-by-gate statevector
    per sample, then one |vdot|^2 per pair (j >= i, mirrored) -- the algorithm
    FidelityStatevectorKernel runs, minus its Python overhead and its O(N^3) eig."""
    from oracle import qk_oracle as O
    from oracle import qk_oracle_c as OC

    OC.load()
    ops = O.zz_featuremap(n, repeats)
    Xs = np.ascontiguousarray(X_host[:rows])
    t0 = time.perf_counter()
    psi = OC.simulate(ops, n, Xs)
    t1 = time.perf_counter()
    K = OC.fidelity(psi)
    t2 = time.perf_counter()
    out = {
        "value": rows * rows / (t2 - t0),
        "unit": "entries/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {rows} rows of the same X, n={n} zz_featuremap r={repeats}, train x train; "
                  f"simulate {t1 - t0:.2f}s + pair loop {t2 - t1:.2f}s "
                  f"(pair loop alone: {rows * rows / (t2 - t1):.3e} entries/s, the large-N limit)",
        "host": f"{os.cpu_count()} logical CPUs visible; 1 thread used",
    }
    # SURVEY 8d (ii) "best CPU": the same contraction as one multithreaded zgemm on the host cores
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
        big = max(rows, min(2048, (1 << 31) // (16 << n)))          # at most 2 GiB of Psi
        P = np.ascontiguousarray(np.tile(psi, (-(-big // rows), 1))[:big])
        t3 = time.perf_counter()
        G = P.conj() @ P.T
        Kb = G.real ** 2 + G.imag ** 2
        t4 = time.perf_counter()
        out["best_cpu"] = {"value": big * big / (t4 - t3), "unit": "entries/s", "cores": threads,
                           "kind": "numpy zgemm + square (all N^2 entries, Psi given)",
                           "sample": f"{big} rows (the {rows} simulated rows tiled), {t4 - t3:.2f}s"}
        del P, G, Kb
    except Exception as e:   # the reported baseline is the port above; this leg is extra
        out["best_cpu"] = {"error": repr(e)}
    return out, K, Xs


def mfma_probe(seconds=1.0):
    """The box's FP64 MFMA ceiling right now: tools/mfma_f64_peak.bin (built by build()) keeps a
    register-only v_mfma_f64_16x16x4 loop running for `seconds` in a child process.  Boxes of this
    pool sustain different clocks under matrix load (2.1 - 2.4 GHz seen), so the spec-sheet peak
    alone does not say how close the kernel is to what this GPU can do."""
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "mfma_f64_peak.bin")
    if not os.path.exists(exe):
        return None
    try:
        import subprocess
        r = subprocess.run([exe, "--probe", str(seconds)], capture_output=True, text=True, timeout=60)
        return json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
    except Exception:
        return None


def gram_kernel_name(rows, D):
    """Which gram_dma_kernel instantiation a block of `rows` rows runs on (the library picks per call)."""
    import ctypes as C
    from quake_amd import _lib
    lib = _lib.load()
    lib.qk_gram_variant_for.restype = C.c_char_p
    lib.qk_gram_variant_for.argtypes = [C.c_int64, C.c_int64]
    name = lib.qk_gram_variant_for(rows, D).decode()
    return {"dma3m_128x128_w8": "qk::gram_dma_kernel<2,4,4,2,true> (128x128 tiles)",
            "dma3m_64x64_w4": "qk::gram_dma_kernel<2,2,2,2,true> (64x64 tiles)",
            "dma3m_64x64_w4_g64": "qk::gram_dma_kernel<2,2,2,2,true> (64x64 tiles, 8x8 tile groups)"}.get(name, name)


def pmc_traffic(N, n, repeats, world):
    """HBM bytes per Gram launch from the committed PMC passes (profiles/), when this is the
    workload they were taken on; collected per the guide (separate --pmc passes, FETCH_SIZE x2)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    try:
        t = json.load(open(path))
    except Exception:
        return None
    w = t.get("workload", {})
    if (w.get("rows"), w.get("qubits"), w.get("repeats"), w.get("gpus")) != (N, n, repeats, world):
        return None
    return t.get("gram_hbm_bytes_per_launch")


def pqk_main(a, rank, world, dev, dist):
    """configs[3]: projected quantum kernel.  Each rank simulates its row shard straight into the RDM
    table (fused, no statevectors in HBM), ONE all-gather of that table (N x n x 4 float64), then the
    rank's [N/G, N] block of exp(-gamma sum_q ||rho_i - rho_j||_F).  One JSON line, same contract."""
    from quake_amd import device as qdev, featuremaps as fm, sharded
    from quake_amd.circuit import ParameterVector
    N = a.rows if a.rows != 65536 else 32768
    n = a.qubits if a.qubits != 16 else 12
    sharded.check_shape(N, world)
    r0, r1 = sharded.row_range(rank, world, N)
    rng = np.random.default_rng(1004)
    X = torch.from_numpy(rng.uniform(-np.pi / 2 + 1e-5, np.pi / 2 - 1e-5, size=(N, n))[r0:r1].copy()).to(dev)
    circ = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, a.repeats))
    ev = []

    def step(timed):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        rdm = qdev.simulate_rdm1(circ, X)
        e[1].record()
        K = sharded.sharded_projected_kernel(rdm, 1.0, rank, world, host_staging=os.environ.get("QK_BENCH_REHEARSAL") == "1")
        e[2].record()
        if timed:
            ev.append(e)
        return K

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    K = None
    for _ in range(a.steps):
        K = step(True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if world > 1 and K is not None and os.environ.get("QK_BENCH_REHEARSAL") == "1":
        # the rank's block against one computed from a table this rank built alone
        Xf = torch.from_numpy(np.random.default_rng(1004).uniform(-np.pi / 2 + 1e-5, np.pi / 2 - 1e-5, size=(N, n))).to(dev)
        full = qdev.simulate_rdm1(circ, Xf)
        d = (qdev.pqk(full[r0:r1].contiguous(), full, 1.0) - K).abs()
        d[torch.arange(r1 - r0, device=dev), torch.arange(r0, r1, device=dev)] = 0.0
        print(f"[rehearsal] rank {rank}: rows {r0}:{r1}, max |K_block - direct block| = {float(d.max()):.3e}", flush=True)
        if not float(d.max()) < 1e-12:
            raise SystemExit(f"rank {rank}: sharded projected-kernel block differs from the direct block")
    if rank == 0:
        sim_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev])) if ev else 0.0
        pair_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) if ev else 0.0
        written = 8.0 * (r1 - r0) * N
        print(json.dumps({
            "metric": "projected-kernel entries/sec (BASELINE configs[3])", "value": N * N * a.steps / dt, "unit": "entries/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / max(a.steps, 1) * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n}-qubit zz_featuremap r={a.repeats} projected kernel (1-RDM + Gaussian), N={N}",
                       "rows": N, "qubits": n, "parallelism": f"row-block x{world}"},
            "roofline": {"bound": "hbm", "achieved": written / (pair_ms * 1e-3) / 1e9 if pair_ms else 0.0, "peak": 8000.0,
                         "unit": "GB/s", "frac": written / (pair_ms * 1e-3) / 1e9 / 8000.0 if pair_ms else 0.0, "traffic": None,
                         "kernel": "qk::pqk_kernel", "note": "8 bytes written per entry of rank 0's block / pairwise-kernel time "
                         "(includes the RDM all-gather at N>1); the kernel is bound by FP64 VALU issue (12 operations + one rsq per (pair, qubit); PMC: 80 % VALU busy), not by these bytes"},
            "stages_ms": {"simulate_rdm1": sim_ms, "pairwise": pair_ms}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    probe = mfma_probe() if (world == 1 and not a.no_probe) else None   # child process, before this one touches the GPU
    # QK_BENCH_REHEARSAL=1: rehearse the N>1 code path on a ONE-GPU box -- every rank uses cuda:0 and the
    # exchanges go over gloo (RCCL refuses two ranks on one device).  Timings of such a run mean nothing.
    rehearsal = world > 1 and os.environ.get("QK_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from quake_amd import device as qdev, featuremaps as fm, sharded
    from quake_amd.circuit import ParameterVector

    if a.workload == "pqk":
        return pqk_main(a, rank, world, dev, dist)
    N, n = a.rows, a.qubits
    D = 1 << n
    sharded.check_shape(N, world)
    r0, r1 = sharded.row_range(rank, world, N)

    # synthetic inputs of SURVEY.md 8d (config 5): range produced by rearrange_scale
    rng = np.random.default_rng(1005)
    lo, hi = -np.pi / 2 + 1e-5, np.pi / 2 - 1e-5
    X_host = rng.uniform(lo, hi, size=(N, n))
    # N > 1: every rank simulates all N rows (deterministic, ~30 ms) instead of receiving 7/8 of Psi
    # over xGMI; --psi gather simulates the rank's rows only and all-gathers the shards over RCCL
    gather = world > 1 and a.psi == "gather"
    X = torch.from_numpy(X_host[r0:r1].copy() if gather else X_host).to(dev)

    circ = qdev.DeviceCircuit(fm.zz_featuremap(ParameterVector("x", n), n, a.repeats))
    single = a.precision == "single"
    psi = torch.empty((N, D), dtype=torch.complex64 if single else torch.complex128, device=dev)
    psi_real = torch.view_as_real(psi).view(N, 2 * D)
    K = torch.empty((r1 - r0, N), dtype=torch.float64, device=dev)
    sg = sharded.ShardedGram(N, D, rank, world, host_staging=rehearsal)
    sg.allocate(dev, torch)

    ev = {}

    def timer(tag):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.setdefault(tag, []).append(e)

    def step(timed):
        t = timer if timed else None
        if t: t("sim_begin")
        circ.simulate(X, out=psi[r0:r1] if gather else psi)
        if t: t("sim_end")
        work = None
        if gather:
            # in-place all-gather of the shards (float64 views: RCCL has no complex dtype)
            work = dist.all_gather_into_tensor(psi_real, psi_real[r0:r1], async_op=True)
        sg.compute(psi, K, gather_work=work, timer=t)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rehearsal:
        # the rank's delivered row block against a plain cross block of the same rows (no symmetry, no exchange)
        Kc = qdev.gram(psi[r0:r1], psi)
        d = (Kc - K).abs()
        d[torch.arange(r1 - r0, device=dev), torch.arange(r0, r1, device=dev)] = 0.0   # the diagonal is pinned to exactly 1
        err = float(d.max())
        print(f"[rehearsal] rank {rank}: rows {r0}:{r1}, max |K_block - cross block| = {err:.3e}", flush=True)
        if not err < (1e-5 if single else 1e-12):
            raise SystemExit(f"rank {rank}: sharded row block differs from the direct cross block")
        del Kc, d

    # configs[4] also names 16384 test rows: the test x train cross block, outside `value`
    test_block = None
    T = a.test_rows if a.test_rows >= 0 else (N // 4 if world == 1 else 0)
    if T > 0 and world == 1:
        Xt = torch.from_numpy(rng.uniform(lo, hi, size=(T, n))).to(dev)
        psi_t = torch.empty((T, D), dtype=psi.dtype, device=dev)
        Kt = torch.empty((T, N), dtype=torch.float64, device=dev)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        circ.simulate(Xt, out=psi_t)
        e[1].record()
        qdev.gram(psi_t, psi, out=Kt)
        e[2].record()
        torch.cuda.synchronize()
        t_ms = e[1].elapsed_time(e[2])
        test_block = {"rows": T, "simulate_ms": e[0].elapsed_time(e[1]), "gram_ms": t_ms,
                      "entries_per_s": T * N / ((e[0].elapsed_time(e[2])) * 1e-3),
                      "gram_tflops_algorithmic": 8.0 * D * T * N / (t_ms * 1e-3) / 1e12}
        del psi_t, Kt, Xt

    gram_ms = float(np.mean([b.elapsed_time(e) for b, e in zip(ev["gram_begin"], ev["gram_end"])])) if a.steps else 0.0
    sim_ms = float(np.mean([b.elapsed_time(e) for b, e in zip(ev["sim_begin"], ev["sim_end"])])) if a.steps else 0.0
    entries_rank = sharded.computed_entries(rank, world, N)
    flops_rank = 8.0 * D * entries_rank
    achieved = flops_rank / (gram_ms * 1e-3) / 1e12 if gram_ms > 0 else 0.0

    PEAK = FP32_MFMA_PEAK_TFLOPS if single else FP64_MFMA_PEAK_TFLOPS
    if rank == 0:
        out = {
            "metric": "Gram kernel entries/sec (whole node) at 16 qubits, N=65536; % FP64 MFMA peak",
            "value": N * N * a.steps / dt,
            "unit": "entries/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / max(a.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if single else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{n}-qubit zz_featuremap r={a.repeats} fidelity kernel, N={N} train x train "
                            f"(BASELINE configs[4]), symmetric: N(N+1)/2 entries computed, N^2 delivered",
                "rows": N, "qubits": n, "parallelism": f"row-block x{world}",
                "psi": "single GPU" if world == 1 else ("all-gather of shards" if gather else "simulated on every rank"),
            },
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": PEAK, "unit": "TFLOP/s",
                "frac": achieved / PEAK, "traffic": None if single else pmc_traffic(N, n, a.repeats, world),
                "kernel": "qk::gram32_dma_kernel<2,2,2,2>" if single else gram_kernel_name(N // world, D),
                "ms_per_launch_set": gram_ms,
                "algorithmic_flops": flops_rank,
                "mfma_flops": 0.75 * flops_rank,
                "mfma_frac": 0.75 * achieved / PEAK,
                "note": "achieved = 8*D algorithmic real flops per computed entry (rank 0's share) / kernel time, "
                        "per-GPU peak. The kernel uses the 3-multiplication (Karatsuba) complex product, so the "
                        "matrix cores execute 6*D flops per entry: mfma_frac is the fraction of MFMA issue slots "
                        "actually busy; frac > 1 means fewer hardware flops than the algorithmic count, not a "
                        "faster-than-peak pipe",
            },
            "stages_ms": {"simulate": sim_ms, "gram": gram_ms},
        }
        if test_block:
            out["test_block"] = test_block
        if probe and not single:
            out["roofline"]["mfma_probe_tflops"] = probe["tflops"]
            out["roofline"]["mfma_frac_of_probe"] = 0.75 * achieved / probe["tflops"]
            out["roofline"]["note"] += ("; mfma_probe_tflops = a register-only MFMA loop timed on this box just before "
                                        "the run (its ceiling at the clock it sustains), mfma_frac_of_probe = hardware "
                                        "MFMA flops / that")
        if world == 1 and not a.no_cpu_baseline:
            rows = a.cpu_rows or max(32, min(384, int(384 * (65536 / D) ** 0.5)))
            rows = min(rows, N)
            base, K_cpu, _ = cpu_baseline(X_host, n, a.repeats, rows)
            out["cpu_baseline"] = base
            if a.check:
                err = float(np.abs(K[:rows, :rows].cpu().numpy() - K_cpu).max())
                out["check_max_abs_err_vs_oracle"] = err
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
