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

`--gpus N` with N > 1 from a bare shell starts its own ranks: the parent process (which never
touches the GPU) runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child
and hands on its output and exit code.  Launched under torch.distributed.run (WORLD_SIZE set) it
is a rank.  After the timed steps, outside `value`: the test x train block of configs[4] (sharded
over the ranks), the K-block -> host leg (N>1) or the whole host entry point (N=1: numpy in,
page-locked numpy out, H2D -> simulate -> Gram -> D2H), and the CPU baseline (N=1).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 (f32 in, f32 acc)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense FP64 matrix peak (v_mfma_f64_16x16x4_f64: 256 CU x 4 SIMD x 2048 flop / 64 clk x 2.4 GHz)
MFMA_FLOPS_PER_ALGORITHMIC = 0.75   # 3-multiplication complex product: 6*D hardware flops per entry for 8*D algorithmic


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
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive legs after the timed steps")
    ap.add_argument("--host-calls", type=int, default=3, help="N=1: calls of the host entry point averaged into value_8d")
    ap.add_argument("--cpu-rows", type=int, default=0, help="largest row count of the CPU-baseline sweep (0 = auto)")
    ap.add_argument("--check", action="store_true", help="spot-check K against the oracle after the run")
    ap.add_argument("--precision", choices=["double", "single"], default="double",
                    help="single = complex64 statevectors + FP32-MFMA Gram stage (optional mode, 1e-5 tolerance)")
    ap.add_argument("--workload", choices=["fidelity", "pqk"], default="fidelity",
                    help="pqk = BASELINE configs[3]: projected kernel, N=32768, 12 qubits (use --rows/--qubits to change)")
    ap.add_argument("--pqk-map", choices=["zz", "genome"], default="zz",
                    help="pqk workload: the ZZ map, or a seeded 12-qubit genetic genome (configs[3] names both)")
    ap.add_argument("--psi", choices=["replicate", "gather"], default="replicate",
                    help="N>1: every rank simulates all rows (default, no collective on the data path), or "
                         "simulates its own rows and all-gathers the statevector shards over RCCL")
    ap.add_argument("--test-rows", type=int, default=-1,
                    help="also time the test x train block of configs[4] after the timed steps (default: N/4, 0 = off)")
    return ap.parse_args()


def self_launch(a) -> int:
    """`python bench.py --gpus N` from a bare shell: start the N ranks as a child job.  This process has
    not touched the GPU (importing torch does not) and never will; it does not exec."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = os.environ.copy()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] starting {a.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(X_host, n, repeats, rows_list):
    """The oracle's C twin (oracle/qk_oracle.c) on one host core: gate-by-gate statevector
    per sample, then one |vdot|^2 per pair (j >= i, mirrored) -- the algorithm
    FidelityStatevectorKernel runs, minus its Python overhead and its O(N^3) eig.  Timed at several
    row counts so that the N^2 growth of the pair loop (the large-N limit) can be read off."""
    from oracle import qk_oracle as O
    from oracle import qk_oracle_c as OC

    OC.load()
    ops = O.zz_featuremap(n, repeats)
    sweep, K, psi = [], None, None
    for rows in rows_list:
        Xs = np.ascontiguousarray(X_host[:rows])
        t0 = time.perf_counter()
        psi = OC.simulate(ops, n, Xs)
        t1 = time.perf_counter()
        K = OC.fidelity(psi)
        t2 = time.perf_counter()
        sweep.append({"rows": rows, "simulate_s": t1 - t0, "pair_loop_s": t2 - t1,
                      "entries_per_s": rows * rows / (t2 - t0), "pair_loop_entries_per_s": rows * rows / (t2 - t1)})
    rows = rows_list[-1]
    last = sweep[-1]
    out = {
        "value": last["entries_per_s"],
        "unit": "entries/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {rows} rows of the same X, n={n} zz_featuremap r={repeats}, train x train; "
                  f"simulate {last['simulate_s']:.2f}s + pair loop {last['pair_loop_s']:.2f}s "
                  f"(pair loop alone: {last['pair_loop_entries_per_s']:.3e} entries/s, the large-N limit: "
                  f"simulation grows with N, the pair loop with N^2)",
        "sweep": sweep,
        "host": f"{cpu_model()}; {os.cpu_count()} logical CPUs visible; 1 thread used",
        "note": "a lower bound on the reference's cost: Qiskit adds Python circuit binding per sample, a Python-level "
                "loop over the pairs and an O(N^3) eig",
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
    return out, K


def mfma_probe(seconds=1.0):
    """The box's FP64 MFMA ceiling right now: tools/mfma_f64_peak.bin (built by build()) keeps a
    register-only v_mfma_f64_16x16x4 loop running for `seconds` in a child process.  Boxes of this
    pool sustain different clocks under matrix load (2.1 - 2.4 GHz seen), so the spec-sheet peak
    alone does not say how close the kernel is to what this GPU can do."""
    exe = os.path.join(ROOT, "tools", "mfma_f64_peak.bin")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "--probe", str(seconds)], capture_output=True, text=True, timeout=60)
        return json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
    except Exception:
        return None


def gram_kernel_name(rows, D):
    """Which gram kernel instantiation a block of `rows` rows runs on (the library picks per call)."""
    import ctypes as C
    from quake_amd import _lib
    lib = _lib.load()
    lib.qk_gram_variant_for.restype = C.c_char_p
    lib.qk_gram_variant_for.argtypes = [C.c_int64, C.c_int64]
    name = lib.qk_gram_variant_for(rows, D).decode()
    return {"4m_128x128_w8_bk8": "qk::gram_kernel<2,4,4,2,8,false> (register-staged, circuits under 3 qubits)",
            "dma3m_64x64_w4_g64": "qk::gram_dma_kernel<2,2,2,2,true> (64x64 tiles, rows beyond 2^22 amplitudes)",
            "ring3_64x64_w4": "qk::gram_ring_kernel<3> (64x64 tiles, 3-stage LDS ring, 3 workgroups per CU)",
            "ring5_64x64_w4": "qk::gram_ring_kernel<5> (64x64 tiles, 5-stage LDS ring, 2 workgroups per CU)"}.get(name, name)


def pmc_traffic_committed(N, n, repeats, world):
    """HBM bytes per Gram launch from the committed PMC passes (profiles/), when this is the workload
    they were taken on.  NOT a measurement of this run (that is why `roofline.traffic` stays null);
    collected per the guide (separate --pmc passes, FETCH_SIZE x2)."""
    for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        w = t.get("workload", {})
        if (w.get("rows"), w.get("qubits"), w.get("repeats"), w.get("gpus")) == (N, n, repeats, world):
            return {"bytes_per_launch": t.get("gram_hbm_bytes_per_launch"), "source": f"profiles/{name}"}
    return None


def pqk_circuit(a, n):
    from quake_amd import featuremaps as fm, genetic
    from quake_amd.circuit import ParameterVector
    if a.pqk_map == "zz":
        return fm.zz_featuremap(ParameterVector("x", n), n, a.repeats), n, f"{n}-qubit zz_featuremap r={a.repeats}"
    # a seeded genome over the reference's gate alphabet (genetic_v3.initial_population's shape), decoded by
    # the v3 rule; configs[3] asks for "one genetic genome" next to the ZZ map
    gate_dict = {"single_non_parametric": ["Id", "H", "X", "SX"], "single_parametric": ["RX", "RY", "RZ"],
                 "two_non_parametric": ["CX"], "two_parametric": []}
    F, G = 18, 8
    pop = genetic.initial_population(1, n, G, gate_dict, F, rng=np.random.default_rng(1004))
    fmap, x_idxs = genetic.to_quantum(pop[0], gate_dict, F, G, n, physical_qubits=list(range(n)), verbose=False)
    return fmap, len(x_idxs), f"{n}-qubit genetic genome (seed 1004, G={G}, F={F}, {len(x_idxs)} features used)"


# FP64 vector-ALU issue on gfx950: one wave-instruction (64 lanes) per 4 cycles and SIMD (16 lanes a clock; the 78.6
# TFLOP/s of the vector FP64 peak is this rate x 2 flops x 1024 SIMDs x 2.4 GHz); a transcendental (v_rsq_f64) holds the
# pipe for 16 (profiles/r02_valu_rates.md: 16.7-18.7 ns-cycles measured at a nominal 2.4 GHz, FMA 4.9-6.5)
VALU_F64_CYCLES = 4.0
VALU_RSQ_F64_CYCLES = 16.0
PQK_OPS_PER_ROOT = 10          # three differences, three fused squares, the residual correction of the root, two accumulating FMAs
NOMINAL_GHZ = 2.4


def pqk_roofline(n_qubits, pairs, written_bytes, pair_ms):
    """The pairwise kernel's roof is FP64 vector issue, not its 8 bytes per entry: per (pair, qubit) it needs
    PQK_OPS_PER_ROOT FP64 operations and one v_rsq_f64 (the per-entry exponential, 14 FMAs, is
    counted too).  peak = roots/s the 1024 SIMDs could issue at NOMINAL_GHZ if nothing else ran."""
    roots = float(pairs) * n_qubits
    cycles_per_wave_root = PQK_OPS_PER_ROOT * VALU_F64_CYCLES + VALU_RSQ_F64_CYCLES
    exp_cycles_per_wave_entry = 16 * VALU_F64_CYCLES           # degree-13 polynomial + range reduction and scaling
    lane_rate = 1024 * 64 * NOMINAL_GHZ * 1e9                  # lane-cycles per second
    floor_s = (roots * cycles_per_wave_root + float(pairs) * exp_cycles_per_wave_entry) / lane_rate
    t = pair_ms * 1e-3
    ach = roots / t / 1e9 if t else 0.0
    peak = roots / floor_s / 1e9 if floor_s else 0.0
    hbm = written_bytes / t / 1e9 if t else 0.0
    return {"bound": "valu_f64", "achieved": ach, "peak": peak, "unit": "G(pair,qubit) roots/s", "frac": ach / peak if peak else 0.0,
            "traffic": None, "kernel": "qk::pqk_kernel", "hbm_GBps_written": hbm, "hbm_frac": hbm / 8000.0,
            "valu_floor_ms": floor_s * 1e3, "clock_ghz_assumed": NOMINAL_GHZ,
            "note": f"bound by FP64 vector issue: {PQK_OPS_PER_ROOT} FP64 ops x {VALU_F64_CYCLES:.0f} cycles + one v_rsq_f64 x "
                    f"{VALU_RSQ_F64_CYCLES:.0f} cycles per (pair, qubit) and wave, + 16 FMA slots per entry for exp(); pairs = the "
                    "entries the kernel computes (upper triangle of a symmetric block); peak is that work at 2.4 GHz on all 1024 "
                    "SIMDs with nothing else issued. hbm_frac (8 bytes written per delivered entry / 8 TB/s) is kept beside it: "
                    "the kernel cannot reach that roof. At N>1 the time includes the RDM all-gather"}


def pqk_main(a, rank, world, dev, dist, backend=None):
    """configs[3]: projected quantum kernel.  Each rank simulates its row shard straight into the RDM
    table (fused, no statevectors in HBM), ONE all-gather of that table, then the
    rank's [N/G, N] block of exp(-gamma sum_q ||rho_i - rho_j||_F).  One JSON line, same contract."""
    from quake_amd import device as qdev, sharded
    N = a.rows if a.rows != 65536 else 32768
    n = a.qubits if a.qubits != 16 else 12
    sharded.check_shape(N, world)
    r0, r1 = sharded.row_range(rank, world, N)
    fmap, F, label = pqk_circuit(a, n)
    rng = np.random.default_rng(1004)
    if a.pqk_map == "zz":
        Xh = rng.uniform(-np.pi / 2 + 1e-5, np.pi / 2 - 1e-5, size=(N, F))
    else:
        Xh = rng.uniform(1e-6, 1 - 1e-6, size=(N, F))
    X = torch.from_numpy(Xh[r0:r1].copy()).to(dev)
    circ = qdev.DeviceCircuit(fmap)
    rehearsal = os.environ.get("QK_BENCH_REHEARSAL") == "1"
    ev = []

    def step(timed):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        rdm = qdev.simulate_rdm1(circ, X)
        e[1].record()
        K = sharded.sharded_projected_kernel(rdm, 1.0, rank, world, host_staging=rehearsal)
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
    if world > 1 and K is not None and rehearsal:
        # the rank's block against one computed from a table this rank built alone
        full = qdev.simulate_rdm1(circ, torch.from_numpy(Xh).to(dev))
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
            "config": {"workload": f"{label} projected kernel (1-RDM + Gaussian), N={N}",
                       "rows": N, "qubits": n, "parallelism": f"row-block x{world}"},
            "roofline": pqk_roofline(n, (r1 - r0) * N if world > 1 else N * (N + 1) // 2, written, pair_ms),
            "stages_ms": {"simulate_rdm1": sim_ms, "pairwise": pair_ms}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def host_path(fmap, X_host, K_cpu, rows_cpu, single, calls=3):
    """SURVEY 8d's wall: numpy X in host memory -> H2D -> simulate -> Gram -> D2H -> numpy K (page-locked)
    through the host entry point (HipFidelityKernel.evaluate = qk_evaluate), at the size of the run.

    One call the way a reference user makes it (a fresh array per call: K is page-locked inside the call), then
    `calls` calls into ONE page-locked array kept across them (evaluate(out=...)); value_8d is the mean of those.
    Every call is itemised: the library's own profile of its last call (qk_evaluate_profile) plus what Python adds
    around it (page-locking K, the duplicate-row check)."""
    import ctypes as C
    from quake_amd import _lib
    from quake_amd import kernels as qkern
    from quake_amd.kernels import HipFidelityKernel
    N = X_host.shape[0]
    lib = _lib.load()
    lib.qk_evaluate_profile.restype = C.c_int
    lib.qk_evaluate_profile.argtypes = [C.POINTER(C.c_double), C.c_int]
    k = HipFidelityKernel(fmap, precision="single" if single else "double")
    names = ["device_alloc", "h2d_inputs", "enqueue", "device_work_with_band_copies", "last_band_copy", "device_free", "library_total"]

    def one(out):
        prof = (C.c_double * 8)()
        pin = {"s": 0.0}
        real_pin = qkern._pin_equal_rows

        def timed_pin(*a):
            t = time.perf_counter()
            real_pin(*a)
            pin["s"] += time.perf_counter() - t

        qkern._pin_equal_rows = timed_pin
        try:
            t0 = time.perf_counter()
            K = k.evaluate(X_host, out=out)
            dt = time.perf_counter() - t0
        finally:
            qkern._pin_equal_rows = real_pin
        lib.qk_evaluate_profile(prof, 8)
        item = {n: prof[i] for i, n in enumerate(names)}
        item["duplicate_row_check"] = pin["s"]
        item["python_rest"] = dt - item["library_total"] - pin["s"]     # fresh call: page-locking K is in here
        return K, dt, item

    t0 = time.perf_counter()
    K, dt_fresh, item_fresh = one(None)
    out = {"api": "HipFidelityKernel.evaluate (numpy in, page-locked numpy out: qk_evaluate)", "rows": N,
           "covers": "H2D of X + device allocations + simulate + Gram in row bands + D2H of K overlapped with the later bands "
                     "+ duplicate-row check; fresh_call also page-locks a new K",
           "fresh_call": {"seconds": dt_fresh, "entries_per_s": N * N / dt_fresh, "itemised_s": item_fresh}}
    reuse = []
    for _ in range(calls):
        _, dt, item = one(K)
        reuse.append((dt, item))
    if reuse:
        mean = float(np.mean([d for d, _ in reuse]))
        out["reused_K"] = {"calls": len(reuse), "seconds_each": [d for d, _ in reuse], "seconds_mean": mean,
                           "entries_per_s": N * N / mean,
                           "itemised_s_mean": {n: float(np.mean([it[n] for _, it in reuse])) for n in reuse[0][1]}}
        out["seconds"], out["entries_per_s"] = mean, N * N / mean
    else:
        out["seconds"], out["entries_per_s"] = dt_fresh, N * N / dt_fresh
    sym = float(np.abs(K[:2048, :2048] - K[:2048, :2048].T).max())
    out["max_asymmetry_first_2048"] = sym
    if K_cpu is not None:
        out["max_abs_err_vs_oracle_sample"] = float(np.abs(K[:rows_cpu, :rows_cpu] - K_cpu).max())
    del K
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if os.environ.get("QK_BENCH_LAUNCH_TEST") == "1":
        # launcher plumbing only (tests/test_bench_launcher.py, no GPU): rendezvous over gloo, one collective,
        # one line from rank 0
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.tensor([float(rank)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            assert int(t.item()) == world - 1
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_test": True, "n_gpus": world, "steps": a.steps, "warmup": a.warmup}), flush=True)
        return
    probe = mfma_probe() if (world == 1 and not a.no_probe) else None   # child process, before this one touches the GPU
    # QK_BENCH_REHEARSAL=1: rehearse the N>1 code path on a ONE-GPU box -- every rank uses cuda:0 and the
    # exchanges go over gloo (RCCL refuses two ranks on one device).  Timings of such a run mean nothing.
    rehearsal = world > 1 and os.environ.get("QK_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import datetime
    import torch.distributed as dist
    backend = None
    if world > 1:
        # a rank that never arrives (or a collective that never completes) ends the job after this long instead of
        # hanging it: the rendezvous and every collective below inherit the limit
        limit = datetime.timedelta(seconds=float(os.environ.get("QK_BENCH_DIST_TIMEOUT", "600")))
        backend = "gloo" if rehearsal else "nccl"
        if rehearsal:
            dist.init_process_group("gloo", timeout=limit)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=limit)

    from quake_amd import _lib, device as qdev, featuremaps as fm, sharded
    from quake_amd.circuit import ParameterVector

    if a.workload == "pqk":
        return pqk_main(a, rank, world, dev, dist, backend)
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

    fmap = fm.zz_featuremap(ParameterVector("x", n), n, a.repeats)
    circ = qdev.DeviceCircuit(fmap)
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

    def step(timed, K_host=None):
        t = timer if timed else None
        if t: t("sim_begin")
        circ.simulate(X, out=psi[r0:r1] if gather else psi)
        if t: t("sim_end")
        work = None
        if gather:
            # in-place all-gather of the shards (float64 views: RCCL has no complex dtype)
            work = dist.all_gather_into_tensor(psi_real, psi_real[r0:r1], async_op=True)
        sg.compute(psi, K, gather_work=work, timer=t, K_host=K_host)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(a.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    fence()
    dt = max_over_ranks(time.perf_counter() - t0)

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

    # ---- legs outside `value`.  Each runs inside leg(): an exception is recorded in the line instead of losing it,
    # and with several ranks they agree after every leg whether all of them got through -- once one has not, the
    # remaining legs (which hold collectives) are skipped everywhere, so rank 0 still prints the measured line.
    state = {"comm_ok": True}

    def leg(fn):
        if not state["comm_ok"]:
            return {"skipped": "an earlier leg failed on some rank"}
        ok, res = 1.0, None
        try:
            res = fn()
        except Exception as e:
            ok, res = 0.0, {"error": repr(e)}
        if world > 1:
            try:
                t = torch.tensor([ok], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                if float(t.item()) < 1.0:
                    state["comm_ok"] = False
                    if ok:
                        res = {"error": "this leg failed on another rank", "partial": res}
            except Exception as e:
                state["comm_ok"] = False
                res = {"error": "ranks lost contact after this leg: " + repr(e), "partial": res}
        elif not ok:
            state["comm_ok"] = True     # one rank: the next leg does not depend on this one
        return res

    # the K block -> host leg of the N>1 path (north_star: "each GPU computes its K block, which is gathered to
    # host"): one more step whose row block also lands in page-locked host memory, own blocks copied while the
    # mirrored ones travel
    def k_to_host_leg():
        K_host = torch.from_numpy(_lib.host_empty((r1 - r0, N)))
        fence()
        t1 = time.perf_counter()
        step(False, K_host=K_host)
        fence()
        dth = max_over_ranks(time.perf_counter() - t1)
        bad = float((K_host[:, :min(N, 4096)] - K[:, :min(N, 4096)].cpu()).abs().max())
        bad = max_over_ranks(bad)
        return {"ms_per_step_with_k_to_host": dth * 1e3, "entries_per_s": N * N / dth,
                "host_bytes_per_rank": 8.0 * (r1 - r0) * N, "max_abs_diff_host_vs_device_block": bad,
                "note": "one extra step after the timed ones; every rank's [N/G, N] block copied into page-locked host "
                        "memory on a side stream (qk_block_to_host_async), overlapped with the mirror exchange"}

    k_to_host = leg(k_to_host_leg) if (world > 1 and not a.no_host_path) else None

    # configs[4] also names 16384 test rows: the test x train cross block, sharded like the train rows
    # (each rank: its T/G test rows against the replicated train statevectors), outside `value`
    T = a.test_rows if a.test_rows >= 0 else N // 4

    def test_block_leg():
        q0, q1 = sharded.row_range(rank, world, T)
        Xt_host = np.random.default_rng(2005).uniform(lo, hi, size=(T, n))
        Xt = torch.from_numpy(Xt_host[q0:q1].copy()).to(dev)
        psi_t = torch.empty((q1 - q0, D), dtype=psi.dtype, device=dev)
        Kt = torch.empty((q1 - q0, N), dtype=torch.float64, device=dev)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        fence()
        t2 = time.perf_counter()
        e[0].record()
        circ.simulate(Xt, out=psi_t)
        e[1].record()
        sharded.sharded_cross_gram(psi_t, psi, out=Kt)
        e[2].record()
        fence()
        dtt = max_over_ranks(time.perf_counter() - t2)
        t_ms = e[1].elapsed_time(e[2])
        return {"rows": T, "rows_per_rank": q1 - q0, "seconds": dtt, "entries_per_s": T * N / dtt,
                "simulate_ms": e[0].elapsed_time(e[1]), "gram_ms": t_ms,
                "gram_tflops_algorithmic": 8.0 * D * (q1 - q0) * N / (t_ms * 1e-3) / 1e12}

    test_block = leg(test_block_leg) if (T > 0 and T % world == 0) else None

    gram_ms = float(np.mean([b.elapsed_time(e) for b, e in zip(ev["gram_begin"], ev["gram_end"])])) if a.steps else 0.0
    sim_ms = float(np.mean([b.elapsed_time(e) for b, e in zip(ev["sim_begin"], ev["sim_end"])])) if a.steps else 0.0
    entries_rank = sharded.computed_entries(rank, world, N)
    flops_rank = 8.0 * D * entries_rank
    algorithmic = flops_rank / (gram_ms * 1e-3) / 1e12 if gram_ms > 0 else 0.0
    hardware = MFMA_FLOPS_PER_ALGORITHMIC * algorithmic

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
                "bound": "mfma", "achieved": hardware, "peak": PEAK, "unit": "TFLOP/s",
                "frac": hardware / PEAK, "traffic": None,
                "kernel": "qk::gram32_dma_kernel<2,2,2,2>" if single else gram_kernel_name(N // world, D),
                "ms_per_launch_set": gram_ms,
                "mfma_flops": MFMA_FLOPS_PER_ALGORITHMIC * flops_rank,
                "algorithmic_flops": flops_rank,
                "algorithmic_tflops": algorithmic,
                "algorithmic_frac": algorithmic / PEAK,
                "note": "achieved / frac = flops the matrix cores execute (rank 0's share) / kernel time, per-GPU peak: "
                        "the kernel uses the 3-multiplication (Karatsuba) complex product, 6*D MFMA flops per computed "
                        "entry (what SQ_INSTS_VALU_MFMA_MOPS_F64 counts) for the 8*D algorithmic flops of SURVEY 8d; "
                        "algorithmic_tflops / algorithmic_frac price the same time at 8*D and may exceed 1. traffic is "
                        "null: no counters are collected in this run (traffic_pmc_committed quotes the committed "
                        "rocprofv3 --pmc pass of this workload, when there is one)",
            },
            "stages_ms": {"simulate": sim_ms, "gram": gram_ms},
        }
        committed = None if single else pmc_traffic_committed(N, n, a.repeats, world)
        if committed:
            out["roofline"]["traffic_pmc_committed"] = committed
        if world > 1:
            out["config"]["backend"] = backend
            if rehearsal:
                out["config"]["note"] = ("REHEARSAL: every rank shares cuda:0 and the exchanges go over gloo through host memory "
                                         "(QK_BENCH_REHEARSAL=1) -- this checks the N>1 code path, its timings mean nothing")
            else:
                out["config"]["note"] = ("one process per GPU over RCCL (backend nccl); this line is one run at this GPU count -- a "
                                         "scaling curve exists only where the driver ran 1/2/4/8 back to back (SCALE_rNN.json)")
        if test_block:
            out["test_block"] = test_block
        if k_to_host:
            out["k_to_host"] = k_to_host
        if probe and not single:
            out["roofline"]["mfma_probe_tflops"] = probe["tflops"]
            out["roofline"]["frac_of_probe"] = hardware / probe["tflops"]
            out["roofline"]["note"] += ("; mfma_probe_tflops = a register-only MFMA loop timed on this box just before "
                                        "the run (its ceiling at the clock it sustains), frac_of_probe = achieved / that")
        K_cpu, rows_cpu = None, 0
        if world == 1 and not a.no_cpu_baseline:
            top = a.cpu_rows or max(64, min(1024, int(1024 * (65536 / D) ** 0.5)))
            top = min(top, N)
            rows_list = sorted({max(32, min(top, r)) for r in (top * 3 // 8, top // 2, top)})
            base, K_cpu = cpu_baseline(X_host, n, a.repeats, rows_list)
            rows_cpu = rows_list[-1]
            out["cpu_baseline"] = base
            err = float(np.abs(K[:rows_cpu, :rows_cpu].cpu().numpy() - K_cpu).max())
            out["check_max_abs_err_vs_oracle"] = err
        if world == 1 and not a.no_host_path:
            # free the bench's own 96 GiB first: the host entry point allocates its own Psi and K
            del psi, psi_real, K, sg
            torch.cuda.empty_cache()
            try:
                out["host_path"] = host_path(fmap, X_host, K_cpu, rows_cpu, single, calls=a.host_calls)
                hp = out["host_path"]
                # SURVEY 8d's wall (host X -> H2D -> simulate -> Gram -> D2H -> host K), a mean over calls; `value`
                # stays the device-resident figure the bench contract asks for
                out["value_8d"] = {"value": hp["entries_per_s"], "unit": "entries/s", "seconds_per_call": hp["seconds"],
                                   "calls": hp.get("reused_K", {}).get("calls", 1),
                                   "what": "PCIe-inclusive: numpy X in, page-locked numpy K out, one K array kept across calls "
                                           "(evaluate(out=...)); host_path.fresh_call is the same with a new K per call"}
            except Exception as e:   # e.g. the box cannot page-lock 32 GiB: say so instead of losing the line
                out["host_path"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # a rank that fails says why and leaves with a non-zero status at once (no clean-up that could wait for the
        # other ranks, no re-exec): torch.distributed.run then ends the job and hands the status on
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(1)
