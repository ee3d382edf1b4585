#!/usr/bin/env python3
"""Turn what tools/run_profiles_r03.sh left under gpurun_out/r03/ into the committed summaries under profiles/:
r03_bench_default.json, r03_bench_default_kernel_trace.md, r03_pmc.md, r03_pmc_traffic.json, r03_configs.json,
r03_sim_bench.md, r03_ga_kernel_trace.md, r03_latency.json, r03_pqk.json."""
import json, shutil, sqlite3, subprocess, sys
O = "gpurun_out/r03/"
P = "profiles/r03_"
with open(P + "bench_default_kernel_trace.md", "w") as f:
    subprocess.check_call([sys.executable, "tools/rocpd_summary.py", O + "kt/kt_results.db"], stdout=f)
# the Gram launches of the bench run one by one, next to what bench.py measured with HIP events in the same run
with open(P + "bench_default_kernel_trace.md", "a") as f:
    c = sqlite3.connect(O + "kt/kt_results.db").cursor()
    rows = c.execute("select name, duration, grid_x from kernels where name like '%gram%' order by start").fetchall()
    b = json.loads(open(O + "bench.json").read().strip().splitlines()[-1])
    full = max(r[2] for r in rows)
    timed = [r[1] for r in rows if r[2] == full][-b["steps"]:]
    f.write("\n## Gram launches of this run in launch order (ms)\n\n")
    f.write(", ".join("%.3f" % (r[1] / 1e6) for r in rows) + "\n\n")
    f.write("The last %d launches over the full N = %d block are the timed steps: average %.3f ms; `bench.py` in the same run, "
            "HIP events on the launch stream: ms_per_launch_set = %.3f ms, ms_per_step (wall, barriers included) = %.3f ms.\n"
            % (len(timed), b["config"]["rows"], sum(timed) / len(timed) / 1e6, b["roofline"]["ms_per_launch_set"], b["ms_per_step"]))
with open(P + "ga_kernel_trace.md", "w") as f:
    subprocess.check_call([sys.executable, "tools/rocpd_summary.py", O + "ga_kt/kt_results.db"], stdout=f)
with open(P + "sim_kernel_trace.md", "w") as f:
    f.write("`rocprofv3 --kernel-trace --stats -- python3 tools/sim_bench.py --qubits 12 8 --gib 4 --maps zz_r2 genome_g8`: the register-resident simulator and its pre-pass kernels\n\n")
    f.flush()
    subprocess.check_call([sys.executable, "tools/rocpd_summary.py", O + "sim_kt/kt_results.db"], stdout=f)
shutil.copy(O + "bench.json", P + "bench_default.json")
shutil.copy(O + "configs.json", P + "configs.json")
shutil.copy(O + "latency.json", P + "latency.json")
with open(P + "pqk.json", "w") as f:
    f.write(open(O + "pqk_zz.json").read() + open(O + "pqk_genome.json").read())
with open(P + "ga_bench.json", "w") as f:
    f.write(open(O + "ga_bench.json").read())


def counters(p, pick):
    c = sqlite3.connect(O + p + "/p_results.db").cursor()
    out = {}
    for name, cn, did, val, dur in c.execute("select kernel_name, counter_name, dispatch_id, value, duration from counters_collection order by dispatch_id"):
        k = pick(name)
        if k:
            out.setdefault((k, did), {"dur_ns": dur, "name": name})[cn] = val
    return out


gs = lambda name: "gram" if "gram" in name else ("sim" if ("sim_pass" in name or "wave_sim" in name or "reg_sim" in name) else None)
A, B, C = counters("pmcA", gs), counters("pmcB", gs), counters("pmcC", gs)
first = lambda d, k: [v for (kk, _), v in sorted(d.items(), key=lambda kv: kv[0][1]) if kk == k][0]
gA, gB, gC, sA, sB = first(A, "gram"), first(B, "gram"), first(C, "gram"), first(A, "sim"), first(B, "sim")
fetch, write = gA["FETCH_SIZE"] * 1024 * 2, gB["WRITE_SIZE"] * 1024
N = D = 65536
alg = N * D * 16 + N * N * 8
cyc = gC["GRBM_GUI_ACTIVE"] / 8
t = gC["dur_ns"] * 1e-9
mfma_flops = gC["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512
util = gC["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
md = f"""# r03 PMC passes

Commands: tools/run_profiles_r03.sh (one counter block per `rocprofv3 --pmc` run, no trace flags beside --kernel-trace,
FETCH_SIZE x2 on gfx950, as MI355X_MICROARCH.md prescribes).  Summaries by tools/make_profiles_r03.py.

## default bench (N=65536, 16 qubits, zz r=1, 1 GPU): {gC['name'][:60]} -- one launch, {t:.3f} s on this box

| quantity | raw counter | derived |
|---|---|---|
| FETCH_SIZE | {gA['FETCH_SIZE']:.6g} KiB | x1024 x2 = **{fetch:.4g} B** read at the memory side of L2 = {fetch/gA['dur_ns']:.0f} GB/s |
| WRITE_SIZE | {gB['WRITE_SIZE']:.6g} KiB | x1024 = **{write:.4g} B** (K is N*N*8 = {N*N*8:.4g} B) |
| compulsory traffic | | Psi once + K once = {alg:.4g} B; Psi tiles are re-read {fetch/(N*D*16):.0f}x through L2 ({fetch/gA['dur_ns']/1e3:.2f} TB/s, under the HBM roof: the kernel is MFMA-bound) |
| GRBM_GUI_ACTIVE | {gC['GRBM_GUI_ACTIVE']:.6g} (sum over 8 XCDs) | {cyc:.4g} cycles = **{cyc/t/1e9:.3f} GHz** sustained |
| SQ_INSTS_VALU_MFMA_MOPS_F64 | {gC['SQ_INSTS_VALU_MFMA_MOPS_F64']:.6g} (x512 flops) | **{mfma_flops:.4g} hardware MFMA flops** = 6*D per computed entry; {mfma_flops/t/1e12:.1f} TFLOP/s = {mfma_flops/t/1e12/78.6:.3f} of 78.6 |
| SQ_VALU_MFMA_BUSY_CYCLES | {gC['SQ_VALU_MFMA_BUSY_CYCLES']:.6g} | / (cycles x 1024 SIMDs) = **{util*100:.1f}% MFMA pipe busy** |
| SQ_WAIT_INST_ANY / SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES | {gC['SQ_WAIT_INST_ANY']:.4g} / {gC['SQ_WAIT_ANY']:.4g} / {gC['SQ_ACTIVE_INST_ANY']:.4g} / {gC['SQ_WAVE_CYCLES']:.4g} | issue-stall (matrix pipe back-pressure) {gC['SQ_WAIT_INST_ANY']/gC['SQ_WAVE_CYCLES']*100:.0f}% of wave cycles, parked on waitcnt/barrier {gC['SQ_WAIT_ANY']/gC['SQ_WAVE_CYCLES']*100:.0f}% |

Simulator inside the same run ({sA['name'][:50]}, N=65536 samples, one write-only pass): {sA['dur_ns']/1e6:.2f} ms,
WRITE_SIZE {sB['WRITE_SIZE']*1024:.4g} B (= N * 2^16 * 16 B, the algorithmic bytes) -> {sB['WRITE_SIZE']*1024/sB['dur_ns']:.0f} GB/s; FETCH_SIZE {sA['FETCH_SIZE']:.6g} KiB (tables only).

## simulator kernels: `tools/sim_bench.py --qubits q --gib 1 --maps zz_r1 zz_r2 genome_g8` (per dispatch, launch order: 4 launches per map)

8, 10 and 12 qubits run on the register-resident kernel (qk::reg_sim_kernel<log2 threads, 4, packed>; at 8 qubits 8 samples share a
workgroup; zz_r1 at 12 qubits stays on the LDS kernel's direct-store pass), 16 qubits on the LDS kernel (qk::sim_pass_kernel).  Gate
angles, phase tables and matrices come from the pre-pass kernels (qk::angle_table_kernel, qk::reg_table_kernel, qk::reg_matrix_kernel:
separate dispatches, not listed here; their time is in r03_sim_kernel_trace.md).
Per SAMPLE figures = counter / samples of the dispatch (wave-instructions, not lane-instructions).

| qubits | kernel | dispatch | ms | samples | SQ_INSTS_VALU /sample | SQ_INSTS_SALU /sample | SQ_INSTS_LDS /sample | VALU lane-instr / amplitude | ACTIVE / WAIT_ANY / WAIT_INST of wave cycles |
|---|---|---|---|---|---|---|---|---|---|
"""
for q in (8, 10, 12, 16):
    S = counters("sim%d" % q, gs)
    n_samp = (1 << 30) // (16 << q)
    for (k, d), v in sorted(S.items(), key=lambda kv: kv[0][1]):
        wc = v["SQ_WAVE_CYCLES"]
        md += (f"| {q} | {v['name'][5:40]} | {d} | {v['dur_ns']/1e6:.3f} | {n_samp} | {v['SQ_INSTS_VALU']/n_samp:.0f} | {v['SQ_INSTS_SALU']/n_samp:.0f} | "
               f"{v['SQ_INSTS_LDS']/n_samp:.0f} | {v['SQ_INSTS_VALU']*64/(n_samp*(1<<q)):.0f} | "
               f"{v['SQ_ACTIVE_INST_ANY']/wc*100:.0f}% / {v['SQ_WAIT_ANY']/wc*100:.0f}% / {v['SQ_WAIT_INST_ANY']/wc*100:.0f}% |\n")
md += """
Reading: on deep circuits the simulators are bound by instruction issue and by waits on LDS / barriers, not by HBM.
zz_r2 at 16 qubits is two passes (two dispatches per call).  Round 2's figures for the same circuits: r02_pmc.md.
"""


def pqk_counters(p):
    c = sqlite3.connect(O + p + "/p_results.db").cursor()
    rows = {}
    for name, cn, did, val, dur in c.execute("select kernel_name, counter_name, dispatch_id, value, duration from counters_collection order by dispatch_id"):
        if "pqk_kernel" in name:
            rows.setdefault(did, {"dur_ns": dur})[cn] = val
    return rows[sorted(rows)[-1]]


G, H = pqk_counters("pmcG"), pqk_counters("pmcH")
Np, nq = 32768, 12
roots = Np * (Np + 1) // 2 * nq
cyc = G["GRBM_GUI_ACTIVE"] / 8
f64 = G["SQ_INSTS_VALU_FMA_F64"] + G["SQ_INSTS_VALU_ADD_F64"] + G["SQ_INSTS_VALU_MUL_F64"]
busy = G["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024)
md += f"""
## qk::pqk_kernel: `tools/config_bench.py 4` (projected kernel, N=32768, 12 qubits, symmetric; last launch)

| quantity | value |
|---|---|
| duration | {G['dur_ns']/1e6:.2f} ms at {cyc/G['dur_ns']:.2f} GHz (GRBM_GUI_ACTIVE / 8) |
| WRITE_SIZE | {H['WRITE_SIZE']*1024:.4g} B (K is N*N*8 = {Np*Np*8:.4g} B, the algorithmic bytes) -> {H['WRITE_SIZE']*1024/H['dur_ns']:.0f} GB/s = {H['WRITE_SIZE']*1024/H['dur_ns']/8000:.3f} of 8 TB/s |
| SQ_INSTS_VALU_TRANS_F64 | {G['SQ_INSTS_VALU_TRANS_F64']:.6g} wave instructions = N(N+1)/2 * n / 64 = {roots/64:.6g}: one v_rsq_f64 per (pair, qubit) |
| SQ_INSTS_VALU_{{FMA,ADD,MUL}}_F64 | {G['SQ_INSTS_VALU_FMA_F64']:.4g} + {G['SQ_INSTS_VALU_ADD_F64']:.4g} + {G['SQ_INSTS_VALU_MUL_F64']:.4g} = {f64/G['SQ_INSTS_VALU_TRANS_F64']:.1f} FP64 operations per root (3 differences, 3 squares, 4 for the root and its accumulation; the rest is the exp epilogue) |
| SQ_INSTS_VALU | {G['SQ_INSTS_VALU']:.6g} ({G['SQ_INSTS_VALU']/G['SQ_INSTS_VALU_TRANS_F64']:.1f} per root; 17.8 in round 1) |
| SQ_ACTIVE_INST_VALU | {G['SQ_ACTIVE_INST_VALU']:.6g} (units of 4 cycles) / (cycles x 1024 SIMDs) = **{busy*100:.0f}% VALU busy** |

Reading: the pairwise kernel is bound by FP64 VALU issue, not by the 8 bytes per entry it writes.
"""
open(P + "pmc.md", "w").write(md)
json.dump({"workload": {"rows": 65536, "qubits": 16, "repeats": 1, "gpus": 1},
           "gram_hbm_bytes_per_launch": fetch + write, "gram_fetch_bytes": fetch, "gram_write_bytes": write,
           "source": "profiles/r03_pmc.md: FETCH_SIZE KiB x1024 x2 (gfx950 correction for 16-B/lane loads) + WRITE_SIZE KiB x1024, separate --pmc passes"},
          open(P + "pmc_traffic.json", "w"), indent=1)

# simulator A/B table
rows = {}
for tag, f in (("wave / default", "sim_bench.json"), ("lds forced", "sim_bench_noreg.json")):
    for line in open(O + f):
        if line.startswith("{"):
            r = json.loads(line)
            rows.setdefault((r["n"], r["map"]), {})[tag] = r
sb = "# r03 simulator bench: `tools/sim_bench.py --gib 4` (4 GiB of Psi per launch; algorithmic bytes = 16 * 2^n per sample)\n\n"
sb += "Default path (round 3): the register-resident kernel for every circuit of up to 10 qubits that has such a plan (several samples per workgroup below 10 qubits) and for 11-13 qubits unless the circuit is the product state and one diagonal stage (direct-store pass of the LDS kernel); the LDS kernel above 13 qubits.  `QK_SIM_PATH=noreg` = the round-2 paths (wave-resident kernel up to 10 qubits, LDS kernel above): last two columns, its time and its time / the default path's.\n\n"
sb += "| qubits | circuit | samples | default ms | default GB/s | frac of 8 TB/s | round-2 path ms | speed-up |\n|---|---|---|---|---|---|---|---|\n"
for (n, m), v in sorted(rows.items(), key=lambda kv: (-kv[0][0], kv[0][1])):
    d = v.get("wave / default")
    l = v.get("lds forced")
    if not d:
        continue
    sb += f"| {n} | {m} | {d['N']} | {d['ms']} | {d['alg_GBps']} | {d['frac_of_8TBps']} | {l['ms'] if l else ''} | {(str(round(l['ms']/d['ms'], 2)) + 'x') if l else ''} |\n"
open(P + "sim_bench.md", "w").write(sb)
print(md[:3000]); print(sb)
