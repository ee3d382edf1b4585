#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2) results .db into the per-kernel stats table that
`--stats` prints: name, calls, total/avg/min/max duration (ns), share, registers.
usage: python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/xxx.md"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path).cursor()
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of `{path}`\n")
    print("| kernel | calls | total ns | avg ns | min ns | max ns | % | vgpr | agpr | sgpr | lds B | scratch B | grid_x | wg_x |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
        print(f"| `{name}` | {r[1]} | {r[2]} | {r[3]:.0f} | {r[4]} | {r[5]} | {100 * r[2] / total:.2f} | "
              f"{r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} | {r[12]} |")
    pm = c.execute("select count(*) from pmc_events").fetchone()[0]
    if pm:
        print("\n## counters (sum over dispatches, per kernel)\n")
        try:
            q = ("select k.name, p.counter_name, count(*), sum(p.value) from counters_collection p "
                 "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name")
            print("| kernel | counter | dispatches | sum |\n|---|---|---|---|")
            for r in c.execute(q):
                print(f"| `{r[0][:60]}` | {r[1]} | {r[2]} | {r[3]:.6g} |")
        except sqlite3.Error as e:
            print(f"(could not join counters: {e})")


if __name__ == "__main__":
    main(sys.argv[1])
