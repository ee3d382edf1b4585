#!/usr/bin/env python3
"""Same-box A/B of two builds of the library on tools/sim_bench.py lines (fresh process per run, three rounds).
usage: python tools/sim_ab.py <libA.so> <libB.so> [sim_bench arguments ...]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bench = os.path.join(root, "tools", "sim_bench.py")
libs = [os.path.abspath(p) for p in sys.argv[1:3]]
args = sys.argv[3:] or ["--qubits", "16", "14", "12", "--maps", "zz_r2"]
code = ("import sys; sys.path.insert(0, %r)\nfrom pathlib import Path\nfrom quake_amd import _lib\n_lib.LIB_PATH = Path(%r)\n"
        "sys.argv = ['sim_bench.py'] + %r\nf = %r\nexec(compile(open(f).read(), f, 'exec'), {'__file__': f, '__name__': '__main__'})\n")
for rnd in range(3):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", code % (root, lib, args, bench)], capture_output=True, text=True).stdout
        print(os.path.basename(lib), [json.loads(l)["ms"] for l in out.splitlines() if l.startswith("{")], flush=True)
