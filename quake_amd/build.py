"""Build libquake_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
INCLUDE = PKG_DIR.parent / "include"
LIB_PATH = PKG_DIR / "libquake_hip.so"
SOURCES = ["qk_api.hip", "qk_circuit.cpp", "qk_sim.hip", "qk_wave.hip", "qk_reg.hip", "qk_gram.hip", "qk_pqk.hip", "qk_stats.hip"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found; libquake_hip.so cannot be built")
    return exe


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [CSRC / "qk_common.h", CSRC / "qk_rdm.h", CSRC / "qk_math.h", CSRC / "qk_wave.h", CSRC / "qk_reg.h", INCLUDE / "quake_hip.h"]
    return any(d.stat().st_mtime > t for d in deps)


def build_library(force: bool = False, verbose: bool = True, extra_flags=()) -> Path:
    if not force and not needs_build():
        return LIB_PATH
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{INCLUDE}", f"-I{CSRC}",
             "-Wall", "-Wno-unused-function", *extra_flags]
    # qk_wave.hip keeps a statevector in registers across a large scalar dispatch; two middle-end transforms
    # defeat that (common-code sinking merges the per-qubit arms into pointer selects, the AMDGPU alloca-to-vector
    # pass turns the amplitude array into one 32-register tuple): both off for that file
    keep_in_regs = ["-mllvm", "-simplifycfg-sink-common=false", "-mllvm", "-disable-promote-alloca-to-vector"]
    # In addition: the AMDGPU back end structurizes wave-uniform control flow as well (its StructurizeCFG runs
    # with SkipUniformRegions off), which turns the op dispatch's if / else-if chain into a chain of flow blocks that
    # carry BOTH the old and the new amplitudes -- twice the registers (247 instead of 128) and a copy of all of them
    # at every join.  Uniform regions are left as scalar branches instead.
    uniform_flow = keep_in_regs + ["-mllvm", "-structurizecfg-skip-uniform-regions"]
    # (the LDS simulator's stage dispatch gains a few per cent from the same switch: 16 qubits ZZ r = 2 3.10 -> 3.02 ms per 4 GiB;
    # the Gram and projected-kernel files compile to the same code with and without it and are left alone)
    per_file = {"qk_wave.hip": uniform_flow, "qk_reg.hip": uniform_flow, "qk_sim.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions"]}
    procs = []
    for s in SOURCES:
        obj = objdir / (s.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc(), *flags, *per_file.get(s, []), "-x", "hip", "-c", str(CSRC / s), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, obj, subprocess.Popen(cmd)))
    objs = []
    for s, obj, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
        objs.append(str(obj))
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", str(LIB_PATH)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


def build_probe(verbose: bool = True) -> Path:
    """tools/mfma_f64_peak.bin: the register-only MFMA loop bench.py runs as a ceiling probe."""
    src = PKG_DIR.parent / "tools" / "mfma_f64_peak.hip"
    out = src.with_suffix(".bin")
    if out.exists() and out.stat().st_mtime >= src.stat().st_mtime:
        return out
    cmd = [hipcc(), "-O3", "--offload-arch=gfx950", "-Wno-unused-value", str(src), "-o", str(out)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB_PATH)
