"""Host logic of the simulator's planner (no GPU: qk_circuit_create only plans): which passes, workgroup sizes and
sweep radices the hand-written maps and a GA genome get."""
import re

import numpy as np

from quake_amd import _lib, featuremaps as fm, genetic
from quake_amd.circuit import ParameterVector
from tests.test_circuit_ir import GATE_DICT_FULL


def plan(circ):
    s = _lib.CircuitHandle(circ.compile()).plan_summary
    passes = [dict(L=int(m[0]), T=int(m[1]), maxg=int(m[2]), maxh=int(m[3]), narrow=int(m[4]), u1=int(m[5]), ng=int(m[6]),
                   diag=int(m[7]), other=int(m[8]))
              for m in re.findall(r"pass\d+ L(\d+) T(\d+) maxg(\d+) maxh(\d+) narrow(\d+) u1=(\d+)\(<=(\d+)\) diag=(\d+) other=(\d+)", s)]
    return s, passes


def zz(n, r):
    return fm.zz_featuremap(ParameterVector("x", n), n, r)


def test_zz_r1_is_one_write_only_pass_at_any_width():
    for n in (4, 8, 12, 16):
        s, p = plan(zz(n, 1))
        assert len(p) == 1 and p[0]["u1"] == 0 and p[0]["diag"] == 1 and p[0]["other"] == 0, s
        assert p[0]["narrow"] == 0 and "rdm_plan no" in s


def test_hadamard_layers_go_four_to_a_sweep():
    # 12 local qubits: 256 threads, 16 amplitudes each, a layer of 12 Hadamards = 3 sweeps
    s, p = plan(zz(12, 2))
    assert len(p) == 1 and p[0] == dict(L=12, T=256, maxg=3, maxh=4, narrow=0, u1=3, ng=4, diag=2, other=0), s
    s, p = plan(zz(12, 3))
    assert p[0]["u1"] == 6 and p[0]["diag"] == 3, s
    # 11 and 10 qubits: fewer threads so that a thread still holds 16 amplitudes
    s, p = plan(zz(11, 2))
    assert p[0]["T"] == 128 and p[0]["maxh"] == 4 and p[0]["u1"] == 3 and p[0]["ng"] == 4, s
    s, p = plan(zz(10, 2))
    assert p[0]["T"] == 64 and p[0]["maxh"] == 4 and p[0]["u1"] == 3, s
    # 13 qubits: one pass of 13 local qubits, 512 threads
    s, p = plan(zz(13, 2))
    assert len(p) == 1 and p[0]["L"] == 13 and p[0]["T"] == 512 and p[0]["ng"] == 4 and p[0]["u1"] == 4, s
    # 16 qubits: two passes of 12 local qubits; the second one has the four high qubits' Hadamards in one sweep
    s, p = plan(zz(16, 2))
    assert len(p) == 2 and all(q["L"] == 12 and q["maxh"] == 4 for q in p), s
    assert p[0]["u1"] == 3 and p[1]["u1"] == 1 and p[1]["ng"] == 4, s


def test_genome_pass_runs_wide_and_keeps_a_plan_for_the_fused_rdm_mode():
    n, G, F = 12, 8, 12
    genes = genetic.initial_population(1, n, G, GATE_DICT_FULL, F, rng=np.random.default_rng(7))[0]
    circ, _ = genetic.to_quantum(genes, GATE_DICT_FULL, F, G, n, physical_qubits=list(range(n)), verbose=False)
    s, p = plan(circ)
    assert len(p) == 1 and p[0]["L"] == 12, s
    if p[0]["u1"] + p[0]["other"] > 0:      # butterflies or swaps beyond the product state: 512 threads, radix 4
        assert p[0]["T"] == 512 and p[0]["narrow"] == 1 and p[0]["maxg"] == 2 and p[0]["ng"] <= 2, s
        assert "rdm_plan yes" in s
