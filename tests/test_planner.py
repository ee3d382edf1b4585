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


def _shared_angle_program():
    """rz(x0) and p(x0) on two qubits reading ONE expression slot: legal through the C-ABI (compile() never emits it)."""
    from quake_amd.circuit import CompiledCircuit, GATE_CODES, QuantumCircuit
    x = ParameterVector("x", 1)
    probe = QuantumCircuit(2)
    probe.rz(x[0], 0)
    one = probe.compile()            # the postfix program of the single expression x0
    ops = np.array([[GATE_CODES["h"], 0, -1, -1], [GATE_CODES["h"], 1, -1, -1], [GATE_CODES["cx"], 0, 1, -1],
                    [GATE_CODES["rz"], 0, -1, 0], [GATE_CODES["p"], 1, -1, 0], [GATE_CODES["cx"], 1, 0, -1],
                    [GATE_CODES["h"], 0, -1, -1]], dtype=np.int32)
    return CompiledCircuit(n_qubits=2, n_params=1, ops=ops, prog=one.prog, expr_off=one.expr_off, consts=one.consts)


def test_wave_program_is_refused_when_one_expression_feeds_half_and_full_angle_gates():
    assert "wave_program yes" in plan(zz(4, 2))[0]
    s = _lib.CircuitHandle(_shared_angle_program()).plan_summary
    assert "wave_program no" in s


def test_wave_program_is_refused_when_its_tables_exceed_the_dynamic_lds_limit():
    """A long narrow circuit: (cos, sin) pairs and composite matrices of one sample beyond 40 KiB (64 KiB less the
    staged program) -> the LDS simulator runs it instead of a launch that HIP would refuse."""
    from quake_amd.circuit import QuantumCircuit
    x = ParameterVector("x", 3)
    def narrow(layers):
        qc = QuantumCircuit(3)
        for k in range(layers):
            for q in range(3):
                qc.rx(x[q] * (k + 1), q)
                qc.rz(x[(q + 1) % 3], q)
                qc.ry(x[q] + 0.1 * k, q)
            qc.cx(0, 1)
            qc.cx(1, 2)
        return qc
    assert "wave_program yes" in plan(narrow(20))[0]
    assert "wave_program no" in plan(narrow(200))[0]       # 1800 expressions + 600 composite matrices: 66 KiB


def test_packed_register_plans_below_ten_qubits():
    """4 to 9 qubits: the register-resident plan packs 2^(L - n) samples into a workgroup (qk_reg.h); deep circuits on
    4-5 qubits, whose per-sample tables and matrices would not fit next to that many samples, have no such plan and
    stay on the wave-resident kernel; 10 qubits and more: one sample per workgroup."""
    from tests.test_gpu_parity import _gate_soup

    def plan_of(circ):
        s = plan(circ)[0]
        m = re.search(r"register-resident plan: (\d+) segments, (\d+) ops, (\d+) tables, (\d+) samples per workgroup of (\d+) threads", s)
        return tuple(int(v) for v in m.groups()) if m else None

    for n in (4, 5, 6, 7, 8, 9):
        p = plan_of(zz(n, 1))
        per_thread = 8 if n <= 7 else 16      # 7 qubits and below: 8 amplitudes per thread on twice the threads
        assert p is not None and p[3] * (1 << n) == per_thread * p[4], (n, p)      # samples x amplitudes
        assert p[3] >= 2 and p[4] in (64, 128, 256, 512)
    assert plan_of(zz(8, 2))[0] == 2 and plan_of(zz(9, 2))[0] == 3           # segments of four register qubits
    assert plan_of(zz(12, 2))[3:] == (1, 256) and plan_of(zz(10, 2))[3:] == (1, 64)
    assert plan_of(_gate_soup(4, 30, 36)) is None      # 64 samples' tables and matrices next to each other: no room
    assert plan_of(_gate_soup(6, 32, 54))[3:] == (32, 256)      # 8 amplitudes per thread: 32 samples, not 64, share a workgroup
    assert plan_of(_gate_soup(8, 34, 72)) is not None and plan_of(_gate_soup(13, 22, 117)) is not None
    assert plan_of(zz(3, 1)) is None and plan_of(zz(14, 2)) is None          # below 4 qubits / two passes


def _reg_plan(circ):
    s = plan(circ)[0]
    m = re.search(r"register-resident plan: (\d+) segments, (\d+) ops, (\d+) tables, (\d+) samples per workgroup of (\d+) threads(, ratio tables for (\d+) register positions)?", s)
    if not m:
        return None
    g = m.groups()
    return {"segments": int(g[0]), "ops": int(g[1]), "tables": int(g[2]), "samples": int(g[3]), "threads": int(g[4]),
            "ratio_positions": int(g[6]) if g[6] else 0}


def test_register_plan_picks_amplitudes_per_thread_by_circuit():
    """qk_circuit.cpp section 4: the Z / ZZ maps -- diagonal stages and Hadamard layers -- keep 16 amplitudes
    per thread from 8 qubits on; circuits with a layer's worth of matrix butterflies and swaps (GA genomes) take 8 on
    twice the threads; 7 qubits and below always 8; a 13-qubit genome whose LDS plan keeps a chunk bit still gets a
    register plan."""
    from tests.test_gpu_parity import _gate_soup
    for n in (8, 10, 12, 13):
        p = _reg_plan(zz(n, 2))
        assert p and p["samples"] * (1 << n) == 16 * p["threads"], (n, p)
    for n in (5, 6, 7):
        p = _reg_plan(zz(n, 2))
        assert p and p["samples"] * (1 << n) == 8 * p["threads"], (n, p)
    for n, seed in ((10, 3), (11, 20), (12, 21)):
        p = _reg_plan(_gate_soup(n, seed, 4 * n))
        assert p and p["samples"] == 1 and 8 * p["threads"] == (1 << n), (n, p)      # few diagonal terms, many butterflies
    gd = {"single_non_parametric": ["Id", "H", "X", "SX"], "single_parametric": ["RX", "RY", "RZ"], "two_non_parametric": ["CX"], "two_parametric": []}
    genes = genetic.initial_population(1, 13, 8, gd, 13, rng=np.random.default_rng(7))[0]
    g13, _ = genetic.to_quantum(genes, gd, 13, 8, 13, physical_qubits=list(range(13)), verbose=False)
    s = plan(g13)[0]
    assert "pass0 L12" in s and _reg_plan(g13) is not None, s


def test_register_plan_ratio_tables_need_many_register_thread_terms():
    """Terms between a register and a thread qubit become ratio tables from three per register position (full entanglement),
    not for the linear maps, whose stages have one or two; every register position that has such terms gets its pair of
    tables (low / high thread bits; at 13 qubits the high-bits table has two slots), counted in the plan's tables."""
    x12 = ParameterVector("x", 12)
    full = _reg_plan(fm.zz_featuremap(x12, 12, 2, entanglement="full"))
    lin = _reg_plan(zz(12, 2))
    assert lin["ratio_positions"] == 0 and lin["tables"] == 6           # INIT's (A, Blo, Bhi) + ONE set for both stages: the
    # second stage's tables are the first one's in another order (its register qubits were thread qubits there)
    assert _reg_plan(zz(12, 3))["tables"] == 6 and _reg_plan(zz(8, 2))["tables"] == 4
    assert full["ratio_positions"] == 8 and full["tables"] == 9 + 2 * 8
    x13 = ParameterVector("x", 13)
    full13 = _reg_plan(fm.zz_featuremap(x13, 13, 2, entanglement="full"))
    assert full13["ratio_positions"] == 8 and full13["tables"] == 3 * 4 + 3 * 8      # (A, Blo, 2 x Bhi) and (Qlo, 2 x Qhi) per position


def test_register_plan_takes_the_widest_workgroup_for_term_heavy_diagonal_stages():
    """qk_circuit.cpp section 4, packed plans below 10 qubits: a ZZ map with full entanglement leaves 2n or more terms per
    thread at the narrowest width (10), so the planner takes the widest workgroup that fits (12: 4096 amplitudes) -- where the
    ratio tables per register position are affordable -- while the linear map keeps the narrowest."""
    for n, samples in ((7, 32), (8, 16), (9, 8)):
        full = _reg_plan(fm.zz_featuremap(ParameterVector("x", n), n, 2, entanglement="full"))
        assert full and full["samples"] == samples and full["ratio_positions"] > 0, (n, full)
        lin = _reg_plan(zz(n, 2))
        assert lin and lin["samples"] * (1 << n) == 1024, (n, lin)


def test_rz_around_one_gate_is_one_matrix_when_no_two_qubit_diagonal_term_exists():
    """qk_circuit.cpp section 1b: in circuits whose diagonal gates are all single-qubit ones (GA genomes of the hardware-native
    alphabet: rz sx rz between ecr gates) a chain with ONE non-diagonal gate and rz around it becomes a composite matrix
    and leaves no diagonal stage behind; a chain that is the first thing on its qubit and starts with a Hadamard (h p: the
    Z map) keeps gate and phase apart, and so does every chain of a circuit with cz / zz terms."""
    from quake_amd.circuit import QuantumCircuit
    x = ParameterVector("x", 4)

    def counts(c):
        s = plan(c)[0]
        m = re.search(r"composite u1 (\d+), cx (\d+), ecr (\d+), diagonal stages (\d+) with (\d+) terms", s)
        return {"uc": int(m.group(1)), "diag": int(m.group(4)), "terms": int(m.group(5))}, s

    c = QuantumCircuit(4)
    for q in range(4):
        c.sx(q)
    c.ecr(0, 1); c.ecr(2, 3)
    for q in range(4):
        c.rz(x[q], q); c.sx(q); c.rz(x[(q + 1) % 4] * 0.5, q)
    c.ecr(1, 2)
    for q in (1, 2):
        c.sx(q); c.rz(x[q], q)
    got, s = counts(c)
    assert got == {"uc": 6, "diag": 0, "terms": 0}, s
    # the same chains next to a cz: as before (two diagonal stages take the rz terms)
    c2 = QuantumCircuit(4)
    for ins in c.data:
        c2.data.append(ins)
    c2.cz(0, 3)
    got2, s2 = counts(c2)
    assert got2["uc"] == 0 and got2["diag"] >= 2 and got2["terms"] == 11, s2
    # h p as the first thing on a qubit: product state + one diagonal stage (the Z map's shape)
    got3, s3 = counts(fm.z_featuremap(x, 4, 1))
    assert got3 == {"uc": 0, "diag": 1, "terms": 4}, s3
    # ry rz as the first thing on a qubit: the whole run joins the product state
    got4, s4 = counts(fm.custom2_featuremap(x, 4, 1))
    assert got4["diag"] == 0 and "product-state gates 4 (4 composite)" in s4, s4


def test_controls_that_are_still_basis_states_cost_nothing():
    """qk_circuit.cpp section 1: while only id, x and diagonal gates have acted on a qubit it is |0> or |1> up to a phase, so
    a cx it controls is the identity (|0>) or an x on the target (|1>), and a cz with such a |0> qubit is the identity.  GA
    genomes start many qubits that way; without the cx the qubit's single-qubit gates join the product state."""
    from quake_amd.circuit import QuantumCircuit
    x = ParameterVector("x", 3)

    def counts(c):
        s = plan(c)[0]
        m = re.search(r"gates (\d+): product-state gates (\d+) .*?then u1 (\d+), composite u1 (\d+), cx (\d+), ecr (\d+), diagonal stages (\d+) with (\d+) terms", s)
        return {"gates": int(m.group(1)), "init": int(m.group(2)), "cx": int(m.group(5)), "terms": int(m.group(8))}, s

    c = QuantumCircuit(3)
    c.rz(x[0], 0)                 # a phase: q0 is still |0>
    c.cx(0, 1)                    # identity
    c.ry(x[1], 1); c.rx(x[2], 1)  # q1's first gates: product state
    c.x(2); c.cx(2, 0)            # q2 is |1>: an x on q0, which stays a basis state
    c.cx(0, 1)                    # q0 is |1> now: an x on q1
    c.cz(0, 2)                    # neither is |0>: stays
    c.h(0); c.cx(0, 2)            # a real cx
    got, s = counts(c)
    assert got["cx"] == 1 and got["terms"] == 2, s      # rz and cz are the two diagonal terms
    # an ecr whose first qubit is still a basis state is single-qubit gates (and a constant angle program for its phase)
    c3 = QuantumCircuit(3)
    c3.ecr(0, 1); c3.x(2); c3.ecr(2, 0); c3.ecr(1, 2)      # the third one has sx-ed qubit 1 first: it stays
    s3 = plan(c3)[0]
    assert re.search(r"ecr 1, diagonal stages", s3), s3
    # nothing is dropped once the control has seen a non-diagonal gate
    c2 = QuantumCircuit(3)
    c2.h(0); c2.cx(0, 1); c2.sx(2); c2.cx(2, 1); c2.cz(0, 2)
    got2, s2 = counts(c2)
    assert got2["cx"] == 2 and got2["terms"] == 1, s2


def test_adjacent_self_inverse_pairs_cancel():
    """qk_circuit.cpp section 1: cx, cz and ecr are their own inverses -- two of them on the same qubits with nothing but id
    gates (or gates that commute with them) in between on either qubit are dropped together; any other gate in between keeps both."""
    from quake_amd.circuit import QuantumCircuit

    def two_qubit_counts(c):
        s = plan(c)[0]
        m = re.search(r"cx (\d+), ecr (\d+), diagonal stages (\d+) with (\d+) terms", s)
        return (int(m.group(1)), int(m.group(2)), int(m.group(4))), s

    def opened(n=3):      # every qubit has seen a non-diagonal gate: no basis-state shortcuts
        c = QuantumCircuit(n)
        for q in range(n):
            c.sx(q)
        return c

    c = opened(); c.cx(0, 1); c.id(0); c.cx(0, 1); c.ecr(1, 2); c.ecr(1, 2); c.cz(0, 2); c.id(2); c.cz(2, 0)
    got, s = two_qubit_counts(c)
    assert got == (0, 0, 0), s
    c = opened(); c.cx(0, 1); c.sx(0); c.cx(0, 1); c.ecr(1, 2); c.ecr(2, 1); c.cz(0, 2); c.sx(0); c.cz(0, 2); c.cx(0, 1); c.cx(1, 0)
    got, s = two_qubit_counts(c)
    assert got == (4, 2, 2), s
    # gates that commute with the pair may stand between its halves: diagonal gates on a cx's control and on either qubit
    # of a cz, x / sx / rx on a cx's target and on an ecr's second qubit
    x = ParameterVector("x", 2)
    c = opened(); c.cx(0, 1); c.rz(x[0], 0); c.x(1); c.rx(x[1], 1); c.cx(0, 1)
    c.ecr(1, 2); c.sx(2); c.ecr(1, 2)
    c.cz(0, 2); c.rz(x[1], 0); c.p(x[0], 2); c.cz(2, 0)
    got, s = two_qubit_counts(c)
    assert got[0] == 0 and got[1] == 0, s
    c = opened(); c.ecr(0, 1); c.x(0); c.ecr(0, 1); c.cx(1, 2); c.rz(x[0], 2); c.cx(1, 2)      # x on an ecr's first qubit, rz on a cx's target: both pairs stay
    got, s = two_qubit_counts(c)
    assert got[0] == 2 and got[1] == 2, s
    c = opened(); c.cx(0, 1); c.cx(0, 1); c.cx(0, 1)      # three in a row: one is left
    got, s = two_qubit_counts(c)
    assert got == (1, 0, 0), s
