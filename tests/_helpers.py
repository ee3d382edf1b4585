"""Test-side glue between the product's circuit IR and the CPU oracle."""
import numpy as np

from oracle import qk_oracle as O
from quake_amd import circuit as qc
from quake_amd import featuremaps as fm

_NAMES = {v: k for k, v in qc.GATE_CODES.items()}


def run_postfix(prog, consts, X):
    """Evaluate one postfix angle program for every row of X (numpy, float64)."""
    st = []
    for op, arg in prog:
        if op == qc.EX_CONST:
            st.append(np.full(X.shape[0], consts[arg]))
        elif op == qc.EX_PARAM:
            st.append(X[:, arg].astype(np.float64))
        elif op == qc.EX_NEG:
            st.append(-st.pop())
        elif op == qc.EX_ASIN:
            st.append(np.arcsin(st.pop()))
        elif op == qc.EX_ACOS:
            st.append(np.arccos(st.pop()))
        elif op == qc.EX_SIN:
            st.append(np.sin(st.pop()))
        elif op == qc.EX_COS:
            st.append(np.cos(st.pop()))
        else:
            b, a = st.pop(), st.pop()
            st.append({qc.EX_ADD: a + b, qc.EX_SUB: a - b, qc.EX_MUL: a * b, qc.EX_DIV: a / b}[op])
    assert len(st) == 1
    return st[0]


def compiled_to_oracle_ops(compiled):
    """CompiledCircuit -> oracle op tuples whose angle functions read DATA COLUMNS
    (the binding is already resolved in the compiled program)."""
    ops = []
    for gate, q0, q1, e in compiled.ops:
        name = _NAMES[int(gate)]
        qubits = (int(q0),) if q1 < 0 else (int(q0), int(q1))
        fn = None
        if e >= 0:
            prog = [tuple(map(int, r)) for r in compiled.prog[compiled.expr_off[e]:compiled.expr_off[e + 1]]]
            fn = (lambda X, prog=prog: run_postfix(prog, compiled.consts, X))
        ops.append((name, qubits, fn, ()))
    return ops


def psi_from_compiled(circ, X):
    """Statevectors of a product circuit, simulated by the oracle's numpy simulator."""
    compiled = circ.compile()
    X = np.asarray(X, dtype=np.float64)
    assert X.shape[1] == compiled.n_params
    return O.simulate(compiled_to_oracle_ops(compiled), compiled.n_qubits, X)


# (product builder, oracle builder) pairs for every hand-written map
def map_pairs(n, repeats):
    x = qc.ParameterVector("x", n)
    pairs = {
        "z": (fm.z_featuremap(x, n, repeats), O.z_featuremap(n, repeats), n),
        "zz": (fm.zz_featuremap(x, n, repeats), O.zz_featuremap(n, repeats), n),
        "zz_full": (fm.zz_featuremap(x, n, repeats, entanglement="full"), O.zz_featuremap(n, repeats, "full"), n),
        "custom1": (fm.custom1_featuremap(x, n, repeats), O.custom1_featuremap(n, repeats), n),
        "custom2": (fm.custom2_featuremap(x, n, repeats), O.custom2_featuremap(n, repeats), n),
    }
    return pairs


def two_qubit_pairs(repeats):
    x = qc.ParameterVector("x", 2)
    return {
        "genetic": (fm.genetic_featuremap(x, repeats), O.genetic_featuremap(repeats), 2),
        "genetic_2": (fm.genetic_featuremap_2(x, repeats), O.genetic_featuremap_2(repeats), 2),
        "genetic_highres": (fm.genetic_featuremap_highres(x, repeats), O.genetic_featuremap_highres(repeats), 2),
        "genetic_highres_2": (fm.genetic_featuremap_highres_2(x, repeats), O.genetic_featuremap_highres_2(repeats), 2),
        "genetic_attention": (fm.genetic_attention(x, repeats), O.genetic_attention(repeats), 2),
        "genetic_attention_highres": (fm.genetic_attention_highres(x, repeats), O.genetic_attention_highres(repeats), 2),
    }


def synth_X(N, F, kind, seed):
    """Synthetic inputs of SURVEY.md 8d: 'scaled' = range rearrange_scale produces,
    'unit' = open (0, 1) as after MinMaxScaler + outlier drop."""
    rng = np.random.default_rng(seed)
    if kind == "scaled":
        lo, hi = -np.pi / 2 + 1e-5, np.pi / 2 - 1e-5
        return rng.uniform(lo, hi, size=(N, F))
    return rng.uniform(1e-6, 1 - 1e-6, size=(N, F))
