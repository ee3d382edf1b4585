"""Genome -> feature-map decoder and the genetic-run helpers around the kernel.

Restates the integer-gene decoder of the reference (``genetic_v3.to_quantum`` at
``src/quake/models/qsvm/genetic_v3.py:326-448`` and its coupling-map aware
variants ``genetic_v4.py:221-347`` / ``genetic_v5.py:221-344``) on the build's
circuit IR, plus the cheap host-side pieces the fitness function needs
(population initialisation, gene space, kernel statistics).  The Gram matrix
itself comes from :mod:`quake_amd.kernels` (HIP).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .circuit import ParameterVector, QuantumCircuit


def _gate_list(gate_dict: dict) -> list:
    # reference genetic_v3.py:359-361: concatenation in dict insertion order;
    # duplicated names (["RZ", "RZ"]) act as sampling weights.
    out: list = []
    for gate_set in gate_dict.values():
        out = out + list(gate_set)
    return out


def _angle(x, transform: int, multi: int, f1: int, f2: int):
    """The six feature->angle encodings (reference genetic_v3.py:395-442)."""
    if multi == 0 and transform == 0:
        return 2 * np.pi * (x[f1] - 0.5)
    if multi == 1 and transform == 0:
        return 2 * np.pi * x[f1] * (1 - x[f2]) - np.pi
    if multi == 0 and transform == 1:
        return 2 * np.pi * x[f1] * (1 - x[f1]) - np.pi
    if multi == 1 and transform == 1:
        return (2 * np.pi * x[f1] * (1 - x[f2]) - np.pi) * (2 * np.pi * x[f2] * (1 - x[f1]) - np.pi) / np.pi
    if multi == 0 and transform == 2:
        return 2 * np.arcsin(2 * x[f1] - 1) - np.pi
    if multi == 1 and transform == 2:
        return 2 * np.arcsin((2 * x[f1] - 1) * (2 * x[f2] - 1))
    # the reference leaves `param_expression` unbound here and raises
    # UnboundLocalError at the first such gene; say what is wrong instead.
    raise ValueError(f"invalid gene: feature_transformation={transform}, multi_features={multi}")


def to_quantum(
    genes: Sequence[int],
    gate_dict: dict,
    nb_features: int,
    gates_per_qubits: int,
    nb_qubits: int,
    param_shift: int = 0,
    physical_qubits: Sequence[int] = (0, 1, 2, 3),
    coupling_map: Optional[Sequence] = None,
    variant: str = "v3",
    verbose: bool = True,
) -> Tuple[QuantumCircuit, List[int]]:
    """Convert an integer gene sequence to a feature map.

    Same arguments and return value as the reference: ``(fmap, x_idxs)`` where
    ``x_idxs`` lists the feature indices in first-use order; callers then evaluate
    ``kernel.evaluate(x_vec=data[:, x_idxs])`` (reference genetic_v3.py:604-619).

    ``variant`` picks the two-qubit placement rule:

    * ``"v3"`` (reference genetic_v3.py:381-390): qubit k in {0, 1} -> ``gate(k, k+1)``,
      k == 3 -> ``gate(3, 2)``, any other k emits nothing (and says so);
    * ``"v4"`` (reference genetic_v4.py:278-289): as v3, and only if the tuple
      ``(physical_qubits[k], physical_qubits[k +/- 1])`` is in ``coupling_map``;
    * ``"v5"`` (reference genetic_v5.py:278-286): k < 3 -> ``gate(k, k+1)``,
      k == 3 -> ``gate(3, 2)``, each only if the pair (as list or tuple) is in
      ``coupling_map``; other k emit nothing.
    """
    if variant not in ("v3", "v4", "v5"):
        raise ValueError("variant must be 'v3', 'v4' or 'v5'")
    if variant != "v3" and coupling_map is None:
        # reference: `x in None` raises TypeError at the first two-qubit gene
        raise TypeError("argument of type 'NoneType' is not iterable (coupling_map is required)")
    gate_list = _gate_list(gate_dict)
    genes_unflatted = np.reshape(np.asarray(genes), [gates_per_qubits, nb_qubits, 5])
    x = ParameterVector("x", length=nb_features + param_shift)
    fmap = QuantumCircuit(nb_qubits)
    x_idxs: List[int] = []
    phys = list(physical_qubits)

    def coupled(a: int, b: int) -> bool:
        if variant == "v3":
            return True
        pa, pb = phys[a], phys[b]
        if variant == "v4":
            return (pa, pb) in coupling_map
        return [pa, pb] in coupling_map or (pa, pb) in coupling_map

    for j in range(gates_per_qubits):
        for k in range(nb_qubits):
            gate_type_idx, transform, multi, f1, f2 = (int(v) for v in genes_unflatted[j, k])
            f1 += param_shift
            f2 += param_shift
            name = gate_list[gate_type_idx]
            gate = getattr(fmap, name.lower())
            if name in gate_dict["single_non_parametric"]:
                if name != "I":
                    gate(k)
            elif name in gate_dict["two_non_parametric"]:
                did_something = False
                if k == 0 or k == 1 or (variant == "v5" and k == 2):
                    if coupled(k, k + 1):
                        gate(k, k + 1)
                        did_something = True
                elif k == 3:
                    if coupled(k, k - 1):
                        gate(k, k - 1)
                        did_something = True
                if not did_something and variant != "v5":
                    # reference genetic_v3.py:389-390 indexes physical_qubits[k] for its
                    # message, so a two-qubit gene on qubit k >= len(physical_qubits)
                    # raises IndexError there; keep that (pass a longer list to allow it).
                    who = phys[k]
                    if verbose:
                        print(phys, "for qubit", who, "I did nothing")
            else:
                if f1 not in x_idxs:
                    x_idxs.append(f1)
                if multi == 1 and f2 not in x_idxs:
                    x_idxs.append(f2)
                param_expression = _angle(x, transform, multi, f1, f2)
                if name in gate_dict["single_parametric"]:
                    gate(param_expression, k)
                # two_parametric gates: the reference's emitting branch is commented
                # out (genetic_v3.py:446-447); the gene only lands in x_idxs.
    return fmap, x_idxs


# --------------------------------------------------------------------------- #
# the genetic run's input range (reference genetic_v3.py:24-62)
# --------------------------------------------------------------------------- #
def get_subsample(dataset: np.ndarray, labels: np.ndarray, size: int, seed: int = 42, scaler=None):
    """The GA's cross-validation set: scale with a FITTED ``MinMaxScaler``, drop every row that touches or
    leaves the scaler's ``feature_range`` in any column (strictly: a value equal to a bound is an outlier
    too), then a seeded ``train_test_split`` subsample of ``size`` rows (reference ``genetic_v3.py:24-62``).

    With ``MinMaxScaler((0, 1))`` this is what puts the genome decoder's inputs in the OPEN interval
    (0, 1), so every ``asin`` argument of :func:`_angle` stays inside (-1, 1).  ``scaler``
    is required, as in the reference (which dereferences it unconditionally)."""
    from sklearn.model_selection import train_test_split

    if scaler is None:
        raise AttributeError("'NoneType' object has no attribute 'feature_range' (pass a fitted MinMaxScaler)")
    lo, hi = scaler.feature_range
    scaled = scaler.transform(dataset)
    inside = ~((scaled >= hi) | (scaled <= lo)).any(axis=1)
    picked = train_test_split(scaled[inside], np.asarray(labels)[inside], train_size=size, random_state=seed)
    return picked[0], picked[2]


# --------------------------------------------------------------------------- #
# population helpers (reference genetic_v3.py:65-212)
# --------------------------------------------------------------------------- #
def _nb_possible_gates(gate_dict: dict) -> int:
    return sum(len(gate_dict[k]) for k in
               ("single_non_parametric", "single_parametric", "two_non_parametric", "two_parametric"))


def gen_int(min_val: int, max_val: int, size: int = None, exclude_array=None, rng=None):
    """Integer sequence in [min_val, max_val), optionally excluding one value per
    position (reference genetic_v3.py:181-212).  ``rng=None`` uses numpy's global
    legacy stream like the reference; pass a ``np.random.Generator`` for a seeded,
    version-stable stream."""
    if exclude_array is not None:
        out = []
        for i in range(size):
            valid = np.setdiff1d(np.arange(min_val, max_val), [exclude_array[i]])
            out.append((rng.choice(valid) if rng is not None else np.random.choice(valid)))
        return out
    if rng is not None:
        return rng.integers(min_val, max_val, size)
    return np.random.randint(min_val, max_val, size)


def initial_population(nb_init_individuals: int, nb_qubits: int, gates_per_qubits: int,
                       gate_dict: dict, nb_features: int, rng=None) -> np.ndarray:
    """Generation-0 chromosomes, shape ``[individuals, gates_per_qubits*nb_qubits*5]``
    (reference genetic_v3.py:65-128; gene order: gate, transform in {0,1,2},
    multi in {0,1}, first feature, second feature != first)."""
    size = nb_qubits * gates_per_qubits * nb_init_individuals
    gate_idxs = gen_int(0, _nb_possible_gates(gate_dict), size=size, rng=rng)
    feature_transformation = gen_int(0, 3, size=size, rng=rng)
    multi_features = gen_int(0, 2, size=size, rng=rng)
    first_feature_idx = gen_int(0, nb_features, size=size, rng=rng)
    second_feature_idx = gen_int(0, nb_features, size=size, exclude_array=first_feature_idx, rng=rng)
    gene_array = np.array([gate_idxs, feature_transformation, multi_features,
                           first_feature_idx, second_feature_idx])
    return np.reshape(gene_array.T, [nb_init_individuals, gates_per_qubits, nb_qubits, 5]).reshape(
        nb_init_individuals, -1)


def get_gene_space(gate_dict: dict, nb_features: int, nb_qubits: int, gates_per_qubits: int) -> list:
    """Per-gene value ranges for mutation (reference genetic_v3.py:140-178; note the
    transform range is {0, 1} here although generation 0 draws {0, 1, 2})."""
    space: list = []
    for _ in range(nb_qubits * gates_per_qubits):
        space += [range(_nb_possible_gates(gate_dict)), range(2), range(2),
                  range(nb_features), range(nb_features)]
    return space


# --------------------------------------------------------------------------- #
# kernel statistics used by the fitness function (reference genetic_v3.py:643-660)
# --------------------------------------------------------------------------- #
def kernel_statistics(qker_matrix: np.ndarray, data_labels: np.ndarray) -> dict:
    """``sparsity``, ``offdiagonal_mean`` and ``offdiagonal_std`` exactly as the
    reference computes them -- including the mean/std over the *whole* N x N array
    of ``np.triu(K, 1)`` (zeros of the lower half included)."""
    k0 = qker_matrix[data_labels == 0]
    k0 = np.triu(k0[:, data_labels == 0], 1)
    a0 = k0[np.triu_indices(k0.shape[0], 1)]
    k1 = qker_matrix[data_labels == 1]
    k1 = np.triu(k1[:, data_labels == 1], 1)
    a1 = k1[np.triu_indices(k1.shape[0], 1)]
    k01 = qker_matrix[data_labels == 0][:, data_labels == 1]
    upper = np.triu(qker_matrix, 1)
    return {
        "sparsity": (np.mean(a0) + np.mean(a1)) / 2 - np.mean(k01),
        "offdiagonal_mean": np.mean(upper),
        "offdiagonal_std": np.std(upper),
    }


def add_symmetric_gaussian_noise(qker_matrix: np.ndarray, noise_std: float, rng=None) -> np.ndarray:
    """Symmetric Gaussian noise on the off-diagonal, clipped to [0, 1]
    (reference genetic_v3.py:620-627)."""
    draw = rng.normal if rng is not None else np.random.normal
    noise = np.triu(draw(0, noise_std, size=qker_matrix.shape), k=1)
    return np.clip(qker_matrix + noise + noise.T, 0, 1)


def fitness(solution, data_cv, data_labels, gate_dict, nb_features, gates_per_qubits, nb_qubits,
            fit_fun: Callable, projected: bool = False, noise_std=None, depth_fn: Optional[Callable] = None,
            cv: int = 4, rng=None) -> Tuple[float, dict]:
    """One genome's fitness, the ``with_kernel`` branch of the reference's
    ``fitness_func`` (genetic_v3.py:604-661) with the Gram matrix computed on the GPU.

    ``depth_fn(fmap)`` stands in for ``transpile(...).depth()`` (Qiskit transpiler,
    not part of this build); default is the untranspiled circuit depth."""
    from sklearn.model_selection import cross_val_score
    from sklearn.svm import SVC

    from .kernels import HipFidelityKernel, projected_quantum_kernel

    fmap, x_idxs = to_quantum(solution, gate_dict, nb_features, gates_per_qubits, nb_qubits, verbose=False)
    if projected:
        K = projected_quantum_kernel(fmap, data_cv[:, x_idxs], 1)
    else:
        K = HipFidelityKernel(feature_map=fmap).evaluate(x_vec=data_cv[:, x_idxs])
        if noise_std is not None:
            K = add_symmetric_gaussian_noise(K, noise_std, rng)
    acc = cross_val_score(SVC(kernel="precomputed"), K, data_labels, cv=cv, scoring="accuracy").mean()
    stats = kernel_statistics(K, data_labels)
    depth = depth_fn(fmap) if depth_fn is not None else fmap.depth()
    stats.update(accuracy=acc, depth=depth)
    return fit_fun(acc, stats["offdiagonal_std"], depth), stats
