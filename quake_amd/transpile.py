"""A small native basis-gate depth counter.

The reference scores a genome's circuit with ``transpile(fmap, coupling_map=coupling_map,
basis_gates=basis_gates).depth()`` (reference genetic_v3.py:652-654; the shipped example passes
``None, None``, genetic_v4 passes an IBM backend's coupling map and basis).  The Qiskit transpiler
is a third-party, unpinned dependency that is absent here, and its output depends on its version
and optimisation level, so this module does not try to reproduce it pass by pass: it is a
deterministic stand-in that lets ``fit_fun(accuracy, density, depth)`` run without Qiskit --
**parity unpinned** (no reference-owned depth value exists to pin it).

What it does, in order:
1. drop ``id`` gates and barriers;
2. with ``basis_gates``: rewrite every gate into the basis with textbook identities
   (``h = rz(pi/2) sx rz(pi/2)``, ``rx/ry`` = 2 ``sx`` + 3 ``rz``, ``p -> rz``, ``cz = h cx h``,
   ``ecr <-> cx`` up to single-qubit gates, ...), counting single-qubit rotations about z as one
   ``rz`` and merging neighbouring ``rz`` on the same qubit;
3. with ``coupling_map``: a two-qubit gate on uncoupled qubits is routed by SWAPs along a shortest
   path (3 two-qubit gates each) -- no re-layout;
4. depth = longest path through the resulting gate list.
"""
from __future__ import annotations

from collections import deque
from typing import Iterable, List, Optional, Sequence, Tuple

from .circuit import QuantumCircuit

Gate = Tuple[str, Tuple[int, ...]]

_ONE_Q_COST = {
    # gate -> sequence in the {rz, sx, x} basis (angles do not matter for depth)
    "h": ["rz", "sx", "rz"],
    "x": ["x"],
    "sx": ["sx"],
    "rz": ["rz"],
    "p": ["rz"],
    "rx": ["rz", "sx", "rz", "sx", "rz"],
    "ry": ["sx", "rz", "sx"],          # up to the rz's that merge into neighbours: sx rz(theta+pi) sx rz(pi)
}


def _lower_1q(name: str, q: int, basis: set) -> List[Gate]:
    if name in basis:
        return [(name, (q,))]
    if name == "p" and "rz" in basis:
        return [("rz", (q,))]
    if name == "x" and "x" not in basis and "sx" in basis:
        return [("sx", (q,)), ("sx", (q,))]
    seq = _ONE_Q_COST.get(name)
    if seq is None:
        raise NotImplementedError(f"no decomposition of '{name}' into {sorted(basis)}")
    if name == "ry":
        seq = ["sx", "rz", "sx", "rz"]
    out = []
    for g in seq:
        if g not in basis:
            if g == "x" and "sx" in basis:
                out += [("sx", (q,)), ("sx", (q,))]
                continue
            raise NotImplementedError(f"basis {sorted(basis)} lacks '{g}' needed for '{name}'")
        out.append((g, (q,)))
    return out


def _lower_2q(name: str, a: int, b: int, basis: set) -> List[Gate]:
    if name in basis:
        return [(name, (a, b))]
    h = lambda q: _lower_1q("h", q, basis)
    if name == "cz":
        if "cx" in basis:
            return h(b) + [("cx", (a, b))] + h(b)
        if "ecr" in basis:
            return h(b) + _lower_2q("cx", a, b, basis) + h(b)
    if name == "cx":
        if "cz" in basis:
            return h(b) + [("cz", (a, b))] + h(b)
        if "ecr" in basis:   # cx(a,b) = [rz(-pi/2) on a, then ecr, with x / sx corrections] : 1 ecr + 1q gates
            return _lower_1q("rz", a, basis) + _lower_1q("sx", b, basis) + [("ecr", (a, b))] + _lower_1q("x", a, basis)
    if name == "ecr":
        if "cx" in basis:    # ecr = cx up to rz / sx / x on the two qubits
            return _lower_1q("rz", a, basis) + _lower_1q("sx", b, basis) + [("cx", (a, b))] + _lower_1q("x", a, basis)
        if "cz" in basis:
            return (_lower_1q("rz", a, basis) + _lower_1q("sx", b, basis) + _lower_2q("cx", a, b, basis) +
                    _lower_1q("x", a, basis))
    raise NotImplementedError(f"no decomposition of '{name}' into {sorted(basis)}")


def _shortest_path(adj: dict, a: int, b: int) -> Optional[List[int]]:
    prev = {a: None}
    dq = deque([a])
    while dq:
        u = dq.popleft()
        if u == b:
            path = []
            while u is not None:
                path.append(u)
                u = prev[u]
            return path[::-1]
        for v in sorted(adj.get(u, ())):
            if v not in prev:
                prev[v] = u
                dq.append(v)
    return None


def lower(circ: QuantumCircuit, coupling_map: Optional[Sequence[Sequence[int]]] = None,
          basis_gates: Optional[Iterable[str]] = None) -> List[Gate]:
    """The gate list after steps 1-3 of the module docstring."""
    basis = set(g.lower() for g in basis_gates) if basis_gates is not None else None
    adj: dict = {}
    if coupling_map is not None:
        for a, b in coupling_map:
            adj.setdefault(int(a), set()).add(int(b))
            adj.setdefault(int(b), set()).add(int(a))      # direction is fixed with single-qubit gates: same depth class
    gates: List[Gate] = []

    def emit2(name, a, b):
        gates.extend(_lower_2q(name, a, b, basis) if basis is not None else [(name, (a, b))])

    for ins in circ.data:
        if ins.name == "id":
            continue
        if len(ins.qubits) == 1:
            gates.extend(_lower_1q(ins.name, ins.qubits[0], basis) if basis is not None else [(ins.name, ins.qubits)])
            continue
        a, b = ins.qubits
        if coupling_map is not None and b not in adj.get(a, ()):
            path = _shortest_path(adj, a, b)
            if path is None:
                raise ValueError(f"qubits {a} and {b} are not connected in the coupling map")
            swap2q = "cx" if (basis is None or "cx" in basis) else ("ecr" if "ecr" in basis else "cz")
            hops = list(zip(path[:-2], path[1:-1]))          # bring a next to b, act, and undo
            for u, v in hops:
                for k in range(3):
                    emit2(swap2q, *((u, v) if k != 1 else (v, u)))
            emit2(ins.name, path[-2], b)
            for u, v in reversed(hops):
                for k in range(3):
                    emit2(swap2q, *((u, v) if k != 1 else (v, u)))
        else:
            emit2(ins.name, a, b)
    # merge neighbouring rz on the same qubit
    merged: List[Gate] = []
    last_on = {}
    for g in gates:
        name, qs = g
        if name == "rz" and last_on.get(qs[0]) is not None and merged[last_on[qs[0]]][0] == "rz":
            continue
        merged.append(g)
        for q in qs:
            last_on[q] = len(merged) - 1
    return merged


def depth_of(gates: Sequence[Gate], num_qubits: int) -> int:
    level = [0] * num_qubits
    for _, qs in gates:
        d = 1 + max(level[q] for q in qs)
        for q in qs:
            level[q] = d
    return max(level) if level else 0


def transpiled_depth(circ: QuantumCircuit, coupling_map=None, basis_gates=None) -> int:
    """Stand-in for ``transpile(fmap, coupling_map=..., basis_gates=...).depth()`` (see module docstring)."""
    n = circ.num_qubits
    if coupling_map is not None:
        n = max([n] + [int(q) + 1 for pair in coupling_map for q in pair])
    return depth_of(lower(circ, coupling_map, basis_gates), n)


class LoweredCircuit:
    """What ``transpile(fmap, coupling_map, basis_gates, optimization_level=0)`` hands a ``without_kernel``
    fitness function (reference genetic_v3.py:712; the shipped example's ``fit_fun(fmap)`` reads
    ``fmap.count_ops()``, examples/genetic_featuremap_v3.py:77-85): the lowered, routed gate list with the
    introspection calls a fitness function makes on a circuit.  Angles are not carried -- lowering keeps
    gate kinds and positions only (module docstring) -- so this object cannot be simulated."""

    def __init__(self, gates: Sequence[Gate], num_qubits: int):
        self.data: List[Gate] = list(gates)
        self.num_qubits = int(num_qubits)

    def count_ops(self) -> dict:
        counts: dict = {}
        for name, _ in self.data:
            counts[name] = counts.get(name, 0) + 1
        return dict(sorted(counts.items(), key=lambda kv: -kv[1]))      # most frequent first, as qiskit orders it

    def depth(self) -> int:
        return depth_of(self.data, self.num_qubits)

    def size(self) -> int:
        return len(self.data)

    def num_nonlocal_gates(self) -> int:
        return sum(1 for _, qs in self.data if len(qs) > 1)

    def __len__(self) -> int:
        return len(self.data)


def transpile(circ: QuantumCircuit, coupling_map=None, basis_gates=None, optimization_level: int = 0) -> LoweredCircuit:
    """Stand-in for ``qiskit.transpile`` at the one place the reference passes its RESULT on rather than
    taking ``.depth()`` of it (genetic_v3.py:712).  ``optimization_level`` is accepted for signature
    compatibility; the lowering is the one described in the module docstring at every level."""
    n = circ.num_qubits
    if coupling_map is not None:
        n = max([n] + [int(q) + 1 for pair in coupling_map for q in pair])
    return LoweredCircuit(lower(circ, coupling_map, basis_gates), n)
