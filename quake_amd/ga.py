"""Genetic search over feature-map genomes with every generation evaluated as one GPU batch.

The reference drives the search with ``pygad.GA`` (``genetic_instance``, reference
genetic_v3.py:451-533; options at examples/genetic_featuremap_v3.py:108-123) and evaluates one
chromosome per ``fitness_func`` call (genetic_v3.py:587-726).  pygad is not part of this build;
:class:`GeneticRun` restates the subset of its documented behaviour the reference configures --
steady-state / rank / tournament / roulette-wheel parent selection, single-point / two-points /
uniform crossover with ``crossover_probability``, random mutation inside ``gene_space`` with
``mutation_percent_genes`` / ``mutation_num_genes``, ``keep_parents`` / ``keep_elitism``,
``stop_criteria`` (``saturate_N``, ``reach_X``), ``on_generation`` -- with the attributes the
reference reads (``generations_completed``, ``last_generation_fitness``,
``best_solutions_fitness``, ``best_solution()``).  The random stream is this module's own
(``numpy.random.Generator``; the reference uses numpy's unseeded global stream inside pygad), so
runs are reproducible here but not comparable draw by draw: **parity unpinned** for the search
trajectory; the per-chromosome fitness is the pinned quantity (tests/test_gpu_qsvm.py).

What the GPU changes: all chromosomes of a generation whose fitness is not known yet go through
``kernels.evaluate_population`` -- one C-ABI call that simulates every circuit, forms every Gram
matrix, applies the noise model and reduces the kernel statistics on the device.
"""
from __future__ import annotations

import csv
import inspect
import time
from pathlib import Path
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import genetic


class GeneticRun:
    """A small generational GA with pygad's vocabulary.

    ``fitness_batch(run, solutions, indices) -> sequence of float`` is called once per
    generation with the chromosomes whose fitness is unknown (kept parents / elites and -- with
    ``save_solutions`` -- chromosomes seen before are not re-evaluated).  Alternatively pass
    pygad's per-solution ``fitness_func(run, solution, solution_idx)``.
    """

    def __init__(self, *, initial_population, num_generations: int, num_parents_mating: int,
                 fitness_batch: Optional[Callable] = None, fitness_func: Optional[Callable] = None,
                 gene_space: Optional[Sequence] = None, gene_type=int,
                 parent_selection_type: str = "sss", keep_parents: int = -1, keep_elitism: int = 1,
                 K_tournament: int = 3, crossover_type: Optional[str] = "single_point",
                 crossover_probability: Optional[float] = None, mutation_type: Optional[str] = "random",
                 mutation_probability: Optional[float] = None, mutation_by_replacement: bool = False,
                 mutation_percent_genes="default", mutation_num_genes: Optional[int] = None,
                 random_mutation_min_val: float = -1.0, random_mutation_max_val: float = 1.0,
                 allow_duplicate_genes: bool = True, stop_criteria=None, on_generation: Optional[Callable] = None,
                 save_solutions: bool = False, suppress_warnings: bool = True, random_seed: Optional[int] = None):
        if (fitness_batch is None) == (fitness_func is None):
            raise ValueError("pass exactly one of fitness_batch and fitness_func")
        self.population = np.array(initial_population, dtype=gene_type).copy()
        if self.population.ndim != 2 or self.population.shape[0] < 2:
            raise ValueError("initial_population must be [individuals >= 2, genes]")
        self.sol_per_pop, self.num_genes = self.population.shape
        self.num_generations = int(num_generations)
        self.num_parents_mating = int(num_parents_mating)
        if not 0 < self.num_parents_mating <= self.sol_per_pop:
            raise ValueError("num_parents_mating must be in 1..sol_per_pop")
        if parent_selection_type not in ("sss", "rank", "tournament", "rws", "random"):
            raise NotImplementedError(f"parent_selection_type {parent_selection_type!r} (multi-objective NSGA-II "
                                      "selection is not part of this build)")
        if crossover_type not in (None, "single_point", "two_points", "uniform"):
            raise NotImplementedError(f"crossover_type {crossover_type!r}")
        if mutation_type not in (None, "random"):
            raise NotImplementedError(f"mutation_type {mutation_type!r}")
        if gene_space is not None and len(gene_space) != self.num_genes:
            raise ValueError("gene_space needs one entry per gene")
        if keep_elitism < 0 or keep_elitism > self.sol_per_pop:
            raise ValueError("keep_elitism must be in 0..sol_per_pop")
        if keep_parents < -1 or keep_parents > self.num_parents_mating:
            raise ValueError("keep_parents must be -1, 0 or at most num_parents_mating")
        self.fitness_batch, self.fitness_func = fitness_batch, fitness_func
        self.gene_space = [np.asarray(list(s)) for s in gene_space] if gene_space is not None else None
        self.gene_type = gene_type
        self.parent_selection_type = parent_selection_type
        self.keep_parents, self.keep_elitism, self.K_tournament = keep_parents, keep_elitism, K_tournament
        self.crossover_type, self.crossover_probability = crossover_type, crossover_probability
        self.mutation_type, self.mutation_probability = mutation_type, mutation_probability
        self.mutation_by_replacement = mutation_by_replacement
        self.random_mutation_min_val, self.random_mutation_max_val = random_mutation_min_val, random_mutation_max_val
        self.allow_duplicate_genes = allow_duplicate_genes
        # pygad: percent -> count, at least one gene
        if mutation_num_genes is None:
            pct = 10.0 if isinstance(mutation_percent_genes, str) else float(mutation_percent_genes)
            mutation_num_genes = int(pct * self.num_genes / 100.0)
        self.mutation_num_genes = max(1, min(int(mutation_num_genes), self.num_genes))
        self.stop_criteria = self._parse_stop(stop_criteria)
        self.on_generation = on_generation
        self.save_solutions = save_solutions
        self.rng = np.random.default_rng(random_seed)
        self.generations_completed = 0
        self.last_generation_fitness: Optional[np.ndarray] = None
        self.previous_generation_fitness: Optional[np.ndarray] = None
        self.best_solutions_fitness: List[float] = []
        self.solutions: List[list] = []
        self.solutions_fitness: List[float] = []
        self._known: dict = {}            # chromosome bytes -> fitness (save_solutions / kept individuals)
        self.evaluations = 0              # chromosomes actually sent to the fitness function
        self.run_completed = False

    # ------------------------------------------------------------------ fitness
    @staticmethod
    def _parse_stop(stop):
        if stop is None:
            return []
        out = []
        for item in ([stop] if isinstance(stop, str) else list(stop)):
            word, _, num = item.partition("_")
            if word not in ("reach", "saturate") or not num:
                raise ValueError(f"stop_criteria {item!r}: expected 'reach_<fitness>' or 'saturate_<generations>'")
            out.append((word, float(num)))
        return out

    def _cal_pop_fitness(self, carried: dict) -> np.ndarray:
        """Fitness of the current population; ``carried`` maps population index -> fitness already
        known (kept parents / elites)."""
        fit = np.full(self.sol_per_pop, np.nan)
        todo = []
        for i in range(self.sol_per_pop):
            key = self.population[i].tobytes()
            if i in carried:
                fit[i] = carried[i]
            elif self.save_solutions and key in self._known:
                fit[i] = self._known[key]
            else:
                todo.append(i)
        if todo:
            # identical chromosomes inside one generation are evaluated once
            first = {}
            for i in todo:
                first.setdefault(self.population[i].tobytes(), i)
            uniq = sorted(first.values())
            if self.fitness_batch is not None:
                vals = list(self.fitness_batch(self, self.population[uniq].copy(), list(uniq)))
                if len(vals) != len(uniq):
                    raise ValueError("fitness_batch must return one value per solution")
            else:
                vals = [self.fitness_func(self, self.population[i].copy(), i) for i in uniq]
            for v in vals:
                if np.ndim(v) != 0:
                    raise NotImplementedError("multi-objective fitness (a list per chromosome) is not part of this build")
            self.evaluations += len(uniq)
            by_key = {self.population[i].tobytes(): float(v) for i, v in zip(uniq, vals)}
            for i in todo:
                fit[i] = by_key[self.population[i].tobytes()]
        for i in range(self.sol_per_pop):
            self._known[self.population[i].tobytes()] = fit[i]
        if self.save_solutions:
            self.solutions.extend(self.population.tolist())
            self.solutions_fitness.extend(fit.tolist())
        return fit

    # ------------------------------------------------------------------ operators
    def _select_parents(self, fitness: np.ndarray, num: int):
        order = np.argsort(-fitness, kind="stable")
        kind = self.parent_selection_type
        if kind == "sss":
            idx = order[:num]
        elif kind == "random":
            idx = self.rng.integers(0, self.sol_per_pop, size=num)
        elif kind == "tournament":
            idx = np.empty(num, dtype=int)
            for k in range(num):
                cand = self.rng.integers(0, self.sol_per_pop, size=self.K_tournament)
                idx[k] = cand[np.argmax(fitness[cand])]
        else:
            if kind == "rank":       # probability proportional to rank (worst = 1)
                w = np.empty(self.sol_per_pop)
                w[order] = np.arange(self.sol_per_pop, 0, -1)
            else:                    # rws: proportional to fitness (shifted to be non-negative)
                w = fitness - min(fitness.min(), 0.0)
                if not np.any(w > 0):
                    w = np.ones_like(w)
            idx = self.rng.choice(self.sol_per_pop, size=num, p=w / w.sum())
        return self.population[idx].copy(), np.asarray(idx)

    def _crossover(self, parents: np.ndarray, n_off: int) -> np.ndarray:
        P, L = parents.shape
        off = np.empty((n_off, L), dtype=self.population.dtype)
        for k in range(n_off):
            if self.crossover_probability is not None:
                cand = np.where(self.rng.random(P) <= self.crossover_probability)[0]
                if len(cand) == 0:
                    off[k] = parents[k % P]
                    continue
                if len(cand) == 1:
                    off[k] = parents[cand[0]]
                    continue
                i1, i2 = self.rng.choice(cand, size=2, replace=False)
            else:
                i1, i2 = k % P, (k + 1) % P
            a, b = parents[i1], parents[i2]
            if self.crossover_type is None:
                off[k] = a
            elif self.crossover_type == "single_point":
                pt = int(self.rng.integers(0, L))
                off[k] = np.concatenate([a[:pt], b[pt:]])
            elif self.crossover_type == "two_points":
                if L == 1:
                    off[k] = a
                    continue
                pt1 = int(self.rng.integers(0, int(np.ceil(L / 2 + 1))))
                pt2 = pt1 + L // 2
                off[k] = np.concatenate([a[:pt1], b[pt1:pt2], a[pt2:]])
            else:  # uniform
                take = self.rng.integers(0, 2, size=L).astype(bool)
                off[k] = np.where(take, b, a)
        return off

    def _mutate(self, off: np.ndarray) -> np.ndarray:
        if self.mutation_type is None:
            return off
        L = self.num_genes
        for k in range(off.shape[0]):
            if self.mutation_probability is not None:
                genes = np.where(self.rng.random(L) <= self.mutation_probability)[0]
            else:
                genes = self.rng.choice(L, size=self.mutation_num_genes, replace=False)
            for g in genes:
                if self.gene_space is not None:
                    space = self.gene_space[g]
                    choices = space[space != off[k, g]] if len(space) > 1 else space
                    off[k, g] = self.rng.choice(choices)     # a value of the gene's space other than the current one
                else:
                    step = self.rng.uniform(self.random_mutation_min_val, self.random_mutation_max_val)
                    val = step if self.mutation_by_replacement else off[k, g] + step
                    off[k, g] = self.gene_type(np.round(val)) if self.gene_type is int else val
        return off

    # ------------------------------------------------------------------ main loop
    def run(self) -> "GeneticRun":
        self.last_generation_fitness = self._cal_pop_fitness({})
        self.best_solutions_fitness.append(float(np.max(self.last_generation_fitness)))
        for _ in range(self.num_generations):
            fitness = self.last_generation_fitness
            parents, parent_idx = self._select_parents(fitness, self.num_parents_mating)
            if self.keep_elitism > 0:
                keep_idx = np.argsort(-fitness, kind="stable")[:self.keep_elitism]
            elif self.keep_parents == -1:
                keep_idx = parent_idx
            elif self.keep_parents > 0:
                keep_idx = np.argsort(-fitness, kind="stable")[:self.keep_parents]
            else:
                keep_idx = np.empty(0, dtype=int)
            n_off = self.sol_per_pop - len(keep_idx)
            offspring = self._mutate(self._crossover(parents, n_off)) if n_off > 0 else parents[:0]
            kept = self.population[keep_idx].copy()
            carried = {i: float(fitness[j]) for i, j in enumerate(keep_idx)}
            self.population = np.concatenate([kept, offspring], axis=0)
            self.generations_completed += 1
            self.previous_generation_fitness = fitness
            self.last_generation_fitness = self._cal_pop_fitness(carried)
            self.best_solutions_fitness.append(float(np.max(self.last_generation_fitness)))
            if self.on_generation is not None and self.on_generation(self) == "stop":
                break
            if self._should_stop():
                break
        self.run_completed = True
        return self

    def _should_stop(self) -> bool:
        g = self.generations_completed
        for word, num in self.stop_criteria:
            if word == "reach" and np.max(self.last_generation_fitness) >= num:
                return True
            if word == "saturate" and g >= int(num):
                hist = self.best_solutions_fitness      # hist[k] = best after k generations
                if hist[g - int(num)] == hist[g]:
                    return True
        return False

    def best_solution(self, pop_fitness=None):
        """(chromosome, fitness, index) of the best individual of the current population."""
        fit = self.last_generation_fitness if pop_fitness is None else np.asarray(pop_fitness)
        i = int(np.argmax(fit))
        return self.population[i].copy(), float(fit[i]), i


# --------------------------------------------------------------------------- #
# artefacts: the per-chromosome files the reference appends to (genetic_v3.py:667-708)
# --------------------------------------------------------------------------- #
class GeneticArtifacts:
    """Writes ``Output_genetic/<suffix>/...`` exactly as the reference names and formats it: one line
    per evaluated chromosome in each file."""

    def __init__(self, save_dir, suffix: str):
        self.suffix = suffix
        self.path = Path(save_dir) / suffix
        self.path.mkdir(parents=True, exist_ok=True)

    def _append_line(self, name: str, value) -> None:
        with open(self.path / name, "a", encoding="UTF-8") as f:
            f.write(str(value) + "\n")

    def _append_row(self, name: str, row) -> None:
        with open(self.path / name, "a", encoding="UTF-8") as f:
            csv.writer(f).writerow(row)

    def record(self, solution, fitness_value, generation: int, stats: Optional[dict] = None) -> None:
        s = self.suffix
        self._append_row("genes" + s + ".csv", list(solution))
        if stats is not None:
            self._append_line("depth" + s + ".txt", stats["depth"])
            self._append_line("sparsity" + s + ".txt", stats["sparsity"])
            self._append_line("accuracy" + s + ".txt", stats["accuracy"])
        self._append_row("fitness_values_iter_" + s + ".csv", [fitness_value])
        if stats is not None:
            self._append_line("offdiagonal_mean_" + s + ".txt", stats["offdiagonal_mean"])
            self._append_line("offdiagonal_std_" + s + ".txt", stats["offdiagonal_std"])
            self._append_line("generation_id" + s + ".txt", generation)

    def finish(self, run: GeneticRun, data_cv=None, data_labels=None) -> None:
        """The end-of-run files of reference examples/genetic_featuremap_v3.py:148-160."""
        solution, fitness, _ = run.best_solution()
        with open(self.path / "best_solution.txt", "w") as f:
            np.savetxt(f, solution)
        with open(self.path / "best_fitness.txt", "w") as f:
            f.write(str(fitness) + "\n")
        with open(self.path / "best_fitness_per_generation.txt", "w") as f:
            f.write(str(run.best_solutions_fitness) + "\n")
        if data_cv is not None:
            np.save(self.path / "data_cv", data_cv)
        if data_labels is not None:
            np.save(self.path / "labels_cv", data_labels)


# --------------------------------------------------------------------------- #
# the reference's entry point
# --------------------------------------------------------------------------- #
def generation_fitness(solutions, data_cv, data_labels, gate_dict, nb_features, gates_per_qubits, nb_qubits,
                       fit_fun: Callable, projected: bool = False, noise_std=None, depth_fn: Optional[Callable] = None,
                       cv: int = 4, seed: int = 0, variant: str = "v3", coupling_map=None, basis_gates=None,
                       lower_fn: Optional[Callable] = None):
    """Fitness of many chromosomes at once: the body of the reference's ``fitness_func``
    (genetic_v3.py:604-666) with the kernels of the whole batch computed in one GPU call.

    ``coupling_map`` serves the v4 / v5 decoders and, with ``basis_gates``, the lowering of the
    ``without_kernel`` mode (``lower_fn(fmap, coupling_map, basis_gates, optimization_level=0)``, default
    :func:`quake_amd.transpile.transpile`).
    Returns ``(fitness values, per-chromosome stats dicts or None)``."""
    from sklearn.model_selection import cross_val_score
    from sklearn.svm import SVC

    from .kernels import evaluate_population, projected_quantum_kernel

    data_labels = np.asarray(data_labels)
    decoded = [genetic.to_quantum(s, gate_dict, nb_features, gates_per_qubits, nb_qubits, verbose=False,
                                  variant=variant, coupling_map=coupling_map) for s in solutions]
    depth_of = depth_fn if depth_fn is not None else (lambda fmap: fmap.depth())
    if "accuracy" not in inspect.signature(fit_fun).parameters:
        # "without_kernel" mode (:710-725): the fitness function is handed the circuit as
        # transpile(fmap, coupling_map, basis_gates, optimization_level=0) leaves it, not the decoder's
        if lower_fn is None:
            from .transpile import transpile as lower_fn
        return [fit_fun(lower_fn(fmap, coupling_map, basis_gates, optimization_level=0)) for fmap, _ in decoded], None
    maps = [fmap for fmap, _ in decoded]
    xs = [data_cv[:, x_idxs] for _, x_idxs in decoded]
    if projected:
        Ks = [projected_quantum_kernel(f, x, 1) for f, x in zip(maps, xs)]
        stats = [genetic.kernel_statistics(K, data_labels) for K in Ks]
    else:
        noise = ("gaussian", float(noise_std)) if noise_std is not None else None
        Ks, stats = evaluate_population(maps, xs, labels=data_labels, noise=noise, seed=seed)
    values = []
    for fmap, K, st in zip(maps, Ks, stats):
        acc = cross_val_score(SVC(kernel="precomputed"), K, data_labels, cv=cv, scoring="accuracy").mean()
        st.update(accuracy=acc, depth=depth_of(fmap))
        values.append(fit_fun(acc, st["offdiagonal_std"], st["depth"]))
    return values, stats


def genetic_instance(gene_space, data_cv, data_labels, backend, gate_dict, nb_features, gates_per_qubits, nb_qubits,
                     projected, suffix, coupling_map, basis_gates, fit_fun, noise_std=None, *,
                     save_dir="../../Output_genetic", depth_fn: Optional[Callable] = None, variant: str = "v3",
                     verbose: bool = True, **kwargs) -> GeneticRun:
    """Same positional signature as reference ``genetic_instance`` (genetic_v3.py:451-533); returns
    a :class:`GeneticRun` where the reference returns ``pygad.GA``.  ``backend`` is accepted and
    unused (as in the reference's statevector path).  ``depth`` comes from
    ``transpile.transpiled_depth(fmap, coupling_map, basis_gates)``, the native stand-in for
    ``transpile(...).depth()`` (reference :652-654), unless ``depth_fn`` is given."""
    from .transpile import transpiled_depth

    if depth_fn is None:
        depth_fn = lambda fmap: transpiled_depth(fmap, coupling_map, basis_gates)
    start_time = time.time()
    art = GeneticArtifacts(save_dir, suffix) if save_dir is not None else None
    counter = {"seed": int(kwargs.get("random_seed") or 0)}

    def fitness_batch(run: GeneticRun, solutions, indices):
        values, stats = generation_fitness(solutions, data_cv, data_labels, gate_dict, nb_features, gates_per_qubits,
                                           nb_qubits, fit_fun, projected=projected, noise_std=noise_std,
                                           depth_fn=depth_fn, seed=counter["seed"], variant=variant,
                                           coupling_map=coupling_map, basis_gates=basis_gates)
        counter["seed"] += len(solutions)
        for k, sol in enumerate(solutions):
            st = stats[k] if stats is not None else None
            if verbose and st is not None:   # reference :662-665
                print("depth", st["depth"]); print("sparsity", st["sparsity"])
                print("accuracy", st["accuracy"]); print("fitness_value", values[k])
            if art is not None:
                art.record(sol, values[k], run.generations_completed, st)
        return values

    def on_generation(run: GeneticRun):    # reference callback_func, :745-761
        if verbose:
            print("Generation:", run.generations_completed)
            print(f"Best fitness: {run.last_generation_fitness.max()}")
            print("Elapsed time: " + str(time.time() - start_time) + "s")

    kwargs.setdefault("on_generation", on_generation)
    kwargs.setdefault("save_solutions", True)
    run = GeneticRun(fitness_batch=fitness_batch, gene_space=gene_space, gene_type=int, **kwargs)
    run.artifacts = art
    return run
