"""The GA driver that stands where the reference uses pygad (quake_amd/ga.py): operators,
bookkeeping, stop criteria and the artefact files, with a toy fitness (no GPU)."""
import csv

import numpy as np
import pytest

from quake_amd import ga


def _space(L, hi=10):
    return [range(hi)] * L


def _sum_fitness(calls):
    def f(run, solutions, indices):
        calls.append(len(solutions))
        return [float(s.sum()) for s in solutions]
    return f


def test_sss_keeps_parents_and_never_loses_the_best():
    rng = np.random.default_rng(0)
    pop = rng.integers(0, 10, size=(12, 20))
    calls = []
    run = ga.GeneticRun(initial_population=pop, num_generations=15, num_parents_mating=6, fitness_batch=_sum_fitness(calls),
                        gene_space=_space(20), parent_selection_type="sss", keep_elitism=0, keep_parents=-1,
                        crossover_type="two_points", crossover_probability=0.5, mutation_type="random",
                        mutation_percent_genes=10, random_seed=3).run()
    best = run.best_solutions_fitness
    assert len(best) == 16 and run.generations_completed == 15
    assert all(b1 >= b0 for b0, b1 in zip(best, best[1:]))          # parents are kept: monotone
    assert best[-1] > best[0]
    # generation 0 evaluates everyone; later generations only the (at most) 6 offspring
    assert calls[0] == 12 and all(c <= 6 for c in calls[1:])
    assert run.evaluations == sum(calls)
    sol, fit, idx = run.best_solution()
    assert fit == best[-1] == float(sol.sum()) and (run.population[idx] == sol).all()
    assert run.population.min() >= 0 and run.population.max() <= 9   # mutation stays inside gene_space


def test_duplicates_and_seen_solutions_are_not_re_evaluated():
    pop = np.tile(np.arange(8), (6, 1))                              # six identical chromosomes
    calls = []
    run = ga.GeneticRun(initial_population=pop, num_generations=1, num_parents_mating=2, fitness_batch=_sum_fitness(calls),
                        gene_space=_space(8), keep_elitism=0, mutation_type=None, crossover_type="single_point",
                        save_solutions=True, random_seed=0).run()
    assert calls == [1]                                              # gen 0: one unique; gen 1: all children identical and known
    assert len(run.solutions) == 12


def test_crossover_and_mutation_shapes():
    L = 16
    a, b = np.zeros(L, dtype=int), np.ones(L, dtype=int)
    run = ga.GeneticRun(initial_population=np.stack([a, b]), num_generations=1, num_parents_mating=2,
                        fitness_batch=lambda r, s, i: [0.0] * len(s), gene_space=_space(L, 5), crossover_type="two_points",
                        mutation_num_genes=3, random_seed=1)
    off = run._crossover(np.stack([a, b]), 50)
    for k, child in enumerate(off):   # p[:p1] + q[p1:p1+L/2] + p[p1+L/2:]: the second parent gives one run of <= L/2 genes
        mid = np.flatnonzero(child == (1 if k % 2 == 0 else 0))
        assert len(mid) <= L // 2 and (len(mid) == 0 or mid[-1] - mid[0] + 1 == len(mid))
    before = off.copy()
    after = run._mutate(off.copy())
    assert ((before != after).sum(axis=1) == 3).all()                # exactly mutation_num_genes genes change value
    assert after.min() >= 0 and after.max() <= 4
    # percent -> count as pygad: int(pct * L / 100), at least 1
    assert ga.GeneticRun(initial_population=np.stack([a, b]), num_generations=1, num_parents_mating=2,
                         fitness_batch=lambda r, s, i: [0.0] * len(s), mutation_percent_genes=1.5).mutation_num_genes == 1


def test_stop_criteria_and_callback():
    pop = np.random.default_rng(1).integers(0, 3, size=(8, 6))
    const = lambda r, s, i: [1.0] * len(s)
    run = ga.GeneticRun(initial_population=pop, num_generations=50, num_parents_mating=4, fitness_batch=const,
                        gene_space=_space(6, 3), stop_criteria="saturate_5", random_seed=0).run()
    assert run.generations_completed == 5
    run = ga.GeneticRun(initial_population=pop, num_generations=50, num_parents_mating=4,
                        fitness_batch=lambda r, s, i: [float(x.sum()) for x in s], gene_space=_space(6, 3),
                        stop_criteria=["reach_11", "saturate_40"], keep_elitism=2, random_seed=0).run()
    assert run.best_solutions_fitness[-1] >= 11 and run.generations_completed < 50
    seen = []
    ga.GeneticRun(initial_population=pop, num_generations=10, num_parents_mating=4, fitness_batch=const,
                  on_generation=lambda r: seen.append(r.generations_completed) or ("stop" if len(seen) == 3 else None)).run()
    assert seen == [1, 2, 3]
    with pytest.raises(ValueError):
        ga.GeneticRun(initial_population=pop, num_generations=1, num_parents_mating=4, fitness_batch=const, stop_criteria="until_5")
    with pytest.raises(NotImplementedError):
        ga.GeneticRun(initial_population=pop, num_generations=1, num_parents_mating=4, fitness_batch=const,
                      parent_selection_type="tournament_nsga2")


@pytest.mark.parametrize("kind", ["rank", "tournament", "rws", "random"])
def test_other_parent_selections_run(kind):
    pop = np.random.default_rng(2).integers(0, 10, size=(10, 12))
    run = ga.GeneticRun(initial_population=pop, num_generations=8, num_parents_mating=5,
                        fitness_func=lambda r, s, i: float(s.sum()), gene_space=_space(12), parent_selection_type=kind,
                        keep_elitism=1, crossover_type="uniform", random_seed=5).run()
    assert all(b1 >= b0 for b0, b1 in zip(run.best_solutions_fitness, run.best_solutions_fitness[1:]))  # elitism


def test_artifact_files(tmp_path):
    art = ga.GeneticArtifacts(tmp_path, "run7")
    stats = dict(depth=9, sparsity=0.125, accuracy=0.75, offdiagonal_mean=0.2, offdiagonal_std=0.3)
    art.record(np.array([1, 0, 1, 3, 2]), 0.7575, 0, stats)
    art.record(np.array([2, 1, 0, 0, 1]), 0.5, 1, stats)
    d = tmp_path / "run7"
    names = sorted(p.name for p in d.iterdir())
    assert names == sorted(["genesrun7.csv", "depthrun7.txt", "sparsityrun7.txt", "accuracyrun7.txt",
                            "fitness_values_iter_run7.csv", "offdiagonal_mean_run7.txt", "offdiagonal_std_run7.txt",
                            "generation_idrun7.txt"])
    rows = list(csv.reader(open(d / "genesrun7.csv")))
    assert [r for r in rows if r] == [["1", "0", "1", "3", "2"], ["2", "1", "0", "0", "1"]]
    assert (d / "generation_idrun7.txt").read_text() == "0\n1\n"
    assert (d / "accuracyrun7.txt").read_text() == "0.75\n0.75\n"
    pop = np.array([[1, 2], [3, 4]])
    run = ga.GeneticRun(initial_population=pop, num_generations=1, num_parents_mating=2,
                        fitness_batch=lambda r, s, i: [float(x.sum()) for x in s], gene_space=_space(2), random_seed=0).run()
    art.finish(run, data_cv=np.zeros((3, 2)), data_labels=np.zeros(3))
    assert np.loadtxt(d / "best_solution.txt").shape == (2,)
    assert (d / "best_fitness_per_generation.txt").read_text().startswith("[")
    assert np.load(d / "data_cv.npy").shape == (3, 2) and np.load(d / "labels_cv.npy").shape == (3,)


def test_load_feature_files_checks(tmp_path):
    """the .npy stand-in for the reference's extractor stage (train.py:221-240): plain arrays,
    matching lengths and widths, loud errors otherwise"""
    import numpy as np
    from quake_amd import qsvm
    with pytest.raises(FileNotFoundError):
        qsvm.load_feature_files(tmp_path)
    for s, n in (("train", 6), ("val", 4), ("test", 3)):
        np.save(tmp_path / f"features_{s}.npy", np.zeros((n, 3), dtype=np.float32))
        np.save(tmp_path / f"labels_{s}.npy", np.arange(n) % 2)
    feats, labs = qsvm.load_feature_files(tmp_path)
    assert [f.shape for f in feats] == [(6, 3), (4, 3), (3, 3)] and feats[0].dtype == np.float64
    assert [len(y) for y in labs] == [6, 4, 3]
    np.save(tmp_path / "labels_val.npy", np.zeros(5))
    with pytest.raises(ValueError):
        qsvm.load_feature_files(tmp_path)
    np.save(tmp_path / "labels_val.npy", np.zeros(4))
    np.save(tmp_path / "features_test.npy", np.zeros((3, 2)))
    with pytest.raises(ValueError):
        qsvm.load_feature_files(tmp_path)


def test_get_subsample_scales_drops_outliers_and_splits():
    """reference genetic_v3.py:24-62: MinMax transform with a FITTED scaler, rows touching the range's
    bounds dropped (>= / <=), seeded train_test_split subsample"""
    from sklearn.model_selection import train_test_split
    from sklearn.preprocessing import MinMaxScaler
    from quake_amd.genetic import get_subsample
    rng = np.random.default_rng(5)
    fit_on = rng.normal(size=(400, 6))
    data = np.concatenate([fit_on[:300], 3.0 * rng.normal(size=(40, 6))])     # some rows leave the fitted range
    labels = rng.integers(0, 2, size=data.shape[0])
    scaler = MinMaxScaler((0, 1)).fit(fit_on)
    X, y = get_subsample(data, labels, 100, seed=42, scaler=scaler)
    assert X.shape == (100, 6) and y.shape == (100,)
    assert X.min() > 0.0 and X.max() < 1.0            # the OPEN interval: asin arguments of the decoder stay inside (-1, 1)
    # an independent statement of the same selection
    S = scaler.transform(data)
    keep = np.array([not any(v >= 1.0 or v <= 0.0 for v in row) for row in S])
    assert 0 < keep.sum() < data.shape[0]             # rows that hold a fitted minimum / maximum go too
    Xr, yr = train_test_split(S[keep], labels[keep], train_size=100, random_state=42)[::2]
    assert np.array_equal(X, Xr) and np.array_equal(y, yr)
    X2, _ = get_subsample(data, labels, 100, seed=7, scaler=scaler)
    assert not np.array_equal(X, X2)
    with pytest.raises(AttributeError):
        get_subsample(data, labels, 10)
