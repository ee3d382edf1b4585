"""Callers of kernel.evaluate (reference qsvm/train.py:30-166, genetic_v3.py:604-661) on the
GPU path vs the same sklearn plumbing fed by the oracle: identical SVC predictions."""
import numpy as np
import pytest

from oracle import qk_oracle as O
from quake_amd import genetic, qsvm

pytestmark = pytest.mark.gpu


class OracleKernel:
    def __init__(self, ops, n):
        self.ops, self.n = ops, n

    def evaluate(self, x_vec, y_vec=None):
        return O.evaluate(self.ops, self.n, x_vec, y_vec)


def _toy(seed, n, F=2):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, F))
    y = (X[:, 0] * X[:, 1] > 0).astype(float)       # XOR-like, needs a non-linear kernel
    return X, y


def test_train_and_evaluate_match_oracle(tmp_path):
    feats, labs = zip(*[_toy(s, n) for s, n in ((1, 120), (2, 60), (3, 50))])
    setup = {"model": {"qsvm": {"should_do_scaling": True, "split_ratio": 0.5, "should_add_extra_feats": False}}}
    models, metrics = qsvm.qsvm_train_from_features(list(feats), list(labs), tmp_path, setup)
    assert (tmp_path / "c2.sav").exists() and (tmp_path / "gen1.sav").exists() and (tmp_path / "gen2.sav").exists()
    # same flow with oracle kernels
    from sklearn.model_selection import train_test_split
    ds = qsvm.rearrange_scale(*feats, True)
    lb = list(labs)
    ds[0], lb[0] = train_test_split(ds[0], lb[0], train_size=0.5, random_state=42)[::2]
    oks = [OracleKernel(O.custom2_featuremap(2, 1), 2), OracleKernel(O.genetic_featuremap(1), 2),
           OracleKernel(O.genetic_featuremap_2(1), 2)]
    omodels = qsvm.qsvm_hyperparameter_training(None, ds, lb, oks, ["c2", "gen1", "gen2"])
    ometrics = qsvm.evaluate_svm(omodels, ds, lb, oks)
    for key in ("accuracy", "sensitivity", "specificity"):
        assert np.array_equal(metrics[key], ometrics[key]), key
    # AUC goes through predict_proba: Platt scaling with an unseeded internal 5-fold split (the
    # reference passes no random_state), so it moves in the 4th digit from fit to fit
    assert np.abs(metrics["auc"] - ometrics["auc"]).max() < 2e-2
    assert metrics["accuracy"].shape == (3, 3) and metrics["accuracy"][:, 0].min() > 0.5
    # the intended model (kernel="precomputed" on the final refit) also runs
    kernels, titles = qsvm.load_quantum_kernels(2)
    m2 = qsvm.qsvm_hyperparameter_training(None, ds, lb, kernels, titles, final_kernel="precomputed")
    assert m2[0].kernel == "precomputed" and models[0].kernel == "rbf"


def test_qsvm_train_from_npy_folder(tmp_path):
    """reference train.py:190-272 with the extractor replaced by .npy files: same models as the
    in-memory entry on the same arrays"""
    data = tmp_path / "data"
    data.mkdir()
    sets = {s: _toy(seed, n, F=3) for s, seed, n in (("train", 7, 90), ("val", 8, 40), ("test", 9, 30))}
    for s, (x, y) in sets.items():
        np.save(data / f"features_{s}.npy", x)
        np.save(data / f"labels_{s}.npy", y)
    setup = {"model": {"qsvm": {"feature_extractor": "cnn", "should_do_scaling": True, "split_ratio": 0.5,
                                "should_add_extra_feats": False, "feature_maps": [["zz", {"repeats": 1}]]}}}
    models, metrics = qsvm.qsvm_train(data, tmp_path / "out", setup)
    assert (tmp_path / "out" / "zz.sav").exists()
    m2, metrics2 = qsvm.qsvm_train_from_features([sets[s][0] for s in ("train", "val", "test")],
                                                 [sets[s][1] for s in ("train", "val", "test")], None, setup)
    assert np.array_equal(metrics["accuracy"], metrics2["accuracy"])
    assert models[0].C == m2[0].C
    setup["model"]["qsvm"]["feature_extractor"] = "svm"
    with pytest.raises(NotImplementedError):
        qsvm.qsvm_train(data, tmp_path / "out", setup)


def test_runcard_feature_maps(tmp_path):
    card = tmp_path / "runcard.yaml"
    card.write_text("output: !Path out\nmodel:\n  qsvm:\n    should_do_scaling: true\n    split_ratio: 0.6\n"
                    "    feature_maps: [[zz, {repeats: 2}], [z, {}]]\n")
    setup = qsvm.load_runcard(card)
    assert str(setup["output"]) == "out"
    feats, labs = zip(*[_toy(s, n, F=4) for s, n in ((4, 80), (5, 40), (6, 30))])
    models, metrics = qsvm.qsvm_train_from_features(list(feats), list(labs), None, setup)
    assert len(models) == 2 and np.isfinite(metrics["auc"]).all()


def test_genetic_fitness_matches_oracle(pins, data8, data_labels):
    from sklearn.model_selection import cross_val_score
    from sklearn.svm import SVC
    p = pins["P2"]
    fit = lambda accuracy, density, depth: accuracy + 0.05 * density   # examples/torino_run.py:96-98
    val, stats = genetic.fitness(p["genes"], data8, data_labels, p["gate_dict"], 8, 8, 4, fit)
    ops, xi = O.to_quantum(p["genes"], p["gate_dict"], 8, 8, 4)
    K = O.evaluate(ops, 4, data8[:, xi])
    acc = cross_val_score(SVC(kernel="precomputed"), K, data_labels, cv=4, scoring="accuracy").mean()
    ref = genetic.kernel_statistics(K, data_labels)
    assert stats["accuracy"] == acc
    for k in ("sparsity", "offdiagonal_mean", "offdiagonal_std"):
        assert abs(stats[k] - ref[k]) < 1e-13
    assert abs(val - (acc + 0.05 * ref["offdiagonal_std"])) < 1e-13
    # projected variant runs through the same entry point
    v2, s2 = genetic.fitness(p["genes"], data8, data_labels, p["gate_dict"], 8, 8, 4, fit, projected=True)
    Kp = O.projected_kernel(ops, 4, data8[:, xi], 1)
    assert abs(s2["offdiagonal_std"] - np.std(np.triu(Kp, 1))) < 1e-13


def _kernel_pairs():
    """(HIP kernel, oracle kernel) for two maps over two features"""
    from quake_amd import featuremaps as fm
    from quake_amd.circuit import ParameterVector
    from quake_amd.kernels import make_kernels
    x = ParameterVector("x", 2)
    hip = make_kernels([fm.zz_featuremap(x, 2, 2), fm.custom2_featuremap(x, 2)], None)
    ora = [OracleKernel(O.zz_featuremap(2, 2), 2), OracleKernel(O.custom2_featuremap(2, 1), 2)]
    return hip, ora


def test_train_quantum_matches_oracle():
    """reference qsvm_tester.py:625-695: label-sorted training set, train / val x train / test x train
    matrices, one precomputed-kernel SVC per kernel -- GPU kernels against oracle kernels through the
    same plumbing, and the plumbing against a direct statement of what it must deliver"""
    from sklearn.svm import SVC
    (Xtr, ytr), (Xv, _), (Xt, _) = _toy(11, 150), _toy(12, 40), _toy(13, 30)
    Xtr, Xv, Xt = (np.clip(a, -1.5, 1.5) for a in (Xtr, Xv, Xt))
    hip, ora = _kernel_pairs()
    cs = [1.0, 10.0]
    got = qsvm.train_quantum(Xtr, ytr, Xv, Xt, hip, cs)
    ref = qsvm.train_quantum(Xtr, ytr, Xv, Xt, ora, cs)
    for g, r in zip(got[1:4], ref[1:4]):
        for a, b in zip(g, r):
            assert np.array_equal(a, b)
    assert len(got[4]) == 2 and got[4][0].shape == (40, 150)          # 150 training rows < 201: val matrices kept
    assert np.abs(got[4][0] - ref[4][0]).max() < 1e-12
    # the plumbing itself: rows sorted by label (all zeros first), fitted on the sorted matrix
    order = np.argsort(ytr)
    Xs, ys = Xtr[order], ytr[order]
    assert np.all(np.diff(ys) >= 0)
    K = O.evaluate(ora[0].ops, 2, Xs)
    clf = SVC(kernel="precomputed", C=cs[0]).fit(K, ys)
    assert np.array_equal(got[1][0], clf.predict(K))
    assert np.array_equal(got[2][0], clf.predict(O.evaluate(ora[0].ops, 2, Xv, Xs)))
    assert np.array_equal(got[3][0], clf.predict(O.evaluate(ora[0].ops, 2, Xt, Xs)))
    assert np.array_equal(got[0][0].support_, clf.support_) and got[0][1].C == 10.0
    # more than 200 training rows: no matrices are kept (reference :687-688)
    big = qsvm.train_quantum(*_toy(14, 220), Xv, Xt, hip[:1], cs)
    assert big[4] == [] and big[1][0].shape == (220,)
    with pytest.raises(IndexError):
        qsvm.train_quantum(Xtr, ytr, Xv, Xt, hip, [1.0])


def test_cv_scores_matches_oracle():
    """reference qsvm_tester.py:1430-1471: seeded subsamples (seed 2) of growing size, cross_val_score per
    kernel and size, std = np.std / sqrt(folds)"""
    from sklearn.model_selection import cross_val_score, train_test_split
    from sklearn.svm import SVC
    X, y = _toy(21, 200)
    X = np.clip(X, -1.5, 1.5)
    hip, ora = _kernel_pairs()
    sizes, cs, folds = [40, 90], [1.0, 100.0], 3
    seen = []
    score, mean, std = qsvm.cv_scores(X, y, folds, hip, sizes, cs, on_score=lambda s: seen.append(s.copy()))
    oscore, omean, ostd = qsvm.cv_scores(X, y, folds, ora, sizes, cs)
    assert score.shape == (2, 2, 3) and np.array_equal(score, oscore)
    assert np.array_equal(mean, omean) and np.array_equal(std, ostd) and len(seen) == 4
    sub, lab = train_test_split(X, y, train_size=90, random_state=2)[::2]
    direct = cross_val_score(SVC(kernel="precomputed", C=100.0), O.evaluate(ora[1].ops, 2, sub), lab, cv=folds)
    assert np.array_equal(score[1, 1], direct)
    assert np.allclose(std, np.std(score, axis=2) / np.sqrt(folds)) and np.allclose(mean, score.mean(axis=2))
