"""QSVM training / evaluation around the HIP kernel: the callers of ``kernel.evaluate``.

Mirrors ``src/quake/models/qsvm/train.py`` of the reference (function names, argument
meaning, the order of operations and the sklearn objects) with two differences that
the scope of this build imposes:

* features arrive as arrays (``.npy`` files or in memory) instead of being pulled out
  of the TensorFlow CNN / attention extractors (reference ``train.py:210-240``); those
  extractors stay upstream and untouched;
* the Gram matrices come from :class:`quake_amd.kernels.HipFidelityKernel`.
"""
from __future__ import annotations

import logging
import pickle
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .circuit import ParameterVector
from .featuremaps import build_featuremap, custom2_featuremap, genetic_featuremap, genetic_featuremap_2
from .kernels import make_kernels

logger = logging.getLogger("quake_amd.qsvm")


def rearrange_scale(train_set, val_set, test_set, should_do_scaling: bool) -> list:
    """Min-max scale to ``(-pi/2 + 1e-5, pi/2 - 1e-5)``, fitted on train+val+test together
    (reference ``svm/utils.py:93-130``): the input range the hand-written maps expect."""
    from sklearn import preprocessing

    full = [train_set, val_set, test_set]
    if should_do_scaling:
        scaler = preprocessing.MinMaxScaler(feature_range=(-np.pi / 2 + 1e-5, np.pi / 2 - 1e-5)).fit(np.concatenate(full))
        full = [scaler.transform(d) for d in full]
    return full


def load_quantum_kernels(qubits: int, maps: Optional[Sequence[Tuple[str, dict]]] = None):
    """Kernel objects and their titles (reference ``train.py:169-187``: c2, gen1, gen2 over
    ``x = ParameterVector("x", qubits)``).  ``maps`` optionally replaces the hard-wired
    list with runcard entries ``[(name, {"repeats": .., "align": ..}), ...]``."""
    if maps is not None:
        circuits = [build_featuremap(name, qubits, **(opts or {})) for name, opts in maps]
        return make_kernels(circuits, None), [name for name, _ in maps]
    x = ParameterVector("x", length=qubits)
    c2 = custom2_featuremap(x, qubits)
    gen1 = genetic_featuremap(x)
    gen2 = genetic_featuremap_2(x)
    return make_kernels([c2, gen1, gen2], None), ["c2", "gen1", "gen2"]


def qsvm_hyperparameter_training(train_folder, dataset, labels, qt_kernels, kernel_titles,
                                 should_add_extra_feats: bool = False, val_size: int = 300,
                                 final_kernel: str = "rbf", verbose: int = 0) -> list:
    """Grid search over C with a predefined train/validation split, then refit on the
    train block (reference ``train.py:30-103``).

    ``final_kernel``: the reference refits ``SVC(probability=True, **best)`` WITHOUT
    ``kernel="precomputed"`` (train.py:95), i.e. an RBF SVC over the rows of the Gram
    matrix.  ``"rbf"`` (default) reproduces that so predictions stay identical to the
    reference; pass ``"precomputed"`` for the evidently intended model."""
    from sklearn.model_selection import GridSearchCV, PredefinedSplit
    from sklearn.svm import SVC

    feature_size = dataset[0].shape[1]
    logger.info("Training, validating, testing QSVMs ... using %d features%s", feature_size,
                ", including Nhits and total energy" if should_add_extra_feats else "")
    quantum_grid = {"C": [0.01, 0.1, 1, 10, 100, 1000, 10000]}
    train_size = labels[0].shape[0]
    val_size = min(val_size, labels[1].shape[0])   # reference hard-codes 300 "for saving computing time"
    partitions = np.append(-np.ones(train_size, dtype=int), np.zeros(val_size, dtype=int))
    validation_idx = PredefinedSplit(partitions)
    set_train_val = np.concatenate((dataset[0], dataset[1][:val_size]), axis=0)
    labels_train_val = np.concatenate((labels[0], labels[1][:val_size]), axis=0)

    models = []
    for k, kernel in enumerate(qt_kernels):
        logger.info("Fitting QSVC with %s kernel", kernel_titles[k])
        grid = GridSearchCV(SVC(kernel="precomputed", probability=True), quantum_grid, refit=True,
                            verbose=verbose, cv=validation_idx)
        ker_matrix_cv = kernel.evaluate(x_vec=set_train_val)          # the hot path
        grid.fit(ker_matrix_cv, labels_train_val)
        logger.info(grid.best_params_)
        if final_kernel == "precomputed":
            quantum_svc = SVC(kernel="precomputed", probability=True, **grid.best_params_)
        else:
            quantum_svc = SVC(probability=True, **grid.best_params_)
        quantum_svc.fit(ker_matrix_cv[:train_size, :train_size], labels[0])
        if train_folder is not None:
            with open(Path(train_folder) / (kernel_titles[k] + ".sav"), "wb") as f:
                pickle.dump(quantum_svc, f)
        models.append(quantum_svc)
    return models


def evaluate_svm(models, dataset, labels, qt_kernels) -> dict:
    """Accuracy / sensitivity / specificity / AUC on train, validation, test
    (reference ``train.py:106-166``); the matrices are logged and returned."""
    from sklearn.metrics import roc_auc_score

    k_len = len(qt_kernels)
    acc, sen, spec, auc = (np.zeros((k_len, 3)) for _ in range(4))
    for k, qsvc in enumerate(models):
        for j in range(3):
            if j == 0:
                ker_matrix = qt_kernels[k].evaluate(x_vec=dataset[j])
            else:
                ker_matrix = qt_kernels[k].evaluate(x_vec=dataset[j], y_vec=dataset[0])
            y, lab = qsvc.predict(ker_matrix), labels[j]
            acc[k, j] = np.sum(y == lab) / lab.shape[0]
            sen[k, j] = np.sum(np.logical_and(y == 1, lab == 1)) / np.sum(lab)
            spec[k, j] = np.sum(np.logical_and(y == 0, lab == 0)) / np.sum(np.logical_not(lab))
            auc[k, j] = roc_auc_score(lab, qsvc.predict_proba(ker_matrix)[:, 1])
    logger.info("Metrics matrices. Columns: train, validation, test")
    for name, m in (("Accuracy", acc), ("Sensitivity", sen), ("Specificity", spec), ("AUC", auc)):
        logger.info("%s: \n%s", name, np.array2string(m, precision=3))
    return {"accuracy": acc, "sensitivity": sen, "specificity": spec, "auc": auc}


def load_runcard(path) -> dict:
    """YAML runcard with the reference's ``!Path`` tag (reference ``utils/utils.py:15-66``)."""
    import yaml

    class _Loader(yaml.SafeLoader):
        pass

    _Loader.add_constructor("!Path", lambda loader, node: Path(loader.construct_scalar(node)))
    with open(path) as f:
        return yaml.load(f, Loader=_Loader)


def qsvm_train_from_features(features, labels, train_folder, setup: dict, seed: int = 42):
    """The tail of the reference's ``qsvm_train`` (train.py:242-272) on already extracted
    features: ``features`` / ``labels`` are ``[train, val, test]`` lists of arrays.

    ``setup["model"]["qsvm"]`` keeps the reference keys (``should_do_scaling``,
    ``split_ratio``, ``should_add_extra_feats``; runcard.yaml:73-78) plus the optional
    ``feature_maps: [[name, {repeats: 1}], ...]``."""
    from sklearn.model_selection import train_test_split

    q = setup["model"]["qsvm"]
    dataset = rearrange_scale(features[0], features[1], features[2], q.get("should_do_scaling", True))
    labels = list(labels)
    dataset[0], labels[0] = train_test_split(dataset[0], labels[0], train_size=q["split_ratio"], random_state=seed)[::2]
    qubits = dataset[0].shape[1]
    kernels, titles = load_quantum_kernels(qubits, q.get("feature_maps"))
    models = qsvm_hyperparameter_training(train_folder, dataset, labels, kernels, titles,
                                          q.get("should_add_extra_feats", False),
                                          final_kernel=q.get("final_kernel", "rbf"))
    metrics = evaluate_svm(models, dataset, labels, kernels)
    return models, metrics


def load_feature_files(data_folder) -> Tuple[list, list]:
    """``[train, val, test]`` features and labels from ``features_<split>.npy`` /
    ``labels_<split>.npy`` under ``data_folder`` (what an upstream extractor run saves;
    stands in for reference ``train.py:221-240``).  Plain arrays only: no pickles."""
    data_folder = Path(data_folder)
    feats, labs = [], []
    for split in ("train", "val", "test"):
        fx, fy = data_folder / f"features_{split}.npy", data_folder / f"labels_{split}.npy"
        if not fx.is_file() or not fy.is_file():
            raise FileNotFoundError(f"expected {fx.name} and {fy.name} in {data_folder}")
        x = np.load(fx, allow_pickle=False)
        y = np.load(fy, allow_pickle=False)
        if x.ndim != 2 or y.ndim != 1 or x.shape[0] != y.shape[0]:
            raise ValueError(f"{split}: features {x.shape} and labels {y.shape} do not match")
        feats.append(np.asarray(x, dtype=np.float64))
        labs.append(y)
    if len({f.shape[1] for f in feats}) != 1:
        raise ValueError("train / val / test features differ in width")
    return feats, labs


def qsvm_train(data_folder, train_folder, setup: dict):
    """Drop-in for the reference's ``qsvm_train(data_folder, train_folder, setup)``
    (train.py:190-272) with the feature extraction replaced by
    :func:`load_feature_files`; everything from ``rearrange_scale`` on is the reference's
    sequence, with the Gram matrices computed on the GPU."""
    extractor_type = str(setup["model"]["qsvm"].get("feature_extractor", "npy")).lower()
    if extractor_type not in ("npy", "cnn", "attention"):
        raise NotImplementedError(f"exctractor model not implemented, found: {extractor_type}")
    features, labels = load_feature_files(data_folder)
    train_folder = Path(train_folder)
    train_folder.mkdir(parents=True, exist_ok=True)
    return qsvm_train_from_features(features, labels, train_folder, setup)


# --------------------------------------------------------------------------- #
# the comparison harness's two kernel consumers (reference qsvm_tester.py)
# --------------------------------------------------------------------------- #
def get_subsample(dataset, labels, size, seed):
    """A seeded subsample of ``size`` rows (reference ``qsvm_tester.py:207-233``: the train half of
    ``train_test_split(..., train_size=size, random_state=seed)``)."""
    from sklearn.model_selection import train_test_split

    return tuple(train_test_split(dataset, labels, train_size=size, random_state=seed)[::2])


def _sorted_by_label(X: np.ndarray, y: np.ndarray):
    """Rows reordered by ``np.argsort`` of the labels, features and labels promoted to one dtype: what the
    reference gets by gluing the label column onto the features and sorting the rows on it
    (``qsvm_tester.py:677-683``).  Same ``argsort`` call on the same values, so ties fall the same way."""
    common = np.result_type(X.dtype, y.dtype)
    y = np.asarray(y).astype(common, copy=False)
    order = np.argsort(y)
    return np.asarray(X).astype(common, copy=False)[order], y[order]


def train_quantum(training_dataset, training_labels, val_dataset, test_dataset, quantum_kernels, cs):
    """One precomputed-kernel SVC per quantum kernel on a label-sorted training set (reference
    ``qsvm_tester.py:625-695``).  Returns ``(svcs, pred_train, pred_val, pred_test, kers)``; ``kers`` holds
    the validation x train matrices when there are at most 200 training rows (reference :687-688).

    The training set is re-sorted before every kernel, from the order the previous kernel left it in --
    the reference reassigns its arguments inside the loop -- so row order, and with it the order of the
    support vectors and of ``pred_train``, follows the reference for any number of kernels."""
    from sklearn.svm import SVC

    X, y = np.asarray(training_dataset), np.asarray(training_labels)
    if len(cs) < len(quantum_kernels):
        raise IndexError("cs needs one cost value per kernel")    # the reference indexes cs[q]
    out = {"svcs": [], "train": [], "val": [], "test": [], "kers": []}
    for kernel, c in zip(quantum_kernels, cs):
        X, y = _sorted_by_label(X, y)
        blocks = {"train": kernel.evaluate(x_vec=X),                       # the hot path, three shapes
                  "val": kernel.evaluate(x_vec=val_dataset, y_vec=X),
                  "test": kernel.evaluate(x_vec=test_dataset, y_vec=X)}
        if y.shape[0] < 201:
            out["kers"].append(np.real(blocks["val"]))
        clf = SVC(kernel="precomputed", C=c).fit(blocks["train"], y)
        out["svcs"].append(clf)
        for split, K in blocks.items():
            out[split].append(clf.predict(K))
    return out["svcs"], out["train"], out["val"], out["test"], out["kers"]


def cv_scores(train_dataset, train_labels, folds: int, kernels, training_size: Sequence[int], cs: Sequence[float],
              seed: int = 2, on_score=None):
    """Cross-validated accuracy of every kernel on seeded subsamples of growing size (reference
    ``SvmsComparison.cv_scores``, ``qsvm_tester.py:1430-1471``; ``training_size`` / ``cs`` are the attributes
    the method reads from its object, ``seed`` the literal 2 it passes to ``get_subsample``).

    Returns ``(score[size, kernel, fold], mean[size, kernel], std[size, kernel])`` with
    ``std = np.std(score, axis=2) / sqrt(folds)``.  ``on_score(score)`` is called after every block, where
    the reference pickles its partial table (``score_TEMP.pkl``)."""
    from sklearn.model_selection import cross_val_score
    from sklearn.svm import SVC

    score = np.zeros((len(training_size), len(kernels), folds))
    for q, kernel in enumerate(kernels):
        for i, size in enumerate(training_size):
            subset, labels = get_subsample(train_dataset, train_labels, size, seed)
            score[i, q] = cross_val_score(SVC(kernel="precomputed", C=cs[q]), kernel.evaluate(x_vec=subset), labels, cv=folds)
            if on_score is not None:
                on_score(score)
    return score, np.mean(score, axis=2), np.std(score, axis=2) / np.sqrt(folds)
