"""Kernel alignment for the ``align=True`` feature maps (reference qsvm_tester.py:236-280).

The reference hands a ``TrainableFidelityQuantumKernel`` to qiskit_machine_learning's
``QuantumKernelTrainer`` with ``SVCLoss(C=c)`` and qiskit's ``SPSA(maxiter=20)`` starting from
theta = 0, on at most 500 samples.  Those classes are third-party and absent here; this module
restates their published algorithms:

* ``svc_loss``: fit ``SVC(kernel="precomputed", C=C)`` on K(theta) and return the dual objective
  ``sum_i |a_i| - 1/2 sum_ij a_i a_j K_ij`` over the support vectors (``a`` = signed dual
  coefficients) -- qiskit_machine_learning ``SVCLoss.evaluate``.
* ``spsa_minimize``: simultaneous-perturbation stochastic approximation with power-series gains
  ``a_k = a / (k + 1)^0.602``, ``c_k = c / (k + 1)^0.101``, ``c = 0.2`` and ``a`` calibrated so
  the first step has magnitude ``2 pi / 10`` (25 probe pairs) -- qiskit ``SPSA`` defaults.

Each SPSA iteration needs K(theta + c_k D) and K(theta - c_k D): both go to the GPU in one
``evaluate_population`` call.  The random perturbations come from a seeded generator here
(qiskit's are unseeded), so the trajectory is **parity unpinned**; what is checked is that the
loss is the SVC dual objective and that alignment lowers it.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np

from .kernels import HipFidelityKernel, evaluate_population


def svc_loss(K: np.ndarray, labels: np.ndarray, C: float = 1.0) -> float:
    from sklearn.svm import SVC

    svc = SVC(kernel="precomputed", C=C).fit(K, labels)
    a = svc.dual_coef_[0]
    sv = svc.support_
    return float(np.sum(np.abs(a)) - 0.5 * a @ K[np.ix_(sv, sv)] @ a)


def spsa_minimize(loss_pair: Callable[[np.ndarray, np.ndarray], Tuple[float, float]], x0: np.ndarray, maxiter: int = 20,
                  rng=None, learning_rate: Optional[float] = None, perturbation: Optional[float] = None,
                  calibration_steps: int = 25, callback: Optional[Callable] = None):
    """Minimise with SPSA.  ``loss_pair(xp, xm)`` returns the two losses of one iteration."""
    rng = rng if rng is not None else np.random.default_rng()
    x = np.array(x0, dtype=np.float64)
    d = x.size
    alpha, gamma, c, target = 0.602, 0.101, 0.2 if perturbation is None else perturbation, 2 * np.pi / 10
    nfev = 0
    if learning_rate is None:
        mags = []
        for _ in range(calibration_steps):
            delta = rng.choice([-1.0, 1.0], size=d)
            fp, fm = loss_pair(x + c * delta, x - c * delta)
            nfev += 2
            mags.append(abs(fp - fm) / (2 * c))
        avg = float(np.mean(mags))
        a = target / avg if avg > 0 else target
        if a < 1e-10:
            a = target
    else:
        a = learning_rate
    history = []
    for k in range(maxiter):
        ak, ck = a / (k + 1) ** alpha, c / (k + 1) ** gamma
        delta = rng.choice([-1.0, 1.0], size=d)
        fp, fm = loss_pair(x + ck * delta, x - ck * delta)
        nfev += 2
        step = ak * (fp - fm) / (2 * ck) * delta
        x = x - step
        history.append({"nfev": nfev, "x": x.copy(), "fx": (fp + fm) / 2, "stepsize": float(np.linalg.norm(step))})
        if callback is not None:
            callback(nfev, x.copy(), (fp + fm) / 2, float(np.linalg.norm(step)), True)
    return x, history


def align_kernel(kernel: HipFidelityKernel, dataset: np.ndarray, labels: np.ndarray, c: float, maxiter: int = 20,
                 seed: Optional[int] = None):
    """Same arguments and return shape as reference ``align_kernel`` (qsvm_tester.py:236-280):
    ``(aligned_kernel, opt_data)`` with ``opt_data = [result dict, history]``."""
    from .qsvm import get_subsample

    if not kernel.training_parameters:
        raise ValueError("the kernel has no training parameters to align")
    dataset, labels = np.asarray(dataset), np.asarray(labels)
    if labels.shape[0] > 500:       # reference :260-261 (get_subsample with seed 42)
        dataset, labels = get_subsample(dataset, labels, 500, 42)
    fmap, theta_params = kernel.feature_map, kernel.training_parameters

    def loss_pair(tp, tm):
        maps = [fmap.assign_parameters(dict(zip(theta_params, t))) for t in (tp, tm)]
        Ks, _ = evaluate_population(maps, [dataset, dataset])
        return svc_loss(Ks[0], labels, c), svc_loss(Ks[1], labels, c)

    x0 = np.zeros(len(theta_params))
    theta, history = spsa_minimize(loss_pair, x0, maxiter=maxiter, rng=np.random.default_rng(seed))
    aligned = HipFidelityKernel(feature_map=fmap, training_parameters=theta_params, parameter_values=theta,
                                enforce_psd=kernel.enforce_psd)
    K0 = HipFidelityKernel(feature_map=fmap, training_parameters=theta_params, parameter_values=x0).evaluate(dataset)
    result = {"optimal_point": theta, "optimal_value": svc_loss(aligned.evaluate(dataset), labels, c),
              "initial_value": svc_loss(K0, labels, c), "optimizer_evals": history[-1]["nfev"] if history else 0,
              "quantum_kernel": aligned}
    return aligned, [result, history]
