#!/usr/bin/env python3
"""Grid search (13 C x 10 folds + refits) on random sparse 0/1 designs with a planted signal: GPU solver vs
scikit-learn/liblinear run serially on the host.  usage: tools/solver_random.py N P [density]"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.engine import PskContext  # noqa: E402
from phenotypeseeker_amd.model import GridSearch, L1LogisticRegression  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n, p = int(args[0]), int(args[1])
dens = float(args[2]) if len(args) > 2 else 0.3
rng = np.random.default_rng(1)
X = (rng.random((n, p)) < dens).astype(np.float64)
logit = 2.5 * X[:, 0] - 2.0 * X[:, 1] + 1.5 * X[:, 2] + 1.0 * X[:, 3] - 1.0
y = (rng.random(n) < 1 / (1 + np.exp(-logit))).astype(int)
Cs = [1 / a for a in np.logspace(-3, 3, 13)]
with PskContext(0) as ctx:
    GridSearch(L1LogisticRegression(tol=1e-4, max_iter=1000), "C", Cs, 10).fit(X[:, :8], y, ctx)
    t = time.time()
    gs = GridSearch(L1LogisticRegression(tol=1e-4, max_iter=1000), "C", Cs, 10).fit(X, y, ctx)
    dt = time.time() - t
print("GPU grid search: %.3f s  best C %.4g  score %.3f  newton max %d  unique cols %d" % (
    dt, gs.best_params_["C"], gs.best_score_, int(gs.n_iter_.max()), gs.n_unique_columns_))
try:
    from sklearn.linear_model import LogisticRegression
    from sklearn.model_selection import GridSearchCV
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t = time.time()
        sk = GridSearchCV(LogisticRegression(penalty="l1", solver="liblinear", tol=1e-4, max_iter=1000), {"C": Cs}, cv=10).fit(X, y)
        print("sklearn GridSearchCV (1 core): %.3f s  best C %.4g  score %.3f" % (time.time() - t, sk.best_params_["C"], sk.best_score_))
        print("mean_test_score max abs diff: %.4f" % np.abs(sk.cv_results_["mean_test_score"] - gs.cv_results_["mean_test_score"]).max())
except ImportError:
    pass
if "--per-fit" in sys.argv:
    from phenotypeseeker_amd import cv as _cv
    folds = _cv.stratified_kfold(y, 10).astype(np.int32)
    with PskContext(0) as ctx:
        ctx.logreg_l1_fit(X[:, :8], y, folds, [1.0], [0], 1e-4, 1000)
        for C in (1000.0, 31.6, 1.0, 0.1):
            t = time.time()
            c, b, it = ctx.logreg_l1_fit(X, y, folds, [C], [-1], 1e-4, 1000)
            dt = time.time() - t
            t = time.time()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m = LogisticRegression(penalty="l1", solver="liblinear", C=C, tol=1e-4, max_iter=1000).fit(X, y)
            ds = time.time() - t
            print("C=%-7g GPU %.3f s newton %3d nnz %4d | liblinear %.3f s newton %3d nnz %4d" % (
                C, dt, it[0], (c[0] != 0).sum(), ds, m.n_iter_[0], (m.coef_ != 0).sum()))
