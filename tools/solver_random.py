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

n, p = int(sys.argv[1]), int(sys.argv[2])
dens = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
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
