#!/usr/bin/env python3
"""Lasso grid search (13 alphas x 10 folds + refits) on random 0/1 designs: GPU solver vs scikit-learn.
usage: tools/lasso_random.py N P [density]"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.engine import PskContext  # noqa: E402
from phenotypeseeker_amd.model import GridSearch, LassoRegression  # noqa: E402

n, p = int(sys.argv[1]), int(sys.argv[2])
dens = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
rng = np.random.default_rng(1)
X = (rng.random((n, p)) < dens).astype(np.float64)
y = 2.5 * X[:, 0] - 2.0 * X[:, 1] + 1.5 * X[:, 2] + 1.0 * X[:, 3] + rng.normal(0, 0.5, n) + 3.0
alphas = [float(a) for a in np.logspace(-3, 3, 13)]
with PskContext(0) as ctx:
    GridSearch(LassoRegression(tol=1e-4, max_iter=1000), "alpha", alphas, 10).fit(X[:, :8], y, ctx)
    t = time.time()
    gs = GridSearch(LassoRegression(tol=1e-4, max_iter=1000), "alpha", alphas, 10).fit(X, y, ctx)
    dt = time.time() - t
print("GPU grid search: %.3f s  best alpha %.4g  score %.4f  sweeps max %d" % (dt, gs.best_params_["alpha"], gs.best_score_, int(gs.n_iter_.max())))
try:
    from sklearn.linear_model import Lasso
    from sklearn.model_selection import GridSearchCV
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t = time.time()
        sk = GridSearchCV(Lasso(tol=1e-4, max_iter=1000), {"alpha": alphas}, cv=10).fit(X, y)
        print("sklearn GridSearchCV (1 core): %.3f s  best alpha %.4g  score %.4f" % (time.time() - t, sk.best_params_["alpha"], sk.best_score_))
        print("mean_test_score max abs diff: %.2e" % np.abs(sk.cv_results_["mean_test_score"] - gs.cv_results_["mean_test_score"]).max())
except ImportError:
    pass
