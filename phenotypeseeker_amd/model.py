"""The estimator boundary: picklable stand-ins for what the reference stores in
its .pkl -- a fitted GridSearchCV over LogisticRegression(penalty='l1', solver='liblinear') or
Lasso (modeling.py:994-1014, :1075-1085, :1208-1216, :975-979) -- exposing the attributes the
reference reads: predict / predict_proba / score, cv_results_['mean_test_score' |
'std_test_score' | 'params'], best_params_, best_estimator_.coef_ (modeling.py:1226-1247,
:1427-1436; prediction.py:126-129,:168-172).  All fits of a grid search (grid x folds + the
refit) are solved on the GPU in one psk_logreg_l1_fit / psk_lasso_fit launch.  `--penalty L2`
(modeling.py:1001-1002, :1015-1019) maps to RidgeRegression / L2LogisticRegression over
psk_ridge_fit / psk_logreg_l2_fit in the same way.
"""
import numpy as np

from . import cv as _cv


class L1LogisticRegression:
    """liblinear-style L1 logistic regression: ||w||_1 + |b| + C sum log(1+exp(-y(w.x+b)))."""
    _is_classifier = True
    _dedupe = True  # an L1 optimum may sit on any copy of a repeated column: solve each pattern once

    def __init__(self, C=1.0, tol=1e-4, max_iter=1000):
        self.C, self.tol, self.max_iter = C, tol, max_iter
        self.penalty, self.solver = "l1", "liblinear"
        self.classes_ = np.array([0, 1])
        self.coef_ = None
        self.intercept_ = None

    def __repr__(self):
        # scikit-learn prints only the parameters that differ from its defaults, alphabetically
        parts = []
        if self.C != 1.0:
            parts.append("C=%r" % self.C)
        if repr(self.max_iter) != "100":
            parts.append("max_iter=%r" % self.max_iter)
        parts += ["penalty='l1'", "solver='liblinear'"]
        if self.tol != 1e-4:
            parts.append("tol=%r" % self.tol)
        return "LogisticRegression(%s)" % ", ".join(parts)

    def _clone(self, **params):
        return type(self)(**params)

    def _engine_fit(self, ctx, X, y, folds, fit_param, fit_fold):
        return ctx.logreg_l1_fit(X, y.astype(np.int32), folds, fit_param, fit_fold, self.tol, int(self.max_iter))

    def _set(self, coef, icpt):
        self.coef_ = np.asarray(coef, dtype=np.float64).reshape(1, -1)
        self.intercept_ = np.array([float(icpt)])
        self.n_features_in_ = self.coef_.shape[1]
        return self

    def decision_function(self, X):
        return np.asarray(X, dtype=np.float64) @ self.coef_[0] + self.intercept_[0]

    def predict(self, X):
        return self.classes_[(self.decision_function(X) > 0).astype(int)]

    def predict_proba(self, X):
        p1 = 1.0 / (1.0 + np.exp(-self.decision_function(X)))
        return np.column_stack([1.0 - p1, p1])

    def score(self, X, y):
        return np.float64(np.mean(self.predict(X) == np.asarray(y)))


class L2LogisticRegression(L1LogisticRegression):
    """0.5 w'w + C sum log(1+exp(-y(w.x+b))): LogisticRegression(penalty='l2', solver=...)
    (modeling.py:1015-1019).  The intercept is free for lbfgs / newton-cg / sag / saga and a penalised
    constant feature for liblinear (get_logreg_solver, modeling.py:256-264); the optimum is unique
    either way, so one Newton solver serves all five names."""
    _dedupe = False

    def __init__(self, C=1.0, tol=1e-4, max_iter=1000, solver="lbfgs"):
        super().__init__(C, tol, max_iter)
        self.penalty, self.solver = "l2", solver

    def __repr__(self):
        parts = []
        if self.C != 1.0:
            parts.append("C=%r" % self.C)
        if repr(self.max_iter) != "100":
            parts.append("max_iter=%r" % self.max_iter)
        if self.solver != "lbfgs":
            parts.append("solver=%r" % self.solver)
        if self.tol != 1e-4:
            parts.append("tol=%r" % self.tol)
        return "LogisticRegression(%s)" % ", ".join(parts)

    def _clone(self, **params):
        return type(self)(solver=self.solver, **params)

    def _engine_fit(self, ctx, X, y, folds, fit_param, fit_fold):
        return ctx.logreg_l2_fit(X, y.astype(np.int32), folds, fit_param, fit_fold, self.tol, int(self.max_iter),
                                 self.solver == "liblinear")


class LassoRegression:
    """(1/2n)||y - Xw - b||^2 + alpha ||w||_1."""
    _is_classifier = False
    # scikit-learn's cyclic descent visits every column, copies included, and a fit that ends at the sweep limit has walked a
    # path the copies were part of (r04: the grid search of a 1,024 x 907 design was 2e-3 off in R^2 on its unconverged row
    # with the copies removed).  Up to 1,024 columns -- the covariance form of solver_lasso.hip -- every column is kept;
    # beyond that (--n_kmers 0 models) each distinct pattern is solved once: the same optimum where the descent converges.
    _dedupe = True
    _dedupe_above = 1024
    _sk_name = "Lasso"

    def __init__(self, alpha=1.0, tol=1e-4, max_iter=1000):
        self.alpha, self.tol, self.max_iter = alpha, tol, max_iter
        self.coef_ = None
        self.intercept_ = None

    def __repr__(self):
        parts = []
        if self.alpha != 1.0:
            parts.append("alpha=%r" % self.alpha)
        if repr(self.max_iter) != "1000":
            parts.append("max_iter=%r" % self.max_iter)
        if self.tol != 1e-4:
            parts.append("tol=%r" % self.tol)
        return "%s(%s)" % (self._sk_name, ", ".join(parts))

    def _clone(self, **params):
        return type(self)(**params)

    def _engine_fit(self, ctx, X, y, folds, fit_param, fit_fold):
        return ctx.lasso_fit(X, y.astype(np.float64), folds, fit_param, fit_fold, self.tol, int(self.max_iter))

    def _set(self, coef, icpt):
        self.coef_ = np.asarray(coef, dtype=np.float64).ravel()
        self.intercept_ = float(icpt)
        self.n_features_in_ = len(self.coef_)
        return self

    def predict(self, X):
        return np.asarray(X, dtype=np.float64) @ self.coef_ + self.intercept_

    def score(self, X, y):
        """R^2 as sklearn.metrics.r2_score returns it (what GridSearchCV scores a regressor with, modeling.py:1208-1216):
        undefined -- nan -- on fewer than two samples (a 10-fold split of fewer than 20 samples has such folds: the
        reference's summary then prints `nan (+/-nan)` for every alpha); a constant target scores 1.0 when it is predicted
        exactly and 0.0 otherwise (force_finite)."""
        y = np.asarray(y, dtype=np.float64)
        if len(y) < 2:
            return np.float64(np.nan)
        res = ((y - self.predict(X)) ** 2).sum()
        tot = ((y - y.mean()) ** 2).sum()
        if tot == 0.0:
            return np.float64(1.0 if res == 0.0 else 0.0)
        return np.float64(1.0 - res / tot)


class RidgeRegression(LassoRegression):
    """||y - Xw - b||^2 + alpha ||w||^2: sklearn Ridge (modeling.py:1001-1002).  max_iter / tol are
    carried for the repr only -- the dense solve is direct in scikit-learn and converged here."""
    _dedupe = False
    _sk_name = "Ridge"

    def __repr__(self):
        parts = []
        if self.alpha != 1.0:
            parts.append("alpha=%r" % self.alpha)
        if self.max_iter is not None:
            parts.append("max_iter=%r" % self.max_iter)
        if self.tol != 1e-3:
            parts.append("tol=%r" % self.tol)
        return "Ridge(%s)" % ", ".join(parts)

    def _engine_fit(self, ctx, X, y, folds, fit_param, fit_fold):
        return ctx.ridge_fit(X, y.astype(np.float64), folds, fit_param, fit_fold)


class GridSearch:
    """GridSearchCV(model, {'C' | 'alpha': grid}, cv=int) with refit.  `engine_ctx` is only needed
    by fit(); the fitted object pickles without it."""

    def __init__(self, estimator, param_name, grid, cv):
        self.estimator = estimator
        self.param_name = param_name
        self.param_grid = {param_name: list(grid)}
        self.cv = int(cv)

    def fit(self, X, y, engine_ctx):
        X_full = np.asarray(X, dtype=np.float64)
        y = np.asarray(y)
        n = len(y)
        # Identical columns (k-mers of one gene share a presence pattern) are solved once: an L1
        # optimum may place a pattern's weight on any of its copies, here on the first (which is also
        # what cyclic coordinate descent -- scikit-learn's Lasso -- does).
        if self.estimator._dedupe and X_full.shape[1] > getattr(self.estimator, "_dedupe_above", 0):
            X, first, inverse = _unique_columns(X_full)
        else:  # an L2 optimum spreads a pattern's weight over its copies: every column stays
            X, first = X_full, np.arange(X_full.shape[1])
        grid = self.param_grid[self.param_name]
        is_clf = self.estimator._is_classifier
        if self.cv < 2:
            raise ValueError("k-fold cross-validation requires at least one train/test split by setting "
                             "n_splits=2 or more, got n_splits=%d." % self.cv)
        folds = _cv.stratified_kfold(y, self.cv) if is_clf else _cv.kfold(n, self.cv)
        fit_param, fit_fold = [], []
        for g in grid:
            for f in range(self.cv):
                fit_param.append(float(g))
                fit_fold.append(f)
        for g in grid:  # refit candidates on everything: pick after scoring, all in one launch
            fit_param.append(float(g))
            fit_fold.append(-1)
        coef, icpt, iters = self.estimator._engine_fit(engine_ctx, X, y, folds, fit_param, fit_fold)
        scores = np.zeros((len(grid), self.cv))
        for gi in range(len(grid)):
            for f in range(self.cv):
                j = gi * self.cv + f
                te = folds == f
                est = self.estimator._clone(**{self.param_name: grid[gi]})._set(coef[j], icpt[j])
                scores[gi, f] = est.score(X[te], y[te])
        mean = scores.mean(axis=1)
        std = scores.std(axis=1)
        self.cv_results_ = {"mean_test_score": mean, "std_test_score": std,
                            "params": [{self.param_name: g} for g in grid],
                            "rank_test_score": _rank_with_nan(mean)}
        for f in range(self.cv):
            self.cv_results_["split%d_test_score" % f] = scores[:, f]
        self.best_index_ = int(np.argmin(self.cv_results_["rank_test_score"]))
        self.best_params_ = {self.param_name: grid[self.best_index_]}
        self.best_score_ = float(mean[self.best_index_])
        j = len(grid) * self.cv + self.best_index_
        best = self.estimator._clone(**{self.param_name: grid[self.best_index_]})
        best.tol, best.max_iter = self.estimator.tol, self.estimator.max_iter
        full = np.zeros(X_full.shape[1])
        full[first] = coef[j]
        self.best_estimator_ = best._set(full, icpt[j])
        self.n_unique_columns_ = int(X.shape[1])
        self.n_splits_ = self.cv
        self.n_iter_ = iters
        self.test_folds_ = folds
        return self

    def predict(self, X):
        return self.best_estimator_.predict(X)

    def predict_proba(self, X):
        return self.best_estimator_.predict_proba(X)

    def score(self, X, y):
        return self.best_estimator_.score(X, y)

    def to_sklearn_shell(self):
        """The same model as to_sklearn() builds, described for skpickle.dumps (no scikit-learn import), or None when
        the installed scikit-learn has no template."""
        from . import skpickle as sp
        be = self.best_estimator_
        if isinstance(be, L1LogisticRegression):
            kw = dict(penalty=be.penalty, solver=be.solver, tol=be.tol, max_iter=int(be.max_iter))
            est = sp.make("LogisticRegression", C=be.C, classes_=np.array([0, 1]), coef_=be.coef_.copy(),
                          intercept_=be.intercept_.copy(), n_iter_=np.array([0], dtype=np.int32),
                          n_features_in_=be.n_features_in_, **kw)
            proto = sp.make("LogisticRegression", **kw)
        else:
            name = "Ridge" if isinstance(be, RidgeRegression) else "Lasso"
            kw = dict(tol=be.tol, max_iter=int(be.max_iter))
            est = sp.make(name, alpha=be.alpha, coef_=be.coef_.copy(), intercept_=be.intercept_,
                          n_iter_=None if name == "Ridge" else 0, n_features_in_=be.n_features_in_, **kw)
            proto = sp.make(name, **kw)
        if est is None or proto is None:
            return None
        return sp.make("GridSearchCV", estimator=proto, param_grid=self.param_grid, cv=self.cv, best_estimator_=est,
                       best_params_=dict(self.best_params_), best_index_=self.best_index_, best_score_=self.best_score_,
                       cv_results_=dict(self.cv_results_), n_splits_=self.n_splits_, refit_time_=0.0, multimetric_=False,
                       scorer_=None)

    def to_sklearn(self):
        """The same fitted model as real scikit-learn objects (for users whose downstream code
        insists on them); needs scikit-learn importable."""
        from sklearn.linear_model import Lasso, LogisticRegression, Ridge
        from sklearn.model_selection import GridSearchCV
        be = self.best_estimator_
        if isinstance(be, L1LogisticRegression):
            kw = dict(penalty=be.penalty, solver=be.solver, tol=be.tol, max_iter=int(be.max_iter))
            est = LogisticRegression(C=be.C, **kw)
            est.classes_ = np.array([0, 1])
            est.coef_, est.intercept_ = be.coef_.copy(), be.intercept_.copy()
            est.n_iter_ = np.array([0], dtype=np.int32)
            proto = LogisticRegression(**kw)
        else:
            cls = Ridge if isinstance(be, RidgeRegression) else Lasso
            est = cls(alpha=be.alpha, tol=be.tol, max_iter=int(be.max_iter))
            est.coef_, est.intercept_ = be.coef_.copy(), be.intercept_
            est.n_iter_ = None if cls is Ridge else 0
            proto = cls(tol=be.tol, max_iter=int(be.max_iter))
        est.n_features_in_ = be.n_features_in_
        gs = GridSearchCV(proto, self.param_grid, cv=self.cv)
        gs.best_estimator_, gs.best_params_ = est, dict(self.best_params_)
        gs.best_index_, gs.best_score_ = self.best_index_, self.best_score_
        gs.cv_results_ = dict(self.cv_results_)
        gs.n_splits_, gs.refit_time_, gs.multimetric_ = self.n_splits_, 0.0, False
        gs.scorer_ = None
        return gs


def _unique_columns(X):
    """Distinct columns of X in order of first appearance: (X_unique, first_index[], inverse[])."""
    seen = {}
    first, inverse = [], np.empty(X.shape[1], dtype=np.int64)
    cols = np.ascontiguousarray(X.T)
    for j in range(X.shape[1]):
        key = cols[j].tobytes()
        u = seen.get(key)
        if u is None:
            u = len(first)
            seen[key] = u
            first.append(j)
        inverse[j] = u
    first = np.array(first, dtype=np.int64)
    return np.ascontiguousarray(X[:, first]) if len(first) else X[:, :0], first, inverse


def _rank_with_nan(mean):
    """GridSearchCV's rank_test_score (sklearn/model_selection/_search.py::_store): rank 1 for everything when every mean
    is nan (the first candidate is then the best one), else nan counts as worse than the worst finite mean."""
    mean = np.asarray(mean, dtype=np.float64)
    if np.isnan(mean).all():
        return np.ones(len(mean), dtype=np.int32)
    return _rank_min(-np.nan_to_num(mean, nan=np.nanmin(mean) - 1.0))


def _rank_min(a):
    """scipy.stats.rankdata(a, method='min') for a 1-D array."""
    a = np.asarray(a)
    order = np.argsort(a, kind="stable")
    ranks = np.empty(len(a), dtype=np.int32)
    r = 0
    for pos, idx in enumerate(order):
        if pos == 0 or a[idx] != a[order[pos - 1]]:
            r = pos + 1
        ranks[idx] = r
    return ranks
