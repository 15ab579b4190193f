"""CPU oracle for the model stage (numpy).  TEST INFRASTRUCTURE ONLY (see psk_oracle.c).

Restates what the reference asks of scikit-learn (pinned scikit-learn==0.22.1, not vendored):
  * LogisticRegression(penalty='l1', solver='liblinear')       modeling.py:1011-1014
      liblinear L1R_LR objective:  ||w||_1 + |b| + C * sum_i log(1 + exp(-y_i (w.x_i + b)))
      (the intercept is a constant-1 feature and is penalised; y in {-1,+1})
  * Lasso(alpha)                                               modeling.py:999-1000
      (1/2n) ||y - Xw - b||^2 + alpha ||w||_1, unpenalised intercept
  * GridSearchCV(model, {'C'|'alpha': grid}, cv=int)           modeling.py:1075-1085,1208-1216
      StratifiedKFold (classifier) / KFold (regressor), no shuffling; accuracy / R^2;
      best = first candidate with the highest mean test score; refit on everything.
Parity: pinned against converged scikit-learn 1.7.2 solutions in tests/golden/model_kat.npz
(objective value, intercept, linear predictor, CV folds and CV scores).  Raw per-column
coefficients are only unique up to identical columns.
"""
import ctypes

import numpy as np


def _clib():
    from . import oracle as _o
    L = _o.lib()
    if not getattr(L, "_model_sigs", False):
        c = ctypes
        L.orc_logreg_l1_fit.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_double, c.c_double, c.c_int,
                                        c.c_void_p, c.POINTER(c.c_double)]
        L.orc_logreg_l1_fit.restype = c.c_int
        L.orc_lasso_fit.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_double, c.c_double, c.c_int,
                                    c.c_void_p, c.POINTER(c.c_double)]
        L.orc_lasso_fit.restype = c.c_int
        L._model_sigs = True
    return L


def stratified_kfold(y, n_splits):
    """test-fold id per sample, as sklearn.model_selection.StratifiedKFold(n_splits) without
    shuffling assigns them (the 0.22+ allocation rule: classes encoded in order of first
    appearance, samples of the sorted class vector dealt round-robin to the folds)."""
    y = np.asarray(y)
    _, y_idx, y_inv = np.unique(y, return_index=True, return_inverse=True)
    _, class_perm = np.unique(y_idx, return_inverse=True)
    y_enc = class_perm[y_inv]
    n_classes = len(y_idx)
    y_order = np.sort(y_enc)
    allocation = np.asarray([np.bincount(y_order[i::n_splits], minlength=n_classes) for i in range(n_splits)])
    folds = np.empty(len(y), dtype=np.int64)
    for k in range(n_classes):
        folds[y_enc == k] = np.arange(n_splits).repeat(allocation[:, k])
    return folds


def kfold(n, n_splits):
    """sklearn KFold(n_splits) without shuffling: contiguous blocks, the first n % k one longer."""
    sizes = np.full(n_splits, n // n_splits)
    sizes[: n % n_splits] += 1
    return np.repeat(np.arange(n_splits), sizes)


def logreg_l1_objective(X, y01, w, b, C):
    z = X @ w + b
    ypm = 2.0 * np.asarray(y01, dtype=np.float64) - 1.0
    return np.abs(w).sum() + abs(b) + C * np.logaddexp(0.0, -ypm * z).sum()


def logreg_l1_fit(X, y01, C, tol=1e-10, max_sweeps=200000):
    """Cyclic coordinate descent with 1-D Newton steps and backtracking (the CDN scheme of
    Yuan et al. 2010, which liblinear's L1R_LR solver descends from) run to convergence.
    The loop itself is orc_logreg_l1_fit in psk_oracle.c."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.ascontiguousarray(y01, dtype=np.int32)
    n, p = X.shape
    w = np.zeros(p)
    b = ctypes.c_double()
    rc = _clib().orc_logreg_l1_fit(X.ctypes.data, y.ctypes.data, n, p, float(C), float(tol), int(max_sweeps),
                                   w.ctypes.data, ctypes.byref(b))
    if rc < 0:
        raise MemoryError
    return w, b.value


def logreg_predict(X, w, b):
    return (np.asarray(X) @ w + b > 0).astype(np.int64)


def lasso_fit(X, y, alpha, tol=1e-14, max_sweeps=1000000):
    """Cyclic coordinate descent on the centred problem (intercept unpenalised);
    the loop is orc_lasso_fit in psk_oracle.c."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    n, p = X.shape
    w = np.zeros(p)
    b = ctypes.c_double()
    rc = _clib().orc_lasso_fit(X.ctypes.data, y.ctypes.data, n, p, float(alpha), float(tol), int(max_sweeps),
                               w.ctypes.data, ctypes.byref(b))
    if rc < 0:
        raise MemoryError
    return w, b.value


def ridge_fit(X, y, alpha):
    """sklearn Ridge(alpha), fit_intercept=True, dense X (set_model, modeling.py:1001-1002): centre X and y by
    their column means, solve (Xc'Xc + alpha I) w = Xc'yc directly, intercept = ybar - xbar.w."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    xm, ym = X.mean(axis=0), y.mean()
    Xc, yc = X - xm, y - ym
    w = np.linalg.solve(Xc.T @ Xc + alpha * np.eye(X.shape[1]), Xc.T @ yc)
    return w, float(ym - xm @ w)


def logreg_l2_objective(X, y01, w, b, C, penalise_intercept=False):
    z = X @ w + b
    ypm = 2.0 * np.asarray(y01, dtype=np.float64) - 1.0
    return 0.5 * (w @ w + (b * b if penalise_intercept else 0.0)) + C * np.logaddexp(0.0, -ypm * z).sum()


def logreg_l2_fit(X, y01, C, penalise_intercept=False, iters=200):
    """LogisticRegression(penalty='l2') (set_model, modeling.py:1015-1019): 0.5 w'w + C sum log(1+exp(-y z)),
    z = Xw + b.  The intercept is free for lbfgs/newton-cg/sag/saga and a penalised constant feature for
    liblinear.  Exact Newton steps with halving until the gradient vanishes (strictly convex)."""
    X = np.asarray(X, dtype=np.float64)
    ypm = 2.0 * np.asarray(y01, dtype=np.float64) - 1.0
    n, p = X.shape
    A = np.hstack([X, np.ones((n, 1))])
    R = np.eye(p + 1)
    if not penalise_intercept:
        R[p, p] = 0.0
    th = np.zeros(p + 1)

    def f(t):
        return 0.5 * t @ R @ t + C * np.logaddexp(0.0, -ypm * (A @ t)).sum()
    for _ in range(iters):
        z = A @ th
        s = 1.0 / (1.0 + np.exp(ypm * z))
        g = R @ th - C * (A.T @ (ypm * s))
        if np.abs(g).max() < 1e-13 * max(1.0, C):
            break
        H = R + C * (A.T * (s * (1 - s))) @ A
        d = np.linalg.solve(H, -g)
        t, f0 = 1.0, f(th)
        while f(th + t * d) > f0 + 1e-4 * t * (g @ d) and t > 1e-12:
            t *= 0.5
        th = th + t * d
    return th[:p].copy(), float(th[p])


def r2_score(y, pred):
    y = np.asarray(y, dtype=np.float64)
    ss_res = ((y - pred) ** 2).sum()
    ss_tot = ((y - y.mean()) ** 2).sum()
    return 1.0 - ss_res / ss_tot


def grid_search(X, y, grid, kind, cv):
    """GridSearchCV restatement.  kind: 'logreg' | 'logreg_l2' | 'logreg_l2_liblinear' (grid of C) or
    'lasso' | 'ridge' (grid of alpha).
    Returns dict(mean_test_score, std_test_score, best_index, coef, intercept)."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y)
    is_clf = kind.startswith("logreg")
    fit = {"logreg": logreg_l1_fit, "logreg_l2": logreg_l2_fit,
           "logreg_l2_liblinear": lambda A, t, g: logreg_l2_fit(A, t, g, True),
           "lasso": lasso_fit, "ridge": ridge_fit}[kind]
    folds = stratified_kfold(y, cv) if is_clf else kfold(len(y), cv)
    scores = np.zeros((len(grid), cv))
    for gi, g in enumerate(grid):
        for f in range(cv):
            tr, te = folds != f, folds == f
            w, b = fit(X[tr], y[tr], g)
            if is_clf:
                scores[gi, f] = (logreg_predict(X[te], w, b) == y[te]).mean()
            else:
                scores[gi, f] = r2_score(y[te], X[te] @ w + b)
    mean = scores.mean(axis=1)
    best = int(np.argmax(mean))  # first maximum, as rank_test_score.argmin() picks
    w, b = fit(X, y, grid[best])
    return {"mean_test_score": mean, "std_test_score": scores.std(axis=1), "best_index": best, "coef": w,
            "intercept": b, "folds": folds}


def collapse_columns(X):
    """distinct column patterns of X in order of first appearance: (Xu, group) with X[:, j] == Xu[:, group[j]]"""
    X = np.asarray(X, dtype=np.float64)
    seen, group, keep = {}, np.empty(X.shape[1], dtype=np.int64), []
    for j in range(X.shape[1]):
        key = X[:, j].tobytes()
        if key not in seen:
            seen[key] = len(keep)
            keep.append(j)
        group[j] = seen[key]
    return X[:, keep], group


def logreg_l1_arbiter(X, y01, C, w_hint, b_hint, support_tol=1e-9, max_rounds=50):
    """The exact optimum of liblinear's L1R_LR objective (modeling.py:1011-1014) by an active-set Newton solve --
    the ARBITER between two approximate solvers (the HIP solver and the liblinear fixture), better than either:
    identical columns are collapsed to one (their coefficient SUM is what is unique), the support and the signs
    are taken from the hint, the smooth problem  s.theta_A + C sum log(1 + exp(-y A theta))  is solved on that
    support by damped Newton steps in f64 until its gradient is below 2e-14 (or no step improves it), and the KKT conditions of
    the full problem are then checked: sign agreement on the support, |grad_j| <= 1 off it.  A violated
    condition moves the coordinate in / out of the support and the solve repeats.
    Returns dict(w_groups (per distinct pattern), group (column -> pattern), b, kkt (max violation),
    rank_deficient (the support's columns are dependent: only A.theta is unique then))."""
    Xu, group = collapse_columns(X)
    n, pu = Xu.shape
    A = np.hstack([Xu, np.ones((n, 1))])        # the intercept is a penalised constant feature
    ypm = 2.0 * np.asarray(y01, dtype=np.float64) - 1.0
    th = np.zeros(pu + 1)
    np.add.at(th, group, np.asarray(w_hint, dtype=np.float64))
    th[pu] = b_hint
    act = np.abs(th) > support_tol
    sgn = np.sign(th)
    th[~act] = 0.0

    def grad_loss(t):
        z = A @ t
        s = 1.0 / (1.0 + np.exp(ypm * z))       # sigma(-y z)
        return -C * (A.T @ (ypm * s)), s

    def obj(t):
        return np.abs(t).sum() + C * np.logaddexp(0.0, -ypm * (A @ t)).sum()
    rank_def = False
    for _ in range(max_rounds):
        idx = np.nonzero(act)[0]
        for _it in range(200):
            g, s = grad_loss(th)
            gs = g[idx] + sgn[idx]
            if idx.size == 0 or np.abs(gs).max() < 2e-14:
                break
            Aa = A[:, idx]
            H = C * (Aa.T * (s * (1.0 - s))) @ Aa
            try:
                d = -np.linalg.solve(H, gs)
                if not np.all(np.isfinite(d)):
                    raise np.linalg.LinAlgError
            except np.linalg.LinAlgError:
                rank_def = True
                d = -np.linalg.lstsq(H, gs, rcond=1e-13)[0]
            # the step may not carry a coordinate across zero (the smooth model is only valid on this orthant)
            t = 1.0
            cross = (th[idx] + d) * sgn[idx] < 0
            if cross.any():
                t = min(1.0, float(np.min(-th[idx][cross] / d[cross])))
            f0 = obj(th)
            while t > 1e-14:
                cand = th.copy()
                cand[idx] += t * d
                if obj(cand) <= f0 + 1e-4 * t * (gs @ d) or t * np.abs(d).max() < 1e-15:
                    break
                t *= 0.5
            th = cand
            hit = idx[np.abs(th[idx]) < 1e-15 * max(1.0, np.abs(th).max())]
            if hit.size and cross.any():
                th[hit] = 0.0
                act[hit] = False
                idx = np.nonzero(act)[0]
        g, _ = grad_loss(th)
        viol_in = (~act) & (np.abs(g) > 1.0 + 1e-12)
        if not viol_in.any():
            break
        j = int(np.argmax(np.where(viol_in, np.abs(g), 0.0)))
        act[j] = True
        sgn[j] = -np.sign(g[j])
    g, _ = grad_loss(th)
    kkt = max(float(np.abs(g[act] + np.sign(th[act])).max()) if act.any() else 0.0,
              float(np.maximum(np.abs(g[~act]) - 1.0, 0.0).max()) if (~act).any() else 0.0)
    if act.any() and act.sum() > np.linalg.matrix_rank(A[:, act]):
        rank_def = True
    return {"w_groups": th[:pu].copy(), "group": group, "b": float(th[pu]), "kkt": kkt, "rank_deficient": rank_def,
            "objective": obj(th), "linpred": A @ th}
