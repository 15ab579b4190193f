import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from phenotypeseeker_amd.engine import PskContext
d = np.load("tools/data/fitw.npz")
X, y, fold, fp, ff = d["X"], d["y"], d["fold"], d["fit_param"], d["fit_fold"]
ypm = 2.0 * y - 1.0
with PskContext(0) as ctx:
    idx = 50
    C, f = float(fp[idx]), int(ff[idx])
    tr = fold != f
    c, b, it = ctx.logreg_l1_fit(X, y, fold, [C], [f], 1e-4, 30)
    A = np.hstack([X[tr], np.ones((tr.sum(), 1))]); th = np.append(c[0], b[0])
    z = A @ th
    g = -C * (A.T @ (ypm[tr] / (1 + np.exp(ypm[tr] * z))))
    v = np.where(th > 0, np.abs(g + 1), np.where(th < 0, np.abs(g - 1), np.maximum(0, np.maximum(-(g + 1), g - 1))))
    j = int(np.argmax(v))
    print("p =", X.shape[1], "violating feature", j, "w_j", th[j], "grad", g[j], "viol", v[j], "col sum (train)", A[:, j].sum(), "of", tr.sum())
    print("top violations", np.sort(v)[-5:], "intercept", th[-1], "g_icpt", g[-1])
    same = [k for k in range(X.shape[1]) if np.array_equal(X[:, k], X[:, j]) and k != j] if j < X.shape[1] else []
    print("duplicates of it in X:", same)
    if j < X.shape[1]:
        sameT = [k for k in range(X.shape[1]) if np.array_equal(X[tr, k], X[tr, j]) and k != j]
        print("identical on the training rows:", sameT, [th[k] for k in sameT], [g[k] for k in sameT])
