import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
from phenotypeseeker_amd.engine import PskContext
d = np.load("tools/data/fitw.npz")
X, y, fold, fp, ff = d["X"], d["y"], d["fold"], d["fit_param"], d["fit_fold"]
ypm = 2.0 * y - 1.0
def obj(w, b, C, tr):
    z = X[tr] @ w + b
    return np.abs(w).sum() + abs(b) + C * np.logaddexp(0, -ypm[tr] * z).sum()
def viol(w, b, C, tr):
    A = np.hstack([X[tr], np.ones((tr.sum(), 1))]); th = np.append(w, b)
    z = A @ th
    g = -C * (A.T @ (ypm[tr] / (1 + np.exp(ypm[tr] * z))))
    v = np.where(th > 0, np.abs(g + 1), np.where(th < 0, np.abs(g - 1), np.maximum(0, np.maximum(-(g + 1), g - 1))))
    return v.sum(), v.max()
with PskContext(0) as ctx:
    for idx in (50, 53):
        C, f = float(fp[idx]), int(ff[idx])
        tr = fold != f
        v0 = viol(np.zeros(X.shape[1]), 0.0, C, tr)[0]
        for mi in (5, 10, 15, 20, 25, 30, 60):
            c, b, it = ctx.logreg_l1_fit(X, y, fold, [C], [f], 1e-4, mi)
            vs, vm = viol(c[0], b[0], C, tr)
            print("fit %d max_iter=%3d newton=%3d nnz=%3d obj=%.10f viol1=%.3e (rel %.3e, need <= %.3e) max=%.2e" % (
                idx, mi, it[0], (c[0] != 0).sum(), obj(c[0], b[0], C, tr), vs, vs / v0, 1e-4 * min((ypm[tr] > 0).sum(), (ypm[tr] < 0).sum()) / tr.sum(), vm), flush=True)
