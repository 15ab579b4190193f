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
with PskContext(0) as ctx:
    for idx in (50, 51, 53, 54):
        C, f = float(fp[idx]), int(ff[idx])
        for mi in (20, 40, 1000):
            t = time.time()
            c, b, it = ctx.logreg_l1_fit(X, y, fold, [C], [f], 1e-4, mi)
            tr = fold != f
            print("fit %d C=%r fold=%d max_iter=%4d newton=%4d nnz=%3d obj=%.10f  %.3fs" % (idx, C, f, mi, it[0], (c[0] != 0).sum(), obj(c[0], b[0], C, tr), time.time() - t), flush=True)
    # the whole batch, twice
    for rep in range(2):
        c, b, it = ctx.logreg_l1_fit(X, y, fold, fp, ff, 1e-4, 1000)
        print("batch newton>100:", [(i, int(it[i])) for i in range(len(it)) if it[i] > 100])
    # pairs
    c, b, it = ctx.logreg_l1_fit(X, y, fold, fp[50:52], ff[50:52], 1e-4, 1000)
    print("pair 50,51:", it.tolist())
    c, b, it = ctx.logreg_l1_fit(X, y, fold, fp[40:60], ff[40:60], 1e-4, 1000)
    print("40..59:", it.tolist())
