#!/usr/bin/env python3
"""The L1-logistic grid of a 2,048-genome run whose 1,000 selected k-mers have 907 distinct patterns (tests/golden/fit2048_907.npz:
143 fits, the register form of the descent): wall-clock, Newton steps, and the objective every fit reached -- with the
library as built, and -- for A/B runs of an experimental descent -- with the variable named by PSK_PROBE_OFF_VAR set to 1.
usage: tools/solver_grid_probe.py [new|old|both]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.engine import PskContext  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "new"
d = np.load(os.path.join(ROOT, "tests", "golden", "fit2048_907.npz"))
X = np.unpackbits(d["Xbits"], axis=1)[:, : int(d["p"])].astype(np.float32)
y, fold, fp, ff = d["y"], d["fold"], d["fit_param"], d["fit_fold"]
ypm = 2.0 * y - 1.0


def objectives(coef, icpt):
    out = np.zeros(len(fp))
    Z = X.astype(np.float64) @ coef.T + icpt[None, :]
    for j in range(len(fp)):
        tr = fold != ff[j]
        out[j] = np.abs(coef[j]).sum() + abs(icpt[j]) + fp[j] * np.logaddexp(0.0, -ypm[tr] * Z[tr, j]).sum()
    return out


res = {}
OFF = os.environ.get("PSK_PROBE_OFF_VAR", "PSK_EXPERIMENT_OFF")
for tag in (["new", "old"] if mode == "both" else [mode]):
    if tag == "old":
        os.environ[OFF] = "1"
    else:
        os.environ.pop(OFF, None)
    with PskContext(0) as ctx:
        ctx.logreg_l1_fit(X[:, :50], y, fold, fp[:2], ff[:2], 1e-4, 50)       # code objects, buffers
        t0 = time.time()
        coef, icpt, iters = ctx.logreg_l1_fit(X, y, fold, fp, ff, float(d["tol"]), int(d["max_iter"]))
        wall = time.time() - t0
    obj = objectives(coef, icpt)
    res[tag] = obj
    print("%-5s wall %.3f s  Newton steps max %d mean %.1f  objective sum %.6e  nnz mean %.0f" % (
        tag, wall, iters.max(), iters.mean(), obj.sum(), (coef != 0).sum(axis=1).mean()), flush=True)
if len(res) == 2:
    rel = (res["new"] - res["old"]) / res["old"]
    print("objective new vs old: worst +%.2e best %.2e (relative; negative = the new form ends lower)" % (rel.max(), rel.min()))
