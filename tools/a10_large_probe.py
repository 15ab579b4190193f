#!/usr/bin/env python3
"""VERDICT r03 #1 probe: the fits of tests/golden/model_large_kat.npz through psk_logreg_l1_fit at a given tolerance, per
form: wall-clock, Newton steps, and the distance of every fit from the certified optimum (objective, linear predictor,
coefficient sums).  usage: tools/a10_large_probe.py TAG TOL MAX_ITER [default|arrays]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import GOLDEN, large_design  # noqa: E402
from phenotypeseeker_amd.engine import PskContext  # noqa: E402

tag, tol, mi = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
form = sys.argv[4] if len(sys.argv) > 4 else "default"
if form == "arrays":
    os.environ["PSK_NO_GRAM_GLOBAL"] = "1"
z = np.load(os.path.join(GOLDEN, "model_large_kat.npz"))
X, y, fold = large_design(z, tag)
fp, ff = z["fit_C_" + tag].astype(np.float64), z["fit_held_" + tag].astype(np.int32)
sel = [int(a) for a in sys.argv[5].split(",")] if len(sys.argv) > 5 else list(range(len(fp)))
with PskContext(0) as ctx:
    ctx.logreg_l1_fit(X[:, :50], y, fold, fp[:2], ff[:2], 1e-4, 50)
    t0 = time.time()
    coef, icpt, iters = ctx.logreg_l1_fit(X, y, fold, fp[sel], ff[sel], tol=tol, max_iter=mi)
    wall = time.time() - t0
print("%s %s tol %g: %.2f s" % (tag, form, tol, wall))
Xd, ypm = X.astype(np.float64), 2.0 * y - 1.0
for q, j in enumerate(sel):
    tr = fold != ff[j]
    aw, ab, grp = z["arb_coef_" + tag][j], float(z["arb_icpt_" + tag][j]), z["arb_group_" + tag][j]
    lin, alin = Xd[tr] @ coef[q] + icpt[q], Xd[tr] @ aw + ab
    obj = np.abs(coef[q]).sum() + abs(icpt[q]) + fp[j] * np.logaddexp(0.0, -ypm[tr] * lin).sum()
    sums, asums = np.zeros(grp.max() + 1), np.zeros(grp.max() + 1)
    np.add.at(sums, grp, coef[q])
    np.add.at(asums, grp, aw)
    scale = max(np.abs(asums).max(), abs(ab))
    print("  fit %2d C %-6g held %2d newton %5d  obj rel %+.2e  linpred %.2e  coef sums %.2e of the largest (%.3g)  icpt %.2e" % (
        j, fp[j], ff[j], iters[q], obj / float(z["arb_obj_" + tag][j]) - 1, np.abs(lin - alin).max() / max(np.abs(alin).max(), 1e-300),
        np.abs(sums - asums).max() / scale, scale, abs(icpt[q] - ab) / scale), flush=True)
