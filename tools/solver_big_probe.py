#!/usr/bin/env python3
"""The L1 grid search of a 2048-genome run whose 1000 selected k-mers have 907 distinct presence patterns (the model
stage was 16.6 s of that run's 17.5 s): whole-call time, time per grid value (its 11 fits as one call), Newton counts
and the objective of every fit, so that solver variants can be compared.  usage: tools/solver_big_probe.py [per-C]
(A library built with `make EXTRA=-DPSK_SV_STATS` prints per fit: Newton steps, inner sweeps, coordinate visits, and the
clock cycles inside the descent and in total -- r02: ~2,800 cycles per coordinate visit on one wave (~450 instructions at a lone
wave's issue rate), ~2,000 on four; the grid's wall-clock is its slowest fit: 4,586 sweeps over ~900 coordinates.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.engine import PskContext  # noqa: E402

d = np.load(os.path.join(ROOT, "tools", "data", "fit2048_907.npz"))
X = np.unpackbits(d["Xbits"], axis=1)[:, : int(d["p"])].astype(np.float32)
y, fold, fp, ff = d["y"], d["fold"], d["fit_param"], d["fit_fold"]
tol, mi = float(d["tol"]), int(d["max_iter"])
ypm = 2.0 * y - 1.0
with PskContext(0) as ctx:
    ctx.logreg_l1_fit(X, y, fold, fp[-1:], ff[-1:], tol, mi)
    t = time.time()
    coef, icpt, it = ctx.logreg_l1_fit(X, y, fold, fp, ff, tol, mi)
    whole = time.time() - t
    objs = []
    for i in range(len(fp)):
        tr = fold != ff[i]
        z = X[tr].astype(np.float64) @ coef[i] + icpt[i]
        objs.append(np.abs(coef[i]).sum() + abs(icpt[i]) + fp[i] * np.logaddexp(0, -ypm[tr] * z).sum())
    print("whole grid %.3f s (r02 start: %.1f s)  objective sum %.8f  newton total %d max %d" % (whole, float(d["secs"]), sum(objs), it.sum(), it.max()))
    if len(sys.argv) > 1:
        for c in sorted(set(fp.tolist()), reverse=True):
            sel = np.nonzero(fp == c)[0]
            t = time.time()
            co, ic, itc = ctx.logreg_l1_fit(X, y, fold, fp[sel], ff[sel], tol, mi)
            print("C=%-10g %2d fits %.3f s  newton %s  nnz %s" % (c, len(sel), time.time() - t, itc.tolist(), [(int((r != 0).sum())) for r in co][:4]))
