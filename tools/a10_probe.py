"""GPU L1-logistic solver against the active-set Newton arbiter (oracle_model.logreg_l1_arbiter) on the model_kat designs:
which tolerance the HIP solver needs for 1e-6 on coefficients / pattern sums / Xw+b, and how the liblinear fixture does."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import oracle_model as OM  # noqa: E402
from phenotypeseeker_amd.engine import PskContext  # noqa: E402

z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model_kat.npz"))
with PskContext(0) as ctx:
    for tag in "12":
        X, y = z["X" + tag], z["y" + tag]
        Cs = [float(c) for c in z["Cs"]]
        arb = [OM.logreg_l1_arbiter(X, y, C, z["logreg_coef" + tag][ci], float(z["logreg_icpt" + tag][ci])) for ci, C in enumerate(Cs)]
        for tol in (1e-8, 1e-10, 1e-12, 1e-14):
            t0 = time.time()
            coef, icpt, iters = ctx.logreg_l1_fit(X, y, np.zeros(len(y), np.int32), Cs, [-1] * len(Cs), tol=tol, max_iter=5000)
            dt = time.time() - t0
            for ci, C in enumerate(Cs):
                a = arb[ci]
                wg = np.zeros(len(a["w_groups"]))
                np.add.at(wg, a["group"], coef[ci])
                scale = max(np.abs(a["w_groups"]).max(), 1e-300)
                lp = X @ coef[ci] + icpt[ci]
                print("design %s tol %.0e C %-8.3g iters %4d  max|dw|/max|w| %.2e  max rel(w) %.2e  |db| %.2e  lp abs %.2e  obj-arb %.2e  (%.3fs)" % (
                    tag, tol, C, iters[ci], np.abs(wg - a["w_groups"]).max() / scale,
                    np.max(np.abs(wg - a["w_groups"]) / np.maximum(np.abs(a["w_groups"]), 1e-300) * (a["w_groups"] != 0)) if (a["w_groups"] != 0).any() else 0.0,
                    abs(icpt[ci] - a["b"]), np.abs(lp - a["linpred"]).max(),
                    OM.logreg_l1_objective(X, y, coef[ci], icpt[ci], C) - a["objective"], dt), flush=True)
