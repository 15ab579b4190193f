#!/usr/bin/env python3
"""Kernel times of the f64 scans (Welch t, GSC-weighted chi2) on a synthetic presence matrix.
usage: tools/scan_probe.py ROWS SAMPLES [min_samples]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.engine import PskContext  # noqa: E402

M, N = int(sys.argv[1]), int(sys.argv[2])
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rng = np.random.default_rng(3)
with PskContext(0) as ctx:
    ctx.begin(16, N)
    ctx.synth_presence(M, N, 7)
    ph = (np.arange(N) % 2).astype(np.int8)
    vals = rng.normal(0, 1, N) + ph * 0.3
    w = rng.uniform(0.5, 1.5, N)
    for rep in range(3):
        out = {}
        n = ctx.chi2_scan(ph, None, mn, N - 2, 0.05, False, M); out["chi2"] = (ctx.last_scan_ms(), n)
        n = ctx.chi2_scan(ph, w, mn, N - 2, 0.05, False, M); out["chi2_weighted"] = (ctx.last_scan_ms(), n)
        n = ctx.ttest_scan(vals, np.ones(N, np.uint8), None, mn, N - 2, 0.05, M); out["ttest"] = (ctx.last_scan_ms(), n)
        n = ctx.ttest_scan(vals, np.ones(N, np.uint8), w, mn, N - 2, 0.05, M); out["ttest_weighted"] = (ctx.last_scan_ms(), n)
        print(" ".join("%s %.3f ms (%d)" % (k, v[0], v[1]) for k, v in out.items()), flush=True)
