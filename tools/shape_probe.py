#!/usr/bin/env python3
"""Unweighted chi2 scan across sample counts: kernel time and achieved GB/s of algorithmic bytes (M * 8 * ceil(N/64))
at ~2 GB of matrix per shape (far beyond the 256-MiB Infinity Cache), once with the synthetic matrix's 1 % of surviving
rows and once with config 2's share (0.008 %).  Output committed as profiles/r0N_shapes.md.
usage: tools/shape_probe.py [N ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.engine import PskContext, words_per_row  # noqa: E402

shapes = [int(a) for a in sys.argv[1:]] or [30, 64, 128, 256, 512, 1024, 2048, 4096]
print("| samples | lanes per row | rows | survivors | kernel ms | algorithmic GB/s | frac of 8 TB/s | stored GB/s |")
print("|---:|---:|---:|---:|---:|---:|---:|---:|")
with PskContext(0) as ctx:
    for N in shapes:
        words = (N + 63) // 64
        wpr = words_per_row(N)
        M = int(2.0e9 // (8 * wpr))
        G = 1
        while G < wpr // 2:
            G *= 2
        if wpr == 1:
            G = 0.5    # 8-byte rows: two per 16-byte load
        for keep in (0, 80):
            ctx.begin(16, N)
            ctx.synth_presence(M, N, 7 | (keep << 48))
            wpr = ctx.presence_shape()[1]
            ph = (np.arange(N) % 2).astype(np.int8)
            n = ctx.chi2_scan(ph, None, 2, N - 2, 0.05, False, M)
            ctx.rescan_timed(50)
            ms = ctx.rescan_timed(100)
            print("| %d | %g | %d | %d (%.3f %%) | %.4f | %.0f | %.3f | %.0f |"
                  % (N, G, M, n, 100.0 * n / M, ms, M * 8 * words / ms / 1e6, M * 8 * words / ms / 1e6 / 8000, M * 8 * wpr / ms / 1e6), flush=True)
