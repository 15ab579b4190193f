#!/usr/bin/env python3
"""Unweighted chi2 scan across sample counts: kernel time and achieved GB/s of algorithmic bytes (M * 8 * ceil(N/64))
at ~0.75 GB of matrix per shape.  usage: tools/shape_probe.py [N ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.engine import PskContext  # noqa: E402

shapes = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024, 2048, 4096]
with PskContext(0) as ctx:
    for N in shapes:
        words = (N + 63) // 64
        _, wpr = 0, words + (words & 1) if words > 1 else 2
        M = int(0.75e9 // (8 * max(wpr, 2)))
        ctx.begin(16, N)
        ctx.synth_presence(M, N, 7)
        wpr = ctx.presence_shape()[1]
        ph = (np.arange(N) % 2).astype(np.int8)
        n = ctx.chi2_scan(ph, None, 2, N - 2, 0.05, False, M)
        ctx.rescan_timed(200)
        ms = ctx.rescan_timed(200)
        print("N %5d  wpr %3d  rows %9d  survivors %7d  kernel %.4f ms  algorithmic %.0f GB/s  stored %.0f GB/s"
              % (N, wpr, M, n, ms, M * 8 * words / ms / 1e6, M * 8 * wpr / ms / 1e6), flush=True)
