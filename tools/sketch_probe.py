#!/usr/bin/env python3
"""Batch counting with and without the MinHash sketches riding on it.  usage: tools/sketch_probe.py N LENGTH"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.engine import PskContext  # noqa: E402
from phenotypeseeker_amd.synth import GenomeSet  # noqa: E402

n, length = int(sys.argv[1]), int(sys.argv[2])
gs = GenomeSet(n, length, seed=12345)
datas = [gs.sample(i)[1] for i in range(n)]
with PskContext(0) as ctx:
    for rep in range(3):
        for sk in (None, (21, 1000, 42)):
            ctx.begin(13, n)
            t = time.time()
            for lo in range(0, n, 64):
                ctx.count_kmers_batch(lo, datas[lo:lo + 64], 8, sketch=sk)
            dt = time.time() - t
            print("rep %d  sketch %-5s  %.3f s  %.0f us per sample" % (rep, "yes" if sk else "no", dt, dt / n * 1e6), flush=True)
