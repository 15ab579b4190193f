#!/usr/bin/env python3
"""Where does the ingest time of a FRESH process on a FRESH box go?  Times library load, psk_init, and every
psk_count_kmers_batch call of the cfg-2 ingest (256 x 5 Mbp, k = 13, four calls of 64 samples), then the
presence build, then the same ingest again in the same process (warm), with the samples generated up front so
that host generation is not in any of the figures.
usage: tools/cold_probe.py [n_samples] [length] [k]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
length = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 13

t0 = time.perf_counter()
from phenotypeseeker_amd import _lib  # noqa: E402
_lib.load()
t1 = time.perf_counter()
from phenotypeseeker_amd.engine import PskContext  # noqa: E402
from phenotypeseeker_amd.synth import GenomeSet  # noqa: E402
ctx = PskContext(0)
t2 = time.perf_counter()
print("load libpsk %.3f s, psk_init %.3f s" % (t1 - t0, t2 - t1), flush=True)
gs = GenomeSet(n, length, seed=12345)
fas = [gs.sample(i)[1] for i in range(n)]
print("generated %d samples" % n, flush=True)
for rnd in range(3):
    ta = time.perf_counter()
    ctx.begin(k, n)
    tb = time.perf_counter()
    calls = []
    for lo in range(0, n, 64):
        t = time.perf_counter()
        ctx.count_kmers_batch(lo, fas[lo:lo + 64], 8)
        calls.append(time.perf_counter() - t)
    tc = time.perf_counter()
    M = ctx.build_presence() if not os.environ.get("PSK_PROBE_NOCHECK") else 1
    td = time.perf_counter()
    pheno = np.array([1 if i % 2 == 0 else 0 for i in range(n)], dtype=np.int8)
    if not os.environ.get("PSK_PROBE_NOCHECK"):
        ctx.chi2_scan(pheno, None, 2, n - 2, 0.05, False, M)
    te = time.perf_counter()
    print("round %d: begin %.3f  count %.3f s (calls: %s)  presence %.3f  first scan %.4f  M=%d"
          % (rnd, tb - ta, tc - tb, " ".join("%.3f" % c for c in calls), td - tc, te - td, M), flush=True)
ctx.close()
