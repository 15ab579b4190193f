#!/usr/bin/env python3
"""Runs scaled-down versions of BASELINE.json configs 3-5 on one GPU and prints stage timings
(parity of these shapes is covered by tests/; this script is for spotting performance cliffs)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from phenotypeseeker_amd.engine import PskContext  # noqa: E402
from phenotypeseeker_amd.synth import GenomeSet, fastq_reads  # noqa: E402


def run(tag, n, length, k, continuous=False, fastq=False, reads=0):
    gs = GenomeSet(n, length, seed=99)
    out = {"config": tag, "n_samples": n, "length": length, "k": k}
    with PskContext(0) as ctx:
        ctx.begin(k, n)
        t0 = time.time()
        tg = tc = 0.0
        for i in range(n):
            a = time.time()
            if fastq:
                data = fastq_reads(gs.codes(i), reads, 150, seed=[5, i])
            else:
                _, data = gs.sample(i)
            b = time.time()
            ctx.count_kmers(i, data)
            tg += b - a
            tc += time.time() - b
        out["generate_s"], out["count_s"] = round(tg, 2), round(tc, 2)
        t1 = time.time()
        M = ctx.build_presence()
        out["presence_s"] = round(time.time() - t1, 3)
        out["rows"] = M
        if continuous:
            vals = np.array([gs.continuous_phenotype(i) for i in range(n)])
            t2 = time.time()
            npass = ctx.ttest_scan(vals, np.ones(n, np.uint8), None, 2, n - 2, 0.05, M)
            out["scan_s"] = round(time.time() - t2, 4)
        else:
            ph = np.array([gs.phenotype(i) for i in range(n)], dtype=np.int8)
            t2 = time.time()
            npass = ctx.chi2_scan(ph, None, 2, n - 2, 0.05, False, M)
            out["scan_s"] = round(time.time() - t2, 4)
            out["first_launch_ms"] = round(ctx.last_scan_ms(), 3)
            ctx.rescan_timed(10)  # warm launches: the first one pays code-object load
        out["scan_kernel_ms"] = round(ctx.last_scan_ms(), 3)
        out["survivors"] = npass
        out["cells_per_s"] = M * n / (ctx.last_scan_ms() * 1e-3)
        out["bits_GBps"] = M * ((n + 63) // 64) * 8 / (ctx.last_scan_ms() * 1e-3) / 1e9
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg3", "cfg4", "cfg5"]
    if "cfg3" in which:
        run("cfg3-scaled: 2048 x 0.5 Mbp, k=16", 2048, 500_000, 16)
    if "cfg4" in which:
        run("cfg4-scaled: 1024 x 1 Mbp continuous, Welch t", 1024, 1_000_000, 13, continuous=True)
    if "cfg5" in which:
        run("cfg5-scaled: 64 samples x 100k reads x 150 bp FASTQ", 64, 1_000_000, 13, fastq=True, reads=100_000)
