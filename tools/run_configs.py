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


def run_cfg3_slab(n=2048, length=5_000_000, k=16, world=8, rank=0, chunk=64):
    """One rank's share of BASELINE config 3 at FULL size: all 2048 samples are tokenised, only the words of
    slab `rank` of `world` are kept (dist.slab_bounds), then presence + chi2 on the slab's rows."""
    from phenotypeseeker_amd.dist import slab_bounds
    gs = GenomeSet(n, length, seed=99)
    lo, hi = slab_bounds(k, world, rank)
    out = {"config": "cfg3 slab %d/%d: %d x %.0f Mbp, k=%d" % (rank, world, n, length / 1e6, k)}
    with PskContext(0) as ctx:
        ctx.begin(k, n, lo, hi)
        tg = tc = 0.0
        for c0 in range(0, n, chunk):
            a = time.time()
            datas = [gs.sample(i)[1] for i in range(c0, min(n, c0 + chunk))]
            b = time.time()
            ctx.count_kmers_batch(c0, datas, 8)
            tg += b - a
            tc += time.time() - b
        out["generate_s"], out["count_s"] = round(tg, 2), round(tc, 2)
        t1 = time.time()
        M = ctx.build_presence()
        out["presence_s"] = round(time.time() - t1, 3)
        out["rows"] = M
        out["matrix_GB"] = round(M * ((n + 63) // 64) * 8 / 1e9, 2)
        ph = np.array([gs.phenotype(i) for i in range(n)], dtype=np.int8)
        npass = ctx.chi2_scan(ph, None, 2, n - 2, 0.05, False, M * world)
        out["first_launch_ms"] = round(ctx.last_scan_ms(), 3)
        ctx.rescan_timed(10)
        out["scan_kernel_ms"] = round(ctx.last_scan_ms(), 3)
        out["survivors"] = npass
        out["cells_per_s"] = M * n / (ctx.last_scan_ms() * 1e-3)
        out["bits_GBps"] = round(M * ((n + 63) // 64) * 8 / (ctx.last_scan_ms() * 1e-3) / 1e9, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg3", "cfg4", "cfg5"]
    if "cfg3-slab" in which:
        run_cfg3_slab()
    if "cfg3" in which:
        run("cfg3-scaled: 2048 x 0.5 Mbp, k=16", 2048, 500_000, 16)
    if "cfg4" in which:
        run("cfg4-scaled: 1024 x 1 Mbp continuous, Welch t", 1024, 1_000_000, 13, continuous=True)
    if "cfg5" in which:
        run("cfg5-scaled: 64 samples x 100k reads x 150 bp FASTQ", 64, 1_000_000, 13, fastq=True, reads=100_000)
