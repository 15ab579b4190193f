#!/usr/bin/env python3
"""Per-step cost of the multi-GPU step (scan + export + all-gather) on ONE GPU with a one-rank nccl group:
what the exchange adds to a scan.  usage: tools/exchange_probe.py [ROWS] [SAMPLES]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PSK_WITH_TORCH"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ["MASTER_PORT"] = "29643"
import torch  # noqa: E402,F401  (before libpsk.so)

from phenotypeseeker_amd import dist  # noqa: E402
from phenotypeseeker_amd.engine import PskContext  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 22_950_458
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = dist.Group()
g.world, g.rank, g.local_rank = 1, 0, 0
g.init("nccl", force=True)
with PskContext(0) as ctx:
    ctx.begin(13, n)
    ctx.synth_presence(m, n, seed=11)
    ph = (np.arange(n) % 2).astype(np.int8)
    x = dist.SurvivorExchange(g, ctx.presence_shape()[1], cap_records=1 << 20)
    args = (ph, None, 2, n - 2, 0.05, False, m)
    for mode in ("scan only", "scan + export + all-gather", "the same, next scan launched before the all-gather is queued",
                 "the same, two scans in flight"):
        pending = []
        for rep in range(2):
            t = time.perf_counter()
            if mode.endswith("two scans in flight"):
                ctx.chi2_scan_begin(*args)
                ctx.chi2_scan_begin(*args)
                for i in range(100):
                    ctx.scan_end()
                    s = x.export(ctx)
                    if i + 2 < 100:
                        ctx.chi2_scan_begin(*args)
                    x.collect(s)
                    pending.append(s)
                    if len(pending) > 1:
                        x.wait(pending.pop(0))
            elif mode.startswith("the same"):
                ctx.chi2_scan_begin(*args)
                for i in range(100):
                    ctx.scan_end()
                    s = x.export(ctx)
                    if i + 1 < 100:
                        ctx.chi2_scan_begin(*args)
                    x.collect(s)
                    pending.append(s)
                    if len(pending) > 1:
                        x.wait(pending.pop(0))
            else:
                for _ in range(100):
                    ctx.chi2_scan(*args)
                    if mode != "scan only":
                        s, _ = x.start(ctx)
                        pending.append(s)
                        if len(pending) > 1:
                            x.wait(pending.pop(0))
            while pending:
                x.wait(pending.pop(0))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 100
        print("%-62s %.3f ms per step (scan kernel %.3f ms)" % (mode, dt * 1e3, ctx.last_scan_ms()))
g.close()
