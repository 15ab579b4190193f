"""Worker for test_device_survivor_exchange_matches_get_results (fresh process, torch first)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PSK_WITH_TORCH"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ["MASTER_PORT"] = "29641"
import torch  # noqa: E402,F401  (before libpsk.so)

from phenotypeseeker_amd import dist  # noqa: E402
from phenotypeseeker_amd.engine import PskContext  # noqa: E402

g = dist.Group()
g.world, g.rank, g.local_rank = 1, 0, 0
g.init("nccl", force=True)
with PskContext(0) as ctx:
    n, m = 200, 300_000
    ctx.synth_presence(m, n, seed=11)
    ph = (np.arange(n) % 2).astype(np.int8)
    npass = ctx.chi2_scan(ph, None, 2, n - 2, 0.05, False, m)
    assert npass > 100, npass
    ref = ctx.get_results(npass)
    ref_bits = ctx.get_rows(ref["row"])
    x = dist.SurvivorExchange(g, ctx.presence_shape()[1], cap_records=64)   # forces a regrow
    res, bits = x.gather(ctx)
    assert x.cap >= npass
    for key in ("word", "stat", "p", "n_with"):
        assert np.array_equal(res[key], ref[key]), key
    assert np.array_equal(bits, ref_bits)
    s0, _ = x.start(ctx)                      # double-buffered form: two exchanges in flight
    ctx.chi2_scan(1 - ph, None, 2, n - 2, 0.05, False, m)
    s1, _ = x.start(ctx)
    a, b = x.finish(s0), x.finish(s1)
    assert np.array_equal(a[0]["word"], ref["word"]) and np.array_equal(b[0]["word"], ref["word"])
g.close()
print("exchange ok", npass)
