#!/usr/bin/env python3
"""Presence build of one rank's share of config 3 (2,048 x 5 Mbp, k = 16, balanced slab 0 of 8), repeated: the phases of
the streaming merge (PSK_TRACE marks on stderr) with the matrix buffers already allocated, beside the sort route.
usage: tools/presence_probe.py [n_samples] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd import dist  # noqa: E402
from phenotypeseeker_amd.engine import PskContext  # noqa: E402
from phenotypeseeker_amd.synth import GenomeSet  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L, k, world = 5_000_000, 16, 8
gs = GenomeSet(n, L, seed=12345)
with PskContext(0) as ctx:
    ctx.begin(k, 1)
    nu0, _ = ctx.count_kmers(0, gs.sample(0)[1])
    bounds = dist.quantile_bounds(dist.pilot_points(ctx.get_list(0, nu0)[0]), k, world)
    ctx.begin(k, n, bounds[0], bounds[1])
    pairs = 0
    for s0 in range(0, n, 64):
        nu, _ = ctx.count_kmers_batch(s0, [gs.sample(i)[1] for i in range(s0, min(s0 + 64, n))], 8)
        pairs += sum(nu)
    for route in (["merge"] * reps + ["sort"] * (1 if os.environ.get("PROBE_SORT") else 0)):
        if route == "sort":
            os.environ["PSK_NO_MERGE_PRESENCE"] = "1"
        t0 = time.time()
        m = ctx.build_presence()
        print("%s route: %d rows from %d pairs in %.1f ms" % (route, m, pairs, (time.time() - t0) * 1e3), flush=True)
