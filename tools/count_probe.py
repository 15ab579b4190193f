#!/usr/bin/env python3
"""Times psk_count_kmers_batch on in-memory FASTA for several framing-thread counts and chunk sizes.
usage: tools/count_probe.py N LENGTH threads[,threads...] [chunk[,chunk...]]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.engine import PskContext  # noqa: E402
from phenotypeseeker_amd.synth import GenomeSet  # noqa: E402

n, length = int(sys.argv[1]), int(sys.argv[2])
threads = [int(t) for t in sys.argv[3].split(",")]
chunks = [int(t) for t in (sys.argv[4] if len(sys.argv) > 4 else "16").split(",")]
gs = GenomeSet(n, length, seed=12345)
datas = [gs.sample(i)[1] for i in range(n)]
tot = sum(len(d) for d in datas)
with PskContext(0) as ctx:
    for rep in range(2):
        for nt in threads:
            for ch in chunks:
                ctx.begin(13, n)
                t = time.time()
                for lo in range(0, n, ch):
                    ctx.count_kmers_batch(lo, datas[lo:lo + ch], nt)
                dt = time.time() - t
                print("rep %d threads %3d chunk %3d: %.3f s  %.2f GB/s" % (rep, nt, ch, dt, tot / dt / 1e9), flush=True)
