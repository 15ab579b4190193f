#!/usr/bin/env python3
"""BASELINE config 5 at full per-sample size: ONE sample of READS x 150-bp FASTQ reads through
psk_count_kmers; checks the size-independent properties (window count, sum of counts, sortedness) and prints
timings (the oracle comparison of a prefix lives in tests/test_gpu_parity.py).  usage: tools/fastq_probe.py READS [k]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.engine import PskContext  # noqa: E402
from phenotypeseeker_amd.synth import GenomeSet, fastq_reads  # noqa: E402

reads = int(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 13
gs = GenomeSet(2, 5_000_000, seed=99)
t = time.time()
data = fastq_reads(gs.codes(0), reads, 150, seed=[5, 0])
print("generated %.1f MB of FASTQ in %.1f s" % (len(data) / 1e6, time.time() - t), flush=True)
with PskContext(0) as ctx:
    ctx.begin(k, 1)
    for rep in range(2):
        t = time.time()
        nu, nt = ctx.count_kmers(0, data)
        dt = time.time() - t
        print("count: %.3f s  (%.2f GB/s of file bytes)  n_unique %d n_total %d" % (dt, len(data) / dt / 1e9, nu, nt), flush=True)
    assert nt == reads * (150 - k + 1), (nt, reads * (150 - k + 1))
    os.environ["PSK_HOST_FRAMING"] = "1"
    t = time.time()
    nu2, nt2 = ctx.count_kmers(0, data)
    dt = time.time() - t
    del os.environ["PSK_HOST_FRAMING"]
    print("host framing: %.3f s  (%.2f GB/s of file bytes)" % (dt, len(data) / dt / 1e9), flush=True)
    assert (nu2, nt2) == (nu, nt)
    path = "/tmp/psk_fastq_probe.fastq"
    with open(path, "wb") as f:
        f.write(data)
    for rep in range(2):
        t = time.time()
        ctx.count_kmers_files(0, [path], 1)
        dt = time.time() - t
        print("from a file: %.3f s  (%.2f GB/s of file bytes)" % (dt, len(data) / dt / 1e9), flush=True)
    os.remove(path)
    words, freqs = ctx.get_list(0, nu)
    assert int(freqs.astype(np.uint64).sum()) == nt
    assert np.all(words[1:] > words[:-1])
    print("properties ok: windows, sum of counts, strictly ascending words; max count %d" % freqs.max())
