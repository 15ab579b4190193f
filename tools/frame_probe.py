#!/usr/bin/env python3
"""The device framing kernels alone (no counting beside them): 20 x psk_frame_sequence_gpu on one 5-Mbp FASTA sample and
on one 0.3-GB FASTQ sample.  Under tools/prof.sh this gives their isolated durations.  usage: tools/frame_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from phenotypeseeker_amd.engine import PskContext  # noqa: E402
from phenotypeseeker_amd.synth import GenomeSet  # noqa: E402
from test_gpu_configs import _fastq_sample  # noqa: E402

gs = GenomeSet(1, 5_000_000, seed=1)
fa = gs.sample(0)[1]
fq = _fastq_sample(gs.codes(0), 1_000_000, 150, seed=[5, 0])
with PskContext(0) as ctx:
    for name, data in (("fasta", fa), ("fastq", fq)):
        t0 = time.time()
        for _ in range(20):
            out = ctx.frame_sequence_gpu(data)
        print("%s: %d -> %d bytes, %.2f ms per call incl. upload and download" % (name, len(data), len(out), (time.time() - t0) / 20 * 1e3))
