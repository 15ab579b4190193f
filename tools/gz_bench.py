#!/usr/bin/env python3
"""Throughput of the device inflate (csrc/gz_inflate.hip) on synthetic .fastq.gz / .fasta.gz images, with zlib on one host
thread beside it.  python tools/gz_bench.py [fastq|fasta|bgzf] [files] [MB of text per file] [level] [reps] [noverify]"""
import gzip
import os
import sys
import time
import zlib
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(args):
    kind, mb, seed, level = args
    rng = np.random.default_rng(seed)
    n = mb << 20
    if kind == "fasta":
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
        s = seq.tobytes()
        text = b">contig\n" + b"\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + b"\n"
    else:
        rl = 150
        genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 5_000_000)]
        n_reads = n // (2 * rl + 20)
        at = rng.integers(0, len(genome) - rl, n_reads)
        # qualities as sequencers write them: a few values, long runs
        q = np.repeat((rng.choice([2, 14, 21, 27, 32, 36, 37], n_reads * 10, p=[.02, .03, .05, .1, .2, .3, .3]) + 33).astype(np.uint8), 15)
        parts = []
        for i in range(n_reads):
            parts.append(b"@SIM:1:FCX:1:%d:%d:%d 1:N:0:ATCACG\n" % (i % 16, i * 7 % 20000, i * 13 % 20000))
            parts.append(genome[at[i]:at[i] + rl].tobytes())
            parts.append(b"\n+\n")
            parts.append(q[i * rl:(i + 1) * rl].tobytes())
            parts.append(b"\n")
        text = b"".join(parts)
    if kind == "bgzf":       # as bgzip writes it: members of at most 64 KB, each with its length in a 'BC' extra field
        import struct
        out = []
        for i in list(range(0, len(text), 0xff00)) + [len(text)]:
            piece = text[i:i + 0xff00] if i < len(text) else b""
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            raw = co.compress(piece) + co.flush()
            out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(raw) + 8 - 1))
            out.append(raw + struct.pack("<II", zlib.crc32(piece), len(piece) & 0xffffffff))
        return b"".join(out), len(text), zlib.crc32(text)
    return gzip.compress(text, level), len(text), zlib.crc32(text)


def make_fastq_like(args):
    """One config-5-sized sample written twice: <dir>/s<i>.fastq and .fastq.gz (level 6).  Returns (plain path, gz path,
    text bytes, gz bytes).  Reads of 150 bp drawn from a 5-Mbp genome with 0.5 % errors; names as an Illumina run writes
    them; qualities from the eight bins of a NovaSeq, in runs.  args: (directory, i, reads[, GenomeSet parameters (n, length,
    seed): the genome is sample i of that set -- related genomes, a phenotype -- instead of a random one; only the .gz is kept])."""
    d, i, n_reads = args[:3]
    rng = np.random.default_rng(1000 + i)
    rl = 150
    if len(args) > 3:
        sys.path.insert(0, ROOT)
        from phenotypeseeker_amd.synth import GenomeSet
        genome = np.frombuffer(b"ACGT", dtype=np.uint8)[GenomeSet(*args[3]).codes(i)]
    else:
        genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 5_000_000)]
    parts = []
    for r0 in range(0, n_reads, 100_000):     # (in blocks: the index arrays of two million reads at once are gigabytes)
        nb = min(100_000, n_reads - r0)
        at = rng.integers(0, len(genome) - rl, nb)
        reads = genome[at[:, None] + np.arange(rl)[None, :]]
        wrong = rng.integers(0, reads.size, int(reads.size * 0.005))
        reads.reshape(-1)[wrong] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, len(wrong))]
        runs = rng.choice(np.frombuffer(b"#+5?FFFF", dtype=np.uint8), (nb, 10), p=[.02, .03, .05, .1, .2, .2, .2, .2])
        qual = np.repeat(runs, 15, axis=1)
        for j in range(nb):
            r = r0 + j
            parts.append(b"@A00123:45:HXXXXXXXX:1:%d:%d:%d 1:N:0:ATCACGTT+AGGCTATA\n" % (1101 + r % 78, 1000 + r * 7 % 30000, 1000 + r * 13 % 35000))
            parts.append(reads[j].tobytes())
            parts.append(b"\n+\n")
            parts.append(qual[j].tobytes())
            parts.append(b"\n")
    text = b"".join(parts)
    plain, packed = os.path.join(d, "s%d.fastq" % i), os.path.join(d, "s%d.fastq.gz" % i)
    if len(args) <= 3:
        with open(plain, "wb") as f:
            f.write(text)
    gz = gzip.compress(text, 6)
    with open(packed, "wb") as f:
        f.write(gz)
    return plain, packed, len(text), len(gz)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "fastq"
    files = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    mb = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    level = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    t0 = time.time()
    with ProcessPoolExecutor(min(files, os.cpu_count() or 4)) as ex:
        made = list(ex.map(make, [(kind, mb, 100 + i, level) for i in range(files)]))
    images = [m[0] for m in made]
    text_bytes = sum(m[1] for m in made)
    comp_bytes = sum(len(b) for b in images)
    print("%d %s files: %.1f MB of text, %.1f MB compressed (level %d), made in %.1f s" % (files, kind, text_bytes / 1e6, comp_bytes / 1e6, level, time.time() - t0))
    t0 = time.time()
    one = zlib.decompress(images[0], 31)
    t_host = time.time() - t0
    print("zlib, one host thread: %.3f s for file 0 = %.3f GB/s of text" % (t_host, len(one) / t_host / 1e9))
    import json
    from phenotypeseeker_amd.engine import PskContext
    best = None
    with PskContext(0) as ctx:
        for r in range(reps):
            t0 = time.time()
            _, lens, routes, ms = ctx.gz_inflate(images, want_text=False)
            wall = time.time() - t0
            assert lens == [m[1] for m in made], "lengths differ"
            ms = ms or wall * 1e3       # (zlib on the library's host threads -- PSK_GZ_DEVICE_MIN_MB -- reports no device time)
            print("device inflate: %.1f ms (call %.1f ms) = %.2f GB/s of text; routes %s" % (ms, wall * 1e3, text_bytes / ms / 1e6, sorted(set(routes))))
            best = ms if best is None or ms < best else best
        if "noverify" not in sys.argv:      # (under the profiler: every launch of the same size, so that the means mean something;
            texts, _, _, _ = ctx.gz_inflate(images[:2])     # the lengths are compared above and the device checks the CRC-32 itself)
            assert [zlib.crc32(t) for t in texts] == [m[2] for m in made[:2]], "text differs"
    print("ok")
    # (for tools/summarise_profiles.py: what the streaming kernels of the inflate move per launch at the least -- the 16-bit
    # symbols read and the text written by gz_resolve_kernel, the text read by gz_crc_kernel; the decoders are bound by the
    # latency of one lane's serial work, not by bytes: no figure for them)
    print(json.dumps({"workload": "gzinflate", "algorithmic_bytes_per_launch": {"gz_resolve_kernel": 3 * text_bytes, "gz_crc_kernel": text_bytes},
                      "notes": {"files": files, "kind": kind, "gzip_level": level, "text_bytes": text_bytes, "gz_bytes": comp_bytes,
                                "device_inflate_ms_best": round(best, 1), "GBps_of_text": round(text_bytes / best / 1e6, 1),
                                "GBps_of_compressed_input": round(comp_bytes / best / 1e6, 1),
                                "zlib_one_host_thread_GBps_of_text": round(len(one) / t_host / 1e9, 3)}}))


if __name__ == "__main__":
    main()
