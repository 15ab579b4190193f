#!/usr/bin/env python3
"""Random byte strings, damaged FASTA and damaged / truncated FASTQ through the host framing (psk_frame_sequence) with
exact-size heap buffers on both sides: meant to run against the host-ASAN build of the library (tools/asan_host.sh);
no GPU needed.  usage: tools/fuzz_framing.py SEED CASES"""
import ctypes
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenotypeseeker_amd import _lib  # noqa: E402

lib = _lib.load()
rng = random.Random(int(sys.argv[1]))
alphabet = b"ACGTacgtNnUu>@+\n\r\t \x00;IJ!#0159"
n_cases = 0
for it in range(int(sys.argv[2])):
    kind = rng.random()
    ln = rng.choice([0, 1, 2, 3, 7, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 255, 256, 257, 1000, 4097, 20000])
    if kind < 0.4:
        data = bytes(rng.choice(alphabet) for _ in range(ln))
    elif kind < 0.7:
        body = bytes(rng.choice(b"ACGT") for _ in range(ln))
        lines = [body[i:i + 70] for i in range(0, len(body), 70)]
        data = b">h\n" + b"\n".join(lines) + (b"\n" if rng.random() < 0.5 else b"")
        if rng.random() < 0.5 and len(data) > 5:
            b = bytearray(data)
            for _ in range(rng.randrange(1, 6)):
                b[rng.randrange(len(b))] = rng.choice(alphabet)
            data = bytes(b)
    else:
        recs = []
        for r in range(rng.randrange(0, 6)):
            s = bytes(rng.choice(b"ACGTN") for _ in range(rng.randrange(0, 200)))
            recs.append(b"@r%d\n" % r + s + b"\n+\n" + b"I" * len(s) + b"\n")
        data = b"".join(recs)
        if rng.random() < 0.5 and len(data) > 5:
            b = bytearray(data)
            for _ in range(rng.randrange(1, 6)):
                b[rng.randrange(len(b))] = rng.choice(alphabet)
            data = bytes(b[: rng.randrange(len(b) + 1)])
    # exact-size heap copies so that ASAN sees any over-read / over-write
    src = (ctypes.c_char * max(len(data), 1)).from_buffer_copy(data if data else b"\x00")
    out = (ctypes.c_uint8 * max(len(data), 1))()
    n = lib.psk_frame_sequence(ctypes.cast(src, ctypes.c_char_p), len(data), ctypes.addressof(out), len(data) if data else 1)
    assert n >= -1 and n <= max(len(data), 1), n
    n_cases += 1
print("fuzz ok", n_cases)
