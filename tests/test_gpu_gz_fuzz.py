"""The device inflate as a parser of UNTRUSTED bytes: >= 20,000 mutated gzip members per run through
psk_gz_inflate with 64-KB guard bands around the text / symbol / match buffers (PSK_GZ_GUARD=1; GPU AddressSanitizer is not
available on this pool, canaries are).  Every case: zlib's text, or a refusal where zlib refuses -- never another text, never a
damaged band, never a call that takes long; and the context inflates a good file afterwards.

Reference behaviour: glistmaker reads .gz through zlib and fails cleanly on a file zlib refuses; the
checker here is zlib itself (Python's zlib module), applied member by member exactly as csrc/gz_inflate.hip's host route does."""
import gzip
import io
import struct
import time
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _device_route_with_guard_bands(monkeypatch):
    monkeypatch.setenv("PSK_GZ_DEVICE_MIN_MB", "0")     # small inputs would go to zlib on the host otherwise
    monkeypatch.setenv("PSK_GZ_GUARD", "1")


def zlib_all(b):
    """Every member of the file, zero padding between / behind them skipped -- gz_host_inflate's loop; None: zlib refuses."""
    out = []
    try:
        while True:
            d = zlib.decompressobj(31)
            out.append(d.decompress(b))
            if not d.eof:
                return None
            b = d.unused_data.lstrip(b"\0")
            if not b:
                return b"".join(out)
    except zlib.error:
        return None


def _text(rng, kind, size):
    if kind == "fasta":
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size)]
        for _ in range(max(1, size // 3000)):
            a, b = (int(rng.integers(0, max(1, size - 400))) for _ in range(2))
            ln = int(rng.integers(20, 300))
            seq[b:b + ln] = seq[a:a + ln].copy()[:len(seq[b:b + ln])]
        s = seq.tobytes()
        return b">c some description\n" + b"\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + b"\n"
    if kind == "fastq":
        out = io.BytesIO()
        g = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 5000)]
        i = 0
        while out.tell() < size:
            at = int(rng.integers(0, len(g) - 100))
            out.write(b"@read_%d/1\n" % i + g[at:at + 100].tobytes() + b"\n+\n" + (rng.integers(0, 41, 100) + 33).astype(np.uint8).tobytes() + b"\n")
            i += 1
        return out.getvalue()
    if kind == "runs":          # long runs: distance-1 matches of length 258, the end of the length / distance alphabets
        return b"".join(bytes([int(rng.integers(65, 91))]) * int(rng.integers(1, 2000)) for _ in range(max(2, size // 1000)))
    return rng.integers(0, 256, size).astype(np.uint8).tobytes()    # noise: stored blocks at any level


def _member(text, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, name=None, zdict=None):
    """One gzip member around a raw DEFLATE stream of `text`.  zdict: the stream may reach back into a dictionary that is NOT
    there when it is inflated -- distances before the member's start ("invalid distance too far back")."""
    kw = {"zdict": zdict} if zdict else {}
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy, **kw)
    raw = co.compress(text) + co.flush()
    head = b"\x1f\x8b\x08" + (b"\x08" if name else b"\x00") + b"\0\0\0\0\x00\x03" + ((name + b"\0") if name else b"")
    return head + raw + struct.pack("<II", zlib.crc32(text), len(text) & 0xffffffff)


def _bgzf(text, block=0x4000):
    out = io.BytesIO()
    for piece in [text[i:i + block] for i in range(0, len(text), block)] + [b""]:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = co.compress(piece) + co.flush()
        bsize = 12 + 6 + len(raw) + 8
        out.write(b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1))
        out.write(raw + struct.pack("<II", zlib.crc32(piece), len(piece) & 0xffffffff))
    return out.getvalue()


def _corpus(rng):
    base = []
    for kind, size in (("fasta", 6000), ("fasta", 30000), ("fastq", 8000), ("fastq", 40000), ("runs", 20000), ("noise", 5000)):
        t = _text(rng, kind, size)
        for level in (1, 6, 9):
            base.append(_member(t, level))
        base.append(_member(t, 6, zlib.Z_FIXED))                       # fixed-Huffman blocks only: no header to find
        base.append(_member(t, 0))                                     # stored blocks
        base.append(_member(t, 6, name=b"reads_of_sample_1.fastq"))
        base.append(_member(t[:len(t) // 2], 6) + _member(t[len(t) // 2:], 9))        # cat a.gz b.gz
        base.append(_member(t[:len(t) // 3], 6) + b"\0" * 7 + _member(t[len(t) // 3:], 1))   # ... with padding between
        base.append(_bgzf(t))
        co = zlib.compressobj(6, zlib.DEFLATED, 31)                    # full flushes: byte-aligned block starts, empty stored blocks
        base.append(b"".join(co.compress(t[i:i + 4000]) + co.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(t), 4000)) + co.flush())
    return base


def _mutants(rng, base, want):
    """`want` mutated files, the kinds of VERDICT r05 next #3 in turn."""
    out = []
    kinds = ("flips", "truncate", "header17", "lengths", "too_far_back", "trailer", "bsize", "stored", "gzip_header", "splice", "burst", "valid_variant")
    dict_text = _text(rng, "fasta", 4000)
    strategies = (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED)
    while len(out) < want:
        kind = kinds[len(out) % len(kinds)]
        m = bytearray(base[int(rng.integers(0, len(base)))])
        n = len(m)
        if kind == "flips":                 # random multi-bit flips anywhere
            for _ in range(int(rng.integers(1, 9))):
                m[int(rng.integers(0, n))] ^= 1 << int(rng.integers(0, 8))
        elif kind == "truncate":            # at every header boundary, in the first block header, in the trailer, anywhere
            cut = int(rng.choice([int(rng.integers(0, 40)), n - int(rng.integers(1, 10)), int(rng.integers(0, n))]))
            m = m[:max(0, min(cut, n))]
        elif kind == "header17":            # forged BFINAL / BTYPE / HLIT / HDIST / HCLEN of the first block
            bits = int(rng.integers(0, 1 << 17))
            if rng.random() < 0.3:
                bits |= 0x1f << 3           # HLIT = 31 -> 288 codes: more than the alphabet has
            if rng.random() < 0.3:
                bits |= 0x1f << 8           # HDIST = 31 -> 32 codes
            at = 10
            for k in range(3):
                m[at + k] = (m[at + k] & ~((0x1ffff >> (8 * k)) & 0xff)) | ((bits >> (8 * k)) & 0xff & (0x1ffff >> (8 * k)))
        elif kind == "lengths":             # the code-length code and the run-length coded lengths behind it
            at = 12 + int(rng.integers(0, 60))
            for k in range(int(rng.integers(1, 12))):
                if at + k < n:
                    m[at + k] = int(rng.integers(0, 256))
        elif kind == "too_far_back":        # a match that reaches before the member's start; alone, or as a later member
            t = dict_text[int(rng.integers(0, 2000)):][:int(rng.integers(300, 1500))] + _text(rng, "fasta", 3000)
            bad = _member(t, 6, zdict=dict_text)
            m = bytearray(bad if rng.random() < 0.5 else bytes(m) + bad)
        elif kind == "trailer":             # lying CRC-32 / ISIZE
            at = n - 8 + int(rng.integers(0, 8))
            m[at] ^= 1 << int(rng.integers(0, 8))
        elif kind == "bsize":               # BGZF: BSIZE fields that lie (only files that have them; others get a forged field)
            at = bytes(m).find(b"BC\x02\x00")
            if at >= 0 and rng.random() < 0.8:
                nth = int(rng.integers(0, 4))
                for _ in range(nth):        # the n-th member's field
                    nxt = bytes(m).find(b"BC\x02\x00", at + 1)
                    at = nxt if nxt >= 0 else at
                v = struct.unpack_from("<H", m, at + 4)[0]
                struct.pack_into("<H", m, at + 4, int(np.clip(v + int(rng.integers(-300, 300)), 0, 65535)) if rng.random() < 0.7 else int(rng.integers(0, 65536)))
            else:                           # a plain member dressed up as BGZF with a wrong length
                m = bytearray(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, int(rng.integers(0, 65536))) + bytes(m[10:]))
        elif kind == "stored":              # LEN / NLEN of stored blocks, the bytes around them
            t = _text(rng, "noise", int(rng.integers(100, 70000)))
            m = bytearray(_member(t, 0))
            at = 10 + int(rng.integers(0, min(len(m) - 10, 12)))
            m[at] ^= 1 << int(rng.integers(0, 8))
        elif kind == "gzip_header":         # FLG values (FHCRC, FEXTRA, reserved bits), CM, the bytes the flags announce
            m[3] = int(rng.integers(0, 256)) if rng.random() < 0.5 else int(rng.choice([2, 4, 6, 8, 16, 18, 32, 64, 128]))
            if rng.random() < 0.3:
                m[2] = int(rng.integers(0, 256))
        elif kind == "splice":              # the head of one file on the tail of another; trailing garbage; a second header
            o = base[int(rng.integers(0, len(base)))]
            r = rng.random()
            if r < 0.4:
                m = m[:int(rng.integers(10, n))] + bytearray(o[int(rng.integers(0, len(o))):])
            elif r < 0.7:
                m = m + bytearray(rng.integers(0, 256, int(rng.integers(1, 40))).astype(np.uint8).tobytes())
            else:
                m = m + bytearray(o[:int(rng.integers(1, 30))])
        elif kind == "valid_variant":       # streams zlib ACCEPTS, of unusual make: tiny blocks (memLevel 1), every strategy, several members
            parts = []
            for _ in range(int(rng.integers(1, 5))):
                t = _text(rng, ("fasta", "fastq", "runs", "noise")[int(rng.integers(0, 4))], int(rng.integers(1, 20000)))
                co = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, 16 + int(rng.integers(9, 16)), int(rng.integers(1, 10)),
                                      strategies[int(rng.integers(0, len(strategies)))])
                parts.append(co.compress(t) + co.flush() + b"\0" * int(rng.integers(0, 3) == 0))
            m = bytearray(b"".join(parts))
        else:                               # burst: a run of random bytes inside the stream
            at, ln = int(rng.integers(10, n)), int(rng.integers(1, 64))
            m[at:at + ln] = rng.integers(0, 256, len(m[at:at + ln])).astype(np.uint8).tobytes()
        out.append((kind, bytes(m)))
    return out


def test_twenty_thousand_mutated_members_never_leave_their_buffers(monkeypatch):
    from phenotypeseeker_amd.engine import PskContext
    rng = np.random.default_rng(20261004)
    base = _corpus(rng)
    assert all(zlib_all(b) is not None for b in base)
    n_total, per_call = 20_480, 2_560
    t_begin = time.time()
    tally = {}
    with PskContext(0) as ctx:
        for call in range(n_total // per_call):
            # chunk cuts of 16 KB (the default floor), and cuts every 2 / 4 KB: matches, block ends and member starts at chunk cuts
            monkeypatch.setenv("PSK_GZ_CHUNK", ("0", "2048", "4096", "0")[call % 4])
            muts = _mutants(rng, base, per_call - 16)
            files = [m for _, m in muts] + [base[int(rng.integers(0, len(base)))] for _ in range(16)]   # ... and good files among them
            kinds = [k for k, _ in muts] + ["good"] * 16
            want = [zlib_all(f) for f in files]
            t0 = time.time()
            texts, lens, routes, device_ms = ctx.gz_inflate(files, per_file=True)     # (a damaged guard band is a PskError: PSK_ESTATE)
            took = time.time() - t0
            assert took < 20.0 and device_ms < 10_000.0, (call, took, device_ms)      # no member keeps a lane spinning
            for i, (w, g, r) in enumerate(zip(want, texts, routes)):
                if w is None:
                    assert g is None and r == -1, (call, i, kinds[i], r, None if g is None else len(g))
                else:
                    assert g == w and r in (0, 1, 2) and lens[i] == len(w), (call, i, kinds[i], r, len(w), None if g is None else len(g))
                key = (kinds[i], "refused" if w is None else ("device" if r else "zlib"))
                tally[key] = tally.get(key, 0) + 1
            assert all(r in (1, 2) for r in routes[-16:]), routes[-16:]               # good files stay on the device among bad ones
        # the context is still good
        fa = _text(rng, "fasta", 200_000)
        texts, _, routes, _ = ctx.gz_inflate([gzip.compress(fa, 6)])
        assert texts[0] == fa and routes == [1]
    took = time.time() - t_begin
    print("gz fuzz: %d files in %.1f s: %s" % (n_total, took, ", ".join("%s/%s %d" % (k[0], k[1], v) for k, v in sorted(tally.items()))))
    # the mutators do what they say: every kind produced files zlib refuses AND files it accepts or the device decoded
    for kind in ("flips", "truncate", "header17", "lengths", "too_far_back", "trailer", "bsize", "stored", "gzip_header", "splice", "burst"):
        assert tally.get((kind, "refused"), 0) > 0, kind
    assert tally.get(("valid_variant", "refused"), 0) == 0 and tally.get(("valid_variant", "device"), 0) > 300
    assert sum(v for (k, how), v in tally.items() if how == "device") > 1000
    assert took < 120.0, took


def test_a_member_whose_text_crosses_four_gigabytes():
    """ISIZE is the length modulo 2^32 and the decoder's positions are 32-bit inside a chunk: 4 GiB + 1 MiB of text in one
    member (a few megabytes compressed) comes back whole -- from the device or, declined, from zlib -- with its true length."""
    from phenotypeseeker_amd.engine import PskContext
    piece = (b"ACGTTGCA" * 8 + b"\n") * 16384          # 1 MiB + ... of a very compressible line pattern
    n_pieces = (4 << 30) // len(piece) + 2
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    gz = b"".join(co.compress(piece) for _ in range(n_pieces)) + co.flush()
    total = n_pieces * len(piece)
    assert total > (4 << 30)
    with PskContext(0) as ctx:
        t0 = time.time()
        _, lens, routes, _ = ctx.gz_inflate([gz], want_text=False)
        assert lens == [total] and routes[0] in (0, 1), (lens, routes)
        print("4-GiB member: %.1f MB compressed, route %d, %.1f s" % (len(gz) / 1e6, routes[0], time.time() - t0))
        # ... and a short good file right after it
        fa = _text(np.random.default_rng(5), "fasta", 100_000)
        texts, _, r2, _ = ctx.gz_inflate([gzip.compress(fa, 6)])
        assert texts[0] == fa and r2 == [1]


def test_files_without_a_dynamic_block_cost_the_call_a_bounded_time():
    """No bound on how long a malformed member may keep a lane spinning: a member of fixed-code blocks only, one
    of stored blocks only and one that is a single 12-MB stored-looking stretch of noise have no dynamic header for the search to
    find -- the first chunk's lane runs on until PSK_GZ_MAX_SPAN (1 MiB: ~0.75 s of one lane) and the file goes to zlib; files of
    five such members do not multiply that.  Texts equal zlib's, the good files among them stay on the device, the call is quick."""
    from phenotypeseeker_amd.engine import PskContext
    rng = np.random.default_rng(9)
    fa = _text(rng, "fasta", 6_000_000)
    noise = rng.integers(0, 256, 12_000_000, dtype=np.uint8).tobytes()
    fixed = _member(fa, 6, zlib.Z_FIXED)
    stored = _member(noise, 0)
    many = b"".join(_member(fa[i * 1_000_000:(i + 1) * 1_000_000], 6, zlib.Z_FIXED) for i in range(5))
    good = gzip.compress(fa[:2_000_000], 6)
    files = [good, fixed, stored, many, gzip.compress(noise, 6), good]
    want = [zlib_all(f) for f in files]
    with PskContext(0) as ctx:
        ctx.gz_inflate([good])          # (the context's first call pays the allocations)
        t0 = time.time()
        texts, lens, routes, device_ms = ctx.gz_inflate(files)
        took = time.time() - t0
    # (`many`: its members are under the span each -- ~0.3 MB of fixed-code blocks -- so one lane a member may decode them after all)
    assert texts == want and routes[0] == 1 and routes[-1] == 1 and routes[1] == routes[2] == routes[4] == 0 and routes[3] in (0, 1), routes
    print("files without a dynamic block: %.2f s for the call, %.0f ms on the device" % (took, device_ms))
    assert device_ms < 4000.0 and took < 12.0, (took, device_ms)
