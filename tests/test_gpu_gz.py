"""Row a1's "(.gz)": DEFLATE decoded on the device (csrc/gz_inflate.hip) against zlib -- byte for byte.

glistmaker reads .gz through zlib (Appendix B: "`.gz` accepted, output cmp-identical to plain"),
so the checker is zlib itself (Python's gzip / zlib modules); the k-mer lists of .gz inputs are compared with the lists of
the same text uploaded plain (which the glistmaker fixtures pin).  Every case asserts the ROUTE as well: a device decoder
that quietly declined everything would pass a comparison of texts through its zlib fall-back."""
import gzip
import io
import os
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _device_route_for_small_inputs(monkeypatch):
    """The library sends a few small files through zlib on host threads (the device route has a floor of ~40 ms: a DEFLATE block is
    decoded by one lane: the device wins by numbers only); the tests' inputs are small, and they are about the device."""
    monkeypatch.setenv("PSK_GZ_DEVICE_MIN_MB", "0")


def _fasta(n_bases, seed, width=60):
    rng = np.random.default_rng(seed)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n_bases)]
    # genes repeated here and there: matches at long distances
    for _ in range(n_bases // 5000):
        a, b, ln = int(rng.integers(0, n_bases - 600)), int(rng.integers(0, n_bases - 600)), int(rng.integers(30, 500))
        seq[b:b + ln] = seq[a:a + ln]
    lines = [b">contig_%d some description" % seed]
    s = seq.tobytes()
    lines += [s[i:i + width] for i in range(0, len(s), width)]
    return b"\n".join(lines) + b"\n"


def _fastq(n_reads, seed, rl=150):
    rng = np.random.default_rng(seed)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 200_000)]
    out = io.BytesIO()
    for i in range(n_reads):
        at = int(rng.integers(0, len(genome) - rl))
        q = (rng.integers(0, 41, rl) + 33).astype(np.uint8)
        out.write(b"@read_%d/1\n" % i + genome[at:at + rl].tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    return out.getvalue()


def _bgzf(text, block=0xff00):
    """BGZF as bgzip writes it: members of at most 64 KB with a 'BC' extra field holding their length, then the empty
    end-of-file member."""
    out = io.BytesIO()
    pieces = [text[i:i + block] for i in range(0, len(text), block)] + [b""]
    for piece in pieces:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = co.compress(piece) + co.flush()
        bsize = 12 + 6 + len(raw) + 8
        out.write(b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1))
        out.write(raw + struct.pack("<II", zlib.crc32(piece), len(piece) & 0xffffffff))
    return out.getvalue()


def _named(text, level=6):
    b = io.BytesIO()
    with gzip.GzipFile(filename="reads_of_sample_1.fastq", mode="wb", fileobj=b, compresslevel=level, mtime=0) as f:
        f.write(text)
    return b.getvalue()


def _cases():
    rng = np.random.default_rng(7)
    fa, fq = _fasta(400_000, 1), _fastq(3000, 2)
    cases = {
        "fasta level 6": gzip.compress(fa, 6),
        "fasta level 1": gzip.compress(fa, 1),
        "fasta level 9": gzip.compress(fa, 9),
        "fastq level 6": gzip.compress(fq, 6),
        "fastq with a name in the header": _named(fq),
        "empty text": gzip.compress(b""),
        "one byte": gzip.compress(b">"),
        "incompressible (stored blocks)": gzip.compress(rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes(), 6),
        "one long run (distance 1)": gzip.compress(b"A" * 300_000 + b"\n", 9),
        "short period": gzip.compress(b"ACGTTGCA" * 40_000, 6),
        "fixed-code block": gzip.compress(b">s\nACGT\n", 6),
        "two members": gzip.compress(fa[:150_000]) + gzip.compress(fq[:100_000]),
        "five members and padding": b"".join(gzip.compress(fa[i * 40_000:(i + 1) * 40_000], 4) for i in range(5)) + b"\0" * 37,
        "bgzf": _bgzf(fq),
        "bgzf of nothing": _bgzf(b""),
        "level 0": gzip.compress(fa[:100_000], 0),
    }
    # zlib's other strategies and memory levels: Huffman codes only (no match), runs only (distance 1), the FIXED code for every
    # block, small blocks (memLevel 1: a block every few hundred symbols -- thousands of headers), large ones
    for name, level, mem, strategy in (("huffman only", 6, 8, zlib.Z_HUFFMAN_ONLY), ("rle", 6, 8, zlib.Z_RLE), ("fixed code", 6, 8, zlib.Z_FIXED),
                                       ("filtered", 6, 8, zlib.Z_FILTERED), ("memLevel 1", 6, 1, zlib.Z_DEFAULT_STRATEGY),
                                       ("memLevel 9 level 9", 9, 9, zlib.Z_DEFAULT_STRATEGY), ("fixed code, small blocks", 1, 1, zlib.Z_FIXED)):
        co = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strategy)
        cases["zlib " + name] = co.compress(fq[:300_000] + fa[:200_000]) + co.flush()
    # sync and full flushes in the middle of a stream (empty stored blocks; a full flush also forgets the window)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = []
    for i in range(0, 400_000, 50_000):
        parts.append(co.compress(fa[i:i + 50_000]))
        parts.append(co.flush(zlib.Z_SYNC_FLUSH if (i // 50_000) % 2 else zlib.Z_FULL_FLUSH))
    cases["zlib with sync and full flushes"] = b"".join(parts) + co.flush()
    # GNU gzip has a deflate of its own (not zlib's: other block sizes, other choices of match); --rsyncable restarts often
    import shutil
    import subprocess
    if shutil.which("gzip"):
        for flags in (["-1"], ["-6"], ["-9"], ["-6", "--rsyncable"]):
            r = subprocess.run(["gzip", "-c"] + flags, input=fq + fa, capture_output=True)
            if r.returncode == 0:
                cases["GNU gzip " + " ".join(flags)] = r.stdout
    return cases


@pytest.mark.parametrize("chunk", [0, 4096, 20_000])
def test_device_inflate_equals_zlib(chunk, monkeypatch):
    """Every case in ONE call (files of a group are laid out and decoded together); chunk: the default cut (one chunk per
    32 KB of a small file) and two small ones, so that texts of a few hundred KB are cut dozens of times."""
    from phenotypeseeker_amd.engine import PskContext
    if chunk:
        monkeypatch.setenv("PSK_GZ_CHUNK", str(chunk))
    cases = _cases()
    names = list(cases)
    with PskContext(0) as ctx:
        texts, lens, routes, _ = ctx.gz_inflate([cases[k] for k in names])
    for name, text, ln, route in zip(names, texts, lens, routes):
        want = gzip.decompress(cases[name])
        assert ln == len(want), (name, ln, len(want))
        assert text == want, (name, next(i for i in range(len(want)) if text[i] != want[i]))
        assert route == (2 if name.startswith("bgzf") else 1), (name, route)


def test_many_chunks_and_links(monkeypatch):
    """A 6-MB text: ~45 chunks by default; with a small cut several hundred, most of them starting on a header the search
    found, every one linked to the next by the counting pass."""
    from phenotypeseeker_amd.engine import PskContext
    fa = _fasta(6_000_000, 11)
    gz = gzip.compress(fa, 6)
    for chunk in (0, 8192):
        if chunk:
            monkeypatch.setenv("PSK_GZ_CHUNK", str(chunk))
        with PskContext(0) as ctx:
            texts, lens, routes, _ = ctx.gz_inflate([gz, gz[:]])
        assert routes == [1, 1] and texts[0] == fa and texts[1] == fa


def test_corrupt_and_truncated_files_are_errors():
    """What zlib refuses, the device route hands to zlib, and zlib's words are the error."""
    from phenotypeseeker_amd.engine import PskContext, PskError
    fa = _fasta(200_000, 3)
    gz = bytearray(gzip.compress(fa, 6))
    with PskContext(0) as ctx:
        bad = bytes(gz[:len(gz) // 2])
        with pytest.raises(PskError, match="not a valid gzip file"):
            ctx.gz_inflate([bad])
        flipped = bytearray(gz)
        flipped[len(gz) // 2] ^= 0x55
        try:
            texts, _, routes, _ = ctx.gz_inflate([bytes(flipped)])
        except PskError as e:
            assert "not a valid gzip file" in str(e)
        else:   # a flip zlib does not notice before the check sums either way: then the texts agree
            assert texts[0] == zlib.decompress(bytes(flipped), 31)
        with pytest.raises(PskError, match="not a valid gzip file"):
            ctx.gz_inflate([b"\x1f\x8b\x08\x00 this is not deflate data, but it is long enough to be looked at"])
        # a flip that leaves a valid DEFLATE stream (a byte of a stored block; a literal of a compressed one): only the
        # check sum of the member knows -- zlib's "incorrect data check", and the device route's
        stored = bytearray(gzip.compress(fa[:50_000], 0))
        stored[len(stored) // 2] ^= 0x01
        with pytest.raises(PskError, match="incorrect data check"):
            ctx.gz_inflate([bytes(stored)])
        for at in range(len(gz) // 3, len(gz) // 3 + 40):
            flipped = bytearray(gz)
            flipped[at] ^= 0x04
            try:
                want = zlib.decompress(bytes(flipped), 31)
            except zlib.error:
                want = None
            try:
                got = ctx.gz_inflate([bytes(flipped)])[0][0]
            except PskError:
                got = None
            assert got == want, at
        # and the context is still good
        texts, _, routes, _ = ctx.gz_inflate([bytes(gz)])
        assert texts[0] == fa and routes == [1]


def _multi_line_fastq(n_reads, seed):
    """Sequence and quality wrapped over two lines each: not four-line FASTQ, so the GPU framing hands the sample to the
    host's state machine."""
    rng = np.random.default_rng(seed)
    out = io.BytesIO()
    for i in range(n_reads):
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 120)].tobytes()
        out.write(b"@r%d\n" % i + seq[:70] + b"\n" + seq[70:] + b"\n+\n" + b"I" * 70 + b"\n" + b"I" * 50 + b"\n")
    return out.getvalue()


@pytest.mark.parametrize("k", [13, 16, 21])
def test_counting_gz_samples_equals_counting_their_text(k, tmp_path, monkeypatch):
    """.gz samples through the batch counter -- as images in memory and as files, mixed with plain samples, with sketches --
    give the lists and sketches of their text; so do two runs of a call cut by a small PSK_GZ_GROUP_MB, an empty member and a
    multi-line FASTQ file whose text has to come back for the host's state machine."""
    from phenotypeseeker_amd.engine import PskContext
    texts = [_fasta(150_000, 21), _fastq(1500, 22), _fasta(90_000, 23), b"", _multi_line_fastq(400, 24), _fastq(900, 25), b">only a header\n",
             _fasta(60_000, 26)]
    packed = [gzip.compress(t, 6) if i != 2 else t for i, t in enumerate(texts)]     # sample 2 stays plain
    packed[5] = _bgzf(texts[5])
    n = len(texts)
    sk = (21, 1000, 42)
    with PskContext(0) as ctx:
        ctx.begin(k, n)
        nu0, nt0, sk0 = ctx.count_kmers_batch(0, texts, 4, sketch=sk)
        want = [ctx.get_list(i, nu0[i]) for i in range(n)]
        for group_mb, one_by_one in ((None, False), ("1", False), ("1", True)):
            if group_mb:
                monkeypatch.setenv("PSK_GZ_GROUP_MB", group_mb)     # (1 MB of text a run: the call is cut into several --
            if one_by_one:                                          # read | inflated | counted as a pipeline, or one after the other)
                monkeypatch.setenv("PSK_GZ_NO_LOOKAHEAD", "1")
            ctx.begin(k, n)
            nu1, nt1, sk1 = ctx.count_kmers_batch(0, packed, 4, sketch=sk)
            assert list(nu1) == list(nu0) and list(nt1) == list(nt0)
            for i in range(n):
                w, f = ctx.get_list(i, nu1[i])
                assert np.array_equal(w, want[i][0]) and np.array_equal(f, want[i][1]), (i, group_mb)
                assert np.array_equal(sk1[i], sk0[i]), i
        monkeypatch.delenv("PSK_GZ_NO_LOOKAHEAD")
        # the same from files, one sample at a time as well
        paths = []
        for i, b in enumerate(packed):
            p = os.path.join(tmp_path, "s%d.%s" % (i, "seq" if i % 2 else "seq.gz"))     # (the name says nothing: magic bytes decide)
            with open(p, "wb") as f:
                f.write(b)
            paths.append(p)
        # (r06: .gz files are MAPPED read-only and uploaded out of the mapping; PSK_GZ_READ=1 is r05's route -- read into host buffers of
        # the library -- which a file that cannot be mapped still takes: both, and a call cut into runs so that mappings are given back
        # while later runs are read)
        monkeypatch.delenv("PSK_GZ_GROUP_MB", raising=False)
        for read_route, group_mb in ((None, None), ("1", None), (None, "1")):
            if read_route:
                monkeypatch.setenv("PSK_GZ_READ", read_route)
            if group_mb:
                monkeypatch.setenv("PSK_GZ_GROUP_MB", group_mb)
            ctx.begin(k, n)
            nu2, nt2 = ctx.count_kmers_files(0, paths, 4)
            monkeypatch.delenv("PSK_GZ_READ", raising=False)
            assert list(nu2) == list(nu0) and list(nt2) == list(nt0)
            for i in range(n):
                w, f = ctx.get_list(i, nu2[i])
                assert np.array_equal(w, want[i][0]) and np.array_equal(f, want[i][1]), (i, read_route, group_mb)
        monkeypatch.delenv("PSK_GZ_GROUP_MB", raising=False)
        ctx.begin(k, 1)
        nu3, nt3 = ctx.count_kmers(0, packed[1])
        w, f = ctx.get_list(0, nu3)
        assert nu3 == nu0[1] and np.array_equal(w, want[1][0]) and np.array_equal(f, want[1][1])


def test_without_the_device_route_files_are_refused_as_before(tmp_path, monkeypatch):
    """PSK_NO_GPU_GZ=1: a .gz FILE is answered with PSK_EGZIP (the caller inflates: modeling.py's thread pool), as until r04."""
    from phenotypeseeker_amd._lib import PSK_EGZIP
    from phenotypeseeker_amd.engine import PskContext, PskError
    p = os.path.join(tmp_path, "a.fasta.gz")
    with open(p, "wb") as f:
        f.write(gzip.compress(_fasta(50_000, 5)))
    monkeypatch.setenv("PSK_NO_GPU_GZ", "1")
    with PskContext(0) as ctx:
        ctx.begin(13, 1)
        with pytest.raises(PskError) as e:
            ctx.count_kmers_files(0, [p], 2)
        assert e.value.code == PSK_EGZIP


def test_small_groups_go_through_zlib_on_host_threads(monkeypatch):
    """The default: a few small files are inflated by zlib on host threads (route 0: the library's estimate of both routes says
    that is faster), larger groups on the device (route 1) -- the same text either way."""
    from phenotypeseeker_amd.engine import PskContext
    fa = _fasta(300_000, 31)
    images = [gzip.compress(fa, 6), gzip.compress(fa[:100_000], 1), _bgzf(fa[:50_000])]
    with PskContext(0) as ctx:
        monkeypatch.delenv("PSK_GZ_DEVICE_MIN_MB")
        texts, _, routes, _ = ctx.gz_inflate(images)
        assert routes == [0, 0, 0] and texts == [fa, fa[:100_000], fa[:50_000]]
        monkeypatch.setenv("PSK_GZ_DEVICE_MIN_MB", "0")
        texts, _, routes, _ = ctx.gz_inflate(images)
        assert routes == [1, 1, 2] and texts == [fa, fa[:100_000], fa[:50_000]]


def test_dictionary_counting_of_gz_samples(tmp_path, monkeypatch):
    """`prediction`'s counting (psk_count_dict_files / _batch) takes .gz samples as they are as well; a call cut into several
    inflate runs still fills every sample's row."""
    from phenotypeseeker_amd.engine import PskContext
    k = 13
    texts = [_fasta(120_000, 40 + i) for i in range(5)] + [_fastq(800, 50)]
    with PskContext(0) as ctx:
        ctx.begin(k, 1)
        nu, _ = ctx.count_kmers(0, texts[0])
        words = ctx.get_list(0, nu)[0][::97][:300].copy()
        want = ctx.count_dict_batch(texts, k, words, 4)
        assert int(want.sum()) > 0
        packed = [gzip.compress(t, 6) for t in texts]
        paths = []
        for i, b in enumerate(packed):
            p = os.path.join(tmp_path, "p%d.fa.gz" % i)
            with open(p, "wb") as f:
                f.write(b)
            paths.append(p)
        for group_mb in (None, "1"):
            if group_mb:
                monkeypatch.setenv("PSK_GZ_GROUP_MB", group_mb)
            assert np.array_equal(ctx.count_dict_batch(packed, k, words, 4), want)
            assert np.array_equal(ctx.count_dict_files(paths, k, words, 4), want)


def test_a_call_of_nothing_but_empty_texts():
    """(found by tests/_stress.py: with no text at all there is nothing for the check-sum kernel to be launched over)"""
    from phenotypeseeker_amd.engine import PskContext
    empty = gzip.compress(b"")
    with PskContext(0) as ctx:
        texts, lens, routes, _ = ctx.gz_inflate([empty, empty])
        assert texts == [b"", b""] and lens == [0, 0] and routes == [1, 1]
        ctx.begin(13, 2)
        nu, nt = ctx.count_kmers_batch(0, [empty, empty], 2)
        assert list(nu) == [0, 0] and list(nt) == [0, 0]


def test_a_file_the_device_declines_goes_through_zlib(monkeypatch):
    """Incompressible data: stored blocks only, no dynamic header for the search to find, so the first chunk runs on and on --
    beyond PSK_GZ_MAX_SPAN the device gives the file to zlib (route 0), the others of the call stay on the device."""
    from phenotypeseeker_amd.engine import PskContext
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, 600_000, dtype=np.uint8).tobytes()
    fa = _fasta(200_000, 61)
    monkeypatch.setenv("PSK_GZ_MAX_SPAN", "100000")
    monkeypatch.setenv("PSK_GZ_CHUNK", "16384")
    with PskContext(0) as ctx:
        texts, _, routes, _ = ctx.gz_inflate([gzip.compress(fa, 6), gzip.compress(noise, 6), gzip.compress(fa[::-1], 6)])
    assert routes == [1, 0, 1] and texts == [fa, noise, fa[::-1]]


def test_a_corrupt_sample_in_the_middle_of_a_call_cut_into_runs(monkeypatch):
    """The stages of a call's runs are threads (read | inflate | count): the failure of one of them ends the call with its
    words, nothing hangs, and the context goes on working."""
    from phenotypeseeker_amd.engine import PskContext, PskError
    texts = [_fasta(80_000, 70 + i) for i in range(6)]
    packed = [gzip.compress(t, 6) for t in texts]
    packed[3] = packed[3][:len(packed[3]) // 2]
    monkeypatch.setenv("PSK_GZ_GROUP_MB", "1")      # (runs of one or two samples)
    with PskContext(0) as ctx:
        ctx.begin(13, 6)
        with pytest.raises(PskError, match="not a valid gzip file"):
            ctx.count_kmers_batch(0, packed, 4)
        ctx.begin(13, 6)
        nu, nt = ctx.count_kmers_batch(0, [gzip.compress(t, 6) for t in texts], 4)
        ctx.begin(13, 6)
        nu0, nt0 = ctx.count_kmers_batch(0, texts, 4)
        assert list(nu) == list(nu0) and list(nt) == list(nt0)
