"""CPU oracle for the MinHash sketch of the -w path (pure Python; small inputs only).
TEST INFRASTRUCTURE ONLY.  Restates Mash 2.2's sketching (third-party, bundled as bin/mash; source
not vendored): canonical k-mer = min(k-mer, reverse complement) as strings, MurmurHash3_x64_128
(seed 42) of the ASCII k-mer, first 8 digest bytes, bottom-s distinct hashes.  Pinned by
tests/golden/mash.json (hash lists from `mash info -d`, distances from `mash dist`)."""

M64 = (1 << 64) - 1


def _rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M64


def _fmix(k):
    k ^= k >> 33
    k = (k * 0xff51afd7ed558ccd) & M64
    k ^= k >> 33
    k = (k * 0xc4ceb9fe1a85ec53) & M64
    k ^= k >> 33
    return k


def murmur3_x64_128(data, seed):
    h1 = h2 = seed
    c1, c2 = 0x87c37b91114253d5, 0x4cf5ad432745937f
    n = len(data)
    nb = n // 16
    for i in range(nb):
        k1 = int.from_bytes(data[16 * i:16 * i + 8], "little")
        k2 = int.from_bytes(data[16 * i + 8:16 * i + 16], "little")
        k1 = (k1 * c1) & M64; k1 = _rotl(k1, 31); k1 = (k1 * c2) & M64; h1 ^= k1
        h1 = _rotl(h1, 27); h1 = (h1 + h2) & M64; h1 = (h1 * 5 + 0x52dce729) & M64
        k2 = (k2 * c2) & M64; k2 = _rotl(k2, 33); k2 = (k2 * c1) & M64; h2 ^= k2
        h2 = _rotl(h2, 31); h2 = (h2 + h1) & M64; h2 = (h2 * 5 + 0x38495ab5) & M64
    tail = data[16 * nb:]
    if len(tail) > 8:
        k2 = int.from_bytes(tail[8:], "little")
        k2 = (k2 * c2) & M64; k2 = _rotl(k2, 33); k2 = (k2 * c1) & M64; h2 ^= k2
    if len(tail) > 0:
        k1 = int.from_bytes(tail[:8], "little")
        k1 = (k1 * c1) & M64; k1 = _rotl(k1, 31); k1 = (k1 * c2) & M64; h1 ^= k1
    h1 ^= n; h2 ^= n
    h1 = (h1 + h2) & M64; h2 = (h2 + h1) & M64
    h1 = _fmix(h1); h2 = _fmix(h2)
    h1 = (h1 + h2) & M64; h2 = (h2 + h1) & M64
    return h1, h2


_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def sketch(fasta_bytes, k=21, sketch_size=1000, seed=42):
    """Bottom-s sketch of a plain FASTA (records separated at '>'; any non-ACGT letter breaks)."""
    hashes = set()
    for rec in fasta_bytes.decode().split(">")[1:]:
        seq = "".join(rec.split("\n")[1:]).upper()
        for i in range(len(seq) - k + 1):
            f = seq[i:i + k]
            if any(ch not in _COMP for ch in f):
                continue
            r = "".join(_COMP[ch] for ch in reversed(f))
            h = murmur3_x64_128(min(f, r).encode(), seed)[0]
            hashes.add(h if k > 16 else h & 0xffffffff)
    return sorted(hashes)[:sketch_size]
