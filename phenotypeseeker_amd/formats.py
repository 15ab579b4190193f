"""Text / binary formats at the edges of the hot path: k-mer words <-> strings, GenomeTester4
.list files, the reference's number formatting (modeling.py:796,:739)."""
import gzip

import numpy as np

LIST_DTYPE = np.dtype([("word", "<u8"), ("freq", "<u4")])  # packed 12-byte records
LIST_MAGIC = 0x47543443

_CODES = np.frombuffer(b"ACGT", dtype=np.uint8)


def words_to_kmers(words, k):
    """u64 words (first base most significant) -> list of k-mer strings."""
    words = np.asarray(words, dtype=np.uint64)
    if len(words) == 0:
        return []
    shifts = np.arange(k - 1, -1, -1, dtype=np.uint64) * np.uint64(2)
    codes = ((words[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.uint8)
    chars = _CODES[codes]
    return [row.tobytes().decode() for row in chars]


def kmer_to_word(kmer):
    w = 0
    for ch in kmer.upper():
        w = (w << 2) | {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3}[ch]
    return w


def canonical(word, k):
    rc, w = 0, int(word)
    for _ in range(k):
        rc = (rc << 2) | (3 - (w & 3))
        w >>= 2
    return min(int(word), rc)


def write_list(path, k, words, freqs):
    """GenomeTester4 list file, byte-compatible with glistmaker 4.2.3 output."""
    words = np.asarray(words, dtype="<u8")
    freqs = np.asarray(freqs, dtype="<u4")
    rec = np.empty(len(words), dtype=LIST_DTYPE)
    rec["word"], rec["freq"] = words, freqs
    with open(path, "wb") as f:
        f.write(np.array([LIST_MAGIC, 4, 2, int(k)], dtype="<u4").tobytes())
        f.write(np.array([len(words), int(freqs.sum(dtype=np.uint64)), 40], dtype="<u8").tobytes())
        f.write(rec.tobytes())


def read_list(path):
    with open(path, "rb") as f:
        data = f.read()
    h32 = np.frombuffer(data, dtype="<u4", count=4)
    if len(data) < 40 or int(h32[0]) != LIST_MAGIC:
        raise ValueError("%s is not a GenomeTester4 list file" % path)
    h64 = np.frombuffer(data, dtype="<u8", count=3, offset=16)
    rec = np.frombuffer(data, dtype=LIST_DTYPE, count=int(h64[0]), offset=int(h64[2]))
    return int(h32[3]), rec["word"].copy(), rec["freq"].copy()


def is_gzip(path):
    """gzip by magic bytes, like read_sequence_file"""
    with open(path, "rb") as f:
        return f.read(2) == b"\x1f\x8b"


def read_sequence_file(path):
    """The text of a sequence file; .gz (by magic bytes) inflated here.  Since r05 only the routes that want the text on the
    host call this (PSK_NO_GPU_GZ=1, one sample at a time, the --kmerDB file): the counting calls take .gz images as they are."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] == b"\x1f\x8b":
        data = gzip.decompress(data)
    return data


def round2(x):
    """round(np.float64, 2) -- numpy rounding (rint(x*100)/100), what the reference's round() does
    to the numpy scalars it holds (modeling.py:739,:796)."""
    return float(np.round(np.float64(x), 2))


def pstring(p):
    return "%.2E" % p
