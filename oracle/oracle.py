"""Python face of the CPU oracle (ctypes over oracle/libpsk_oracle.so + numpy set algebra).

TEST INFRASTRUCTURE ONLY -- see the header of psk_oracle.c.  The product package
(phenotypeseeker_amd/) never imports this module; tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg do, as the checker.

Every function cites the reference lines it restates (modeling.py = /root/reference/
PhenotypeSeeker/modeling.py).  Pinned against tests/golden/ by tests/test_oracle_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpsk_oracle.so")

LIST_DTYPE = np.dtype([("word", "<u8"), ("freq", "<u4")])  # packed, 12 bytes


def build():
    """Compile libpsk_oracle.so with gcc (building the checker is not using it)."""
    subprocess.run(["make", "-C", _HERE, "libpsk_oracle.so"], check=True, stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        c = ctypes
        L.orc_count_kmers.argtypes = [c.c_char_p, c.c_size_t, c.c_int, c.POINTER(c.c_void_p), c.POINTER(c.c_void_p),
                                      c.POINTER(c.c_uint64), c.POINTER(c.c_uint64)]
        L.orc_count_kmers.restype = c.c_int
        L.orc_free.argtypes = [c.c_void_p]
        L.orc_free.restype = None
        L.orc_chi2_row.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int,
                                   c.POINTER(c.c_double), c.POINTER(c.c_double), c.POINTER(c.c_int)]
        L.orc_chi2_row.restype = c.c_int
        L.orc_chi2_keep.argtypes = [c.c_double, c.c_double, c.c_int, c.c_uint64]
        L.orc_chi2_keep.restype = c.c_int
        L.orc_chi2_scan.argtypes = [c.c_void_p, c.c_uint64, c.c_int, c.c_void_p, c.c_void_p, c.c_int, c.c_int,
                                    c.c_int, c.c_double, c.c_int, c.c_uint64, c.c_void_p, c.c_void_p, c.c_void_p,
                                    c.c_void_p]
        L.orc_chi2_scan.restype = None
        L.orc_chi2_scan_mt.argtypes = L.orc_chi2_scan.argtypes + [c.c_int]
        L.orc_chi2_scan_mt.restype = c.c_int
        L.orc_ttest_row.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int,
                                    c.POINTER(c.c_double), c.POINTER(c.c_double), c.POINTER(c.c_double),
                                    c.POINTER(c.c_double), c.POINTER(c.c_int)]
        L.orc_ttest_row.restype = c.c_int
        L.orc_ttest_scan.argtypes = [c.c_void_p, c.c_uint64, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int,
                                     c.c_int, c.c_int, c.c_double, c.c_uint64, c.c_void_p, c.c_void_p, c.c_void_p,
                                     c.c_void_p, c.c_void_p, c.c_void_p]
        L.orc_ttest_scan.restype = None
        L.orc_t_two_sided_p.argtypes = [c.c_double, c.c_double]
        L.orc_t_two_sided_p.restype = c.c_double
        L.orc_betainc.argtypes = [c.c_double, c.c_double, c.c_double]
        L.orc_betainc.restype = c.c_double
        L.orc_count_dict.argtypes = [c.c_char_p, c.c_size_t, c.c_int, c.c_void_p, c.c_uint64, c.c_void_p]
        L.orc_count_dict.restype = c.c_int
        _lib = L
    return _lib


# --------------------------------------------------------------------------------------------
# k-mer plane
# --------------------------------------------------------------------------------------------
def count_kmers(buf, k):
    """glistmaker restatement (modeling.py:303-315).  buf: inflated FASTA/FASTQ bytes.
    Returns (words u64[U] ascending, freqs u32[U], n_total)."""
    L = lib()
    wp, fp = ctypes.c_void_p(), ctypes.c_void_p()
    nu, nt = ctypes.c_uint64(), ctypes.c_uint64()
    buf = bytes(buf)
    rc = L.orc_count_kmers(buf, len(buf), int(k), ctypes.byref(wp), ctypes.byref(fp), ctypes.byref(nu),
                           ctypes.byref(nt))
    if rc != 0:
        raise RuntimeError("orc_count_kmers failed: %d" % rc)
    n = nu.value
    words = np.ctypeslib.as_array(ctypes.cast(wp, ctypes.POINTER(ctypes.c_uint64)), shape=(max(n, 1),))[:n].copy()
    freqs = np.ctypeslib.as_array(ctypes.cast(fp, ctypes.POINTER(ctypes.c_uint32)), shape=(max(n, 1),))[:n].copy()
    L.orc_free(wp)
    L.orc_free(fp)
    return words, freqs, nt.value


def list_bytes(k, words, freqs):
    """The GenomeTester4 .list file image."""
    words = np.asarray(words, dtype="<u8")
    freqs = np.asarray(freqs, dtype="<u4")
    hdr = np.array([0x47543443, 4, 2, int(k)], dtype="<u4").tobytes()
    hdr += np.array([len(words), int(freqs.sum(dtype=np.uint64)), 40], dtype="<u8").tobytes()
    rec = np.empty(len(words), dtype=LIST_DTYPE)
    rec["word"] = words
    rec["freq"] = freqs
    return hdr + rec.tobytes()


def parse_list(data):
    """Inverse of list_bytes: returns (k, words, freqs, n_total)."""
    h32 = np.frombuffer(data, dtype="<u4", count=4)
    assert h32[0] == 0x47543443, "not a GenomeTester4 list"
    h64 = np.frombuffer(data, dtype="<u8", count=3, offset=16)
    rec = np.frombuffer(data, dtype=LIST_DTYPE, count=int(h64[0]), offset=int(h64[2]))
    return int(h32[3]), rec["word"].copy(), rec["freq"].copy(), int(h64[1])


_BASES = "ACGT"


def word_to_kmer(word, k):
    word = int(word)
    return "".join(_BASES[(word >> (2 * (k - 1 - i))) & 3] for i in range(k))


def kmer_to_word(kmer):
    w = 0
    for ch in kmer:
        w = (w << 2) | "ACGT".index(ch.upper().replace("U", "T"))
    return w


def revcomp_word(word, k):
    r = 0
    w = int(word)
    for _ in range(k):
        r = (r << 2) | (3 - (w & 3))
        w >>= 2
    return r


def canonical_word(word, k):
    return min(int(word), revcomp_word(word, k))


def union(word_lists):
    """glistcompare -u tree (modeling.py:350-380): only the key set is used downstream."""
    if not word_lists:
        return np.zeros(0, dtype=np.uint64)
    return np.unique(np.concatenate([np.asarray(w, dtype=np.uint64) for w in word_lists]))


def union_freqs(lists):
    """glistcompare -u sums the frequencies of equal words."""
    words = np.concatenate([np.asarray(w, dtype=np.uint64) for w, _ in lists])
    freqs = np.concatenate([np.asarray(f, dtype=np.uint64) for _, f in lists])
    uw, inv = np.unique(words, return_inverse=True)
    uf = np.zeros(len(uw), dtype=np.uint64)
    np.add.at(uf, inv, freqs)
    return uw, uf.astype(np.uint32)


def intersect(a, b):
    """glistcompare -i (modeling.py:371, --kmerDB)."""
    return np.intersect1d(np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64))


def map_counts(words, freqs, feature_words):
    """glistquery <sample>.list -l feature_vector.list (modeling.py:324-329): for every
    k-mer of the feature vector, in order, its frequency in the sample (0 if absent)."""
    words = np.asarray(words, dtype=np.uint64)
    idx = np.searchsorted(words, feature_words)
    idx_c = np.minimum(idx, max(len(words) - 1, 0))
    hit = (idx < len(words)) & (words[idx_c] == feature_words) if len(words) else np.zeros(len(feature_words), bool)
    out = np.zeros(len(feature_words), dtype=np.uint32)
    out[hit] = np.asarray(freqs)[idx_c[hit]]
    return out


def words_per_row(n_samples):
    return (n_samples + 63) // 64


def presence_bits(sample_word_lists, feature_words, wpr=None):
    """Bit-packed k-mer x sample presence matrix: row r <-> feature_words[r],
    sample i <-> bit (i & 63) of word (i >> 6).  presence = count > 0 (modeling.py:694-695)."""
    n = len(sample_word_lists)
    wpr = wpr or words_per_row(n)
    bits = np.zeros((len(feature_words), wpr), dtype=np.uint64)
    for i, words in enumerate(sample_word_lists):
        words = np.asarray(words, dtype=np.uint64)
        idx = np.searchsorted(feature_words, words)
        ok = idx < len(feature_words)
        ok[ok] &= feature_words[idx[ok]] == words[ok]
        bits[idx[ok], i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    return bits


# --------------------------------------------------------------------------------------------
# association scans
# --------------------------------------------------------------------------------------------
def _pheno_i8(pheno):
    """pheno: sequence of 1 / 0 / 'NA' (or None / -1) -> int8 with -1 for NA (modeling.py:122-126)."""
    out = np.empty(len(pheno), dtype=np.int8)
    for i, v in enumerate(pheno):
        if v in ("NA", None, -1):
            out[i] = -1
        else:
            out[i] = int(v)
    return out


def chi2_row(presence, pheno, weights, min_samples, max_samples):
    """phenotypes.conduct_chi_squared_test up to the p-value (modeling.py:759-794).
    Returns None when the frequency filter (:770-772) drops the row, else (chi2, p, n_with)."""
    L = lib()
    pres = np.ascontiguousarray(np.asarray(presence) != 0, dtype=np.uint8)
    ph = _pheno_i8(pheno)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    c, p, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
    ok = L.orc_chi2_row(pres.ctypes.data, ph.ctypes.data, w.ctypes.data, len(pres), int(min_samples),
                        int(max_samples), ctypes.byref(c), ctypes.byref(p), ctypes.byref(n))
    if not ok:
        return None
    return c.value, p.value, n.value


def chi2_keep(p, pvalue_cutoff, omit_B, n_kmers):
    """modeling.py:795."""
    return bool(lib().orc_chi2_keep(float(p), float(pvalue_cutoff), int(bool(omit_B)), int(n_kmers)))


def chi2_scan(bits, pheno, weights, n_samples, min_samples, max_samples, pvalue_cutoff, omit_B, n_kmers, n_threads=1,
              scratch=None):
    """The hot loop modeling.py:677-714 over a bit matrix.  Returns dict of per-row arrays.  n_threads > 1: the rows
    are cut into that many ranges, one POSIX thread each (orc_chi2_scan_mt).  scratch: a dict this call fills with
    its output arrays and a later call of the same shape reuses (timed passes then measure the scan, not 21 bytes
    per row of fresh zero pages)."""
    L = lib()
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    n_rows, wpr = bits.shape
    ph = _pheno_i8(pheno)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    if scratch is not None and scratch.get("n_rows") == n_rows:
        keep, chi2, p, n_with = scratch["arrays"]
    else:
        keep = np.zeros(n_rows, dtype=np.uint8)
        chi2 = np.zeros(n_rows, dtype=np.float64)
        p = np.zeros(n_rows, dtype=np.float64)
        n_with = np.zeros(n_rows, dtype=np.int32)
        if scratch is not None:
            scratch["n_rows"], scratch["arrays"] = n_rows, (keep, chi2, p, n_with)
    args = (bits.ctypes.data, n_rows, wpr, ph.ctypes.data, w.ctypes.data, int(n_samples), int(min_samples),
            int(max_samples), float(pvalue_cutoff), int(bool(omit_B)), int(n_kmers), keep.ctypes.data,
            chi2.ctypes.data, p.ctypes.data, n_with.ctypes.data)
    if n_threads > 1:
        if L.orc_chi2_scan_mt(*args, int(n_threads)) < 0:
            raise MemoryError
    else:
        L.orc_chi2_scan(*args)
    return {"keep": keep.astype(bool), "stat": chi2, "p": p, "n_with": n_with}


def _pheno_f64(pheno):
    vals = np.zeros(len(pheno), dtype=np.float64)
    valid = np.zeros(len(pheno), dtype=np.uint8)
    for i, v in enumerate(pheno):
        if v in ("NA", None) or (isinstance(v, float) and np.isnan(v)):
            continue
        vals[i] = float(v)
        valid[i] = 1
    return vals, valid


def ttest_row(presence, pheno, weights, min_samples, max_samples):
    """phenotypes.conduct_t_test up to the p-value (modeling.py:716-736).
    Returns None or (t, p, mean_x, mean_y, n_with)."""
    L = lib()
    pres = np.ascontiguousarray(np.asarray(presence) != 0, dtype=np.uint8)
    vals, valid = _pheno_f64(pheno)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    t, p, mx, my, n = (ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int())
    ok = L.orc_ttest_row(pres.ctypes.data, vals.ctypes.data, valid.ctypes.data, w.ctypes.data, len(pres),
                         int(min_samples), int(max_samples), ctypes.byref(t), ctypes.byref(p), ctypes.byref(mx),
                         ctypes.byref(my), ctypes.byref(n))
    if not ok:
        return None
    return t.value, p.value, mx.value, my.value, n.value


def ttest_scan(bits, pheno, weights, n_samples, min_samples, max_samples, pvalue_cutoff, n_kmers):
    L = lib()
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    n_rows, wpr = bits.shape
    vals, valid = _pheno_f64(pheno)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    keep = np.zeros(n_rows, dtype=np.uint8)
    t = np.zeros(n_rows)
    p = np.zeros(n_rows)
    mx = np.zeros(n_rows)
    my = np.zeros(n_rows)
    n_with = np.zeros(n_rows, dtype=np.int32)
    L.orc_ttest_scan(bits.ctypes.data, n_rows, wpr, vals.ctypes.data, valid.ctypes.data, w.ctypes.data,
                     int(n_samples), int(min_samples), int(max_samples), float(pvalue_cutoff), int(n_kmers),
                     keep.ctypes.data, t.ctypes.data, p.ctypes.data, mx.ctypes.data, my.ctypes.data,
                     n_with.ctypes.data)
    return {"keep": keep.astype(bool), "stat": t, "p": p, "mean_x": mx, "mean_y": my, "n_with": n_with}


def t_two_sided_p(t, df):
    return lib().orc_t_two_sided_p(float(t), float(df))


# --------------------------------------------------------------------------------------------
# row formatting + selection (modeling.py:796, :739, :1112-1145)
# --------------------------------------------------------------------------------------------
def round2(x):
    """round(np.float64, 2) as the reference gets it (numpy semantics: rint(x*100)/100)."""
    return float(np.round(np.float64(x), 2))


def pstring(p):
    """"%.2E" % pvalue (modeling.py:796)."""
    return "%.2E" % p


def select_order(kmers, pstrings):
    """Column order of get_ML_df (modeling.py:1128): ascending lexicographic order of the
    p-value STRINGS; ties (left to numpy's unstable sort by the reference) broken by k-mer
    text ascending -- the parity contract of SURVEY.md section 8(a) row a9."""
    return sorted(range(len(kmers)), key=lambda i: (pstrings[i], kmers[i]))


# --------------------------------------------------------------------------------------------
# prediction path
# --------------------------------------------------------------------------------------------
def count_dict(buf, k, dict_words):
    """gmer_counter -db restatement (prediction.py:72-80): occurrences (both strands, with
    multiplicity) of each canonical dictionary word in the input."""
    L = lib()
    buf = bytes(buf)
    d = np.ascontiguousarray(dict_words, dtype=np.uint64)
    out = np.zeros(len(d), dtype=np.uint32)
    rc = L.orc_count_dict(buf, len(buf), int(k), d.ctypes.data, len(d), out.ctypes.data)
    if rc != 0:
        raise RuntimeError("orc_count_dict failed")
    return out
