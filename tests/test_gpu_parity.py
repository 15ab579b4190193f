"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI (libpsk.so via
ctypes), against the CPU oracle on the same inputs and against the committed golden fixtures.
Bit-exact for words / counts / bit rows / row sets; statistic tolerances are written below."""
import base64
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, load_dataset, read_results_tsv, tokenizer_cases

pytestmark = pytest.mark.gpu

CHI2_RTOL = 1e-12   # north_star asks 1e-6 relative; unit-weight rows are evaluated in the same order
WEIGHTED_RTOL = 1e-9  # weighted Welch moments are accumulated in a different order than the reference's
T_RTOL = 1e-8


@pytest.fixture(scope="module")
def ctx():
    from phenotypeseeker_amd.engine import PskContext
    c = PskContext(0)
    yield c
    c.close()


def test_tokenizer_cases_bit_exact(ctx, oracle):
    for data, k, ref in tokenizer_cases():
        ctx.begin(k, 1)
        nu, nt = ctx.count_kmers(0, data)
        w, f = ctx.get_list(0, nu)
        if ref is None:
            assert nu == 0
        else:
            assert oracle.list_bytes(k, w, f) == ref, (data[:60], k)


def test_tokenizer_cases_through_the_batch_path(ctx, oracle):
    """The same crafted / fuzzed inputs through psk_count_kmers_batch: the pipelined path sizes its launches by
    the window count the host framing produced and fails loudly if the GPU counts differently, so every case
    also checks that the two tokeniser halves agree."""
    by_k = {}
    for data, k, ref in tokenizer_cases():
        by_k.setdefault(k, []).append((data, ref))
    for k, cases in sorted(by_k.items()):
        ctx.begin(k, len(cases))
        nu, nt = ctx.count_kmers_batch(0, [c[0] for c in cases], 3)
        for i, (data, ref) in enumerate(cases):
            w, f = ctx.get_list(i, nu[i])
            if ref is None:
                assert nu[i] == 0
            else:
                assert oracle.list_bytes(k, w, f) == ref, (data[:60], k)
            assert int(f.astype(np.uint64).sum()) == nt[i]


@pytest.mark.parametrize("k", [5, 13, 16, 21, 31, 32])
def test_megabase_genome_list_bit_exact(ctx, oracle, k):
    from phenotypeseeker_amd.synth import GenomeSet
    gs = GenomeSet(2, 1_000_003, seed=5 + k)
    name, fa = gs.sample(1)
    # sprinkle window breaks and lower case
    b = bytearray(fa)
    rng = np.random.default_rng(k)
    for pos in rng.integers(20, len(b), 300):
        if b[pos] != 10:
            b[pos] = ord("N") if pos % 3 else ord("a")
    fa = bytes(b)
    ow, of, ont = oracle.count_kmers(fa, k)
    ctx.begin(k, 1)
    nu, nt = ctx.count_kmers(0, fa)
    assert (nu, nt) == (len(ow), ont)
    w, f = ctx.get_list(0, nu)
    assert np.array_equal(w, ow)
    assert np.array_equal(f, of)


def test_slab_sharding_concatenates_to_full_list(ctx, oracle):
    from phenotypeseeker_amd.synth import GenomeSet
    k = 13
    name, fa = GenomeSet(1, 200_000, seed=3).sample(0)
    ow, of, _ = oracle.count_kmers(fa, k)
    space = 1 << (2 * k)
    parts_w, parts_f = [], []
    edges = [0, space // 7, space // 3, space // 2 + 12345, 0]
    for lo, hi in zip(edges[:-1], edges[1:]):
        ctx.begin(k, 1, lo, hi)
        nu, nt = ctx.count_kmers(0, fa)
        w, f = ctx.get_list(0, nu)
        if len(w):
            assert w.min() >= lo and (hi == 0 or w.max() < hi)
        parts_w.append(w)
        parts_f.append(f)
    assert np.array_equal(np.concatenate(parts_w), ow)
    assert np.array_equal(np.concatenate(parts_f), of)


@pytest.mark.parametrize("tag", ["ds_omitB", "ds_bonf", "ds_k21"])       # (ds_k21, r05: the reference run with `-l 21` -- 64-bit words)
def test_dataset_union_and_presence(ctx, oracle, tag):
    ds = load_dataset(tag)
    k, names = ds["meta"]["k"], ds["names"]
    ctx.begin(k, len(names))
    lists = []
    for i, n in enumerate(names):
        nu, nt = ctx.count_kmers(i, ds["files"][n])
        m = ds["meta"]["lists"][n]
        assert (nu, nt) == (m["n_unique"], m["n_total"])
        w, f = ctx.get_list(i, nu)
        assert hashlib.sha256(oracle.list_bytes(k, w, f)).hexdigest() == m["sha256"]
        lists.append(w)
    M = ctx.build_presence()
    assert M == ds["meta"]["n_union"]
    uw = ctx.get_union()
    assert np.array_equal(uw, np.load(os.path.join(ds["dir"], "union_words.npy")))
    bits = ctx.get_rows(np.arange(M, dtype=np.uint64))
    ref_bits = oracle.presence_bits(lists, uw, wpr=bits.shape[1])
    assert np.array_equal(bits, ref_bits)
    # glistquery -l mapping of one sample = per-word count lookup
    ms = names.index(ds["meta"]["mapped_sample"])
    counts = ctx.lookup_counts(ms, uw)
    txt = "".join("%s\t%d\n" % (oracle.word_to_kmer(w, k), c) for w, c in zip(uw, counts))
    assert hashlib.sha256(txt.encode()).hexdigest() == ds["meta"]["mapped_sha256"]


@pytest.mark.parametrize("tag,omit_B", [("ds_omitB", True), ("ds_bonf", False), ("ds_k21", False)])
def test_chi2_results_match_reference_tsv(ctx, oracle, tag, omit_B):
    ds = load_dataset(tag)
    k, names = ds["meta"]["k"], ds["names"]
    n = len(names)
    ctx.begin(k, n)
    for i, nm in enumerate(names):
        ctx.count_kmers(i, ds["files"][nm])
    M = ctx.build_presence()
    ph = np.array([-1 if p == "NA" else p for p in ds["pheno"]], dtype=np.int8)
    npass = ctx.chi2_scan(ph, None, 2, n - 2, 0.05, omit_B, M)
    res = ctx.get_results(npass)
    header, ref = read_results_tsv(os.path.join(ds["dir"], "chi2_results_Pheno.tsv"))
    assert npass == len(ref)
    rows = ctx.get_rows(res["row"])
    got = {}
    for j in range(npass):
        pres = [(int(rows[j, i >> 6]) >> (i & 63)) & 1 for i in range(n)]
        with_names = [names[i] for i in range(n) if pres[i] and ds["pheno"][i] != "NA"]
        got[oracle.word_to_kmer(res["word"][j], k)] = (repr(oracle.round2(res["stat"][j])), oracle.pstring(res["p"][j]),
                                                         str(int(res["n_with"][j])), " ".join(["|"] + with_names))
    assert set(got) == {r[0] for r in ref}
    for kmer, stat, p, nw, nmz in ref:
        assert got[kmer] == (stat, p, nw, nmz), kmer


def _random_matrix(rng, m, n, wpr):
    """rows with varied densities, incl. all-zero / all-one rows"""
    dens = rng.random(m) ** 2
    dens[: m // 10] = 0.0
    dens[m // 10: m // 5] = 1.0
    pres = rng.random((m, n)) < dens[:, None]
    bits = np.zeros((m, wpr), dtype=np.uint64)
    for i in range(n):
        bits[:, i >> 6] |= pres[:, i].astype(np.uint64) << np.uint64(i & 63)
    return bits


@pytest.mark.parametrize("n", [3, 30, 64, 65, 100, 128, 150, 256, 384, 1000, 1024, 2048, 9000])
@pytest.mark.parametrize("weighted", [False, True])
def test_chi2_scan_vs_oracle(ctx, oracle, n, weighted):
    from phenotypeseeker_amd.engine import words_per_row
    rng = np.random.default_rng(1000 + n + int(weighted))
    m = 6000 if n <= 2048 else 1500
    wpr = words_per_row(n)
    bits = _random_matrix(rng, m, n, wpr)
    pheno = rng.integers(0, 2, n).astype(object)
    pheno[rng.random(n) < 0.07] = "NA"
    # associate some rows with the phenotype so that small p-values occur
    p1 = np.array([p == 1 for p in pheno])
    for r in range(0, m, 7):
        row = np.zeros(n, dtype=bool)
        row[p1] = rng.random(p1.sum()) < 0.9
        row[~p1] = rng.random((~p1).sum()) < 0.1
        bits[r] = 0
        for i in np.nonzero(row)[0]:
            bits[r, i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    weights = np.round(rng.uniform(0.05, 3.0, n), 6) if weighted else np.ones(n)
    ph8 = np.array([-1 if p == "NA" else p for p in pheno], dtype=np.int8)
    for (mn, mx, cut, omit_B, nk) in [(2, n - 2, 0.05, True, m), (2, n - 2, 0.05, False, m), (1, n, 1.5, True, 10),
                                      (3, max(n // 2, 3), 1e-3, True, 1000)]:
        ref = oracle.chi2_scan(bits, list(pheno), weights, n, mn, mx, cut, omit_B, nk)
        ctx.set_presence(bits, n)
        npass = ctx.chi2_scan(ph8, weights if weighted else None, mn, mx, cut, omit_B, nk)
        res = ctx.get_results(npass)
        keep = np.nonzero(ref["keep"])[0]
        assert np.array_equal(res["row"], keep.astype(np.uint64))
        assert np.array_equal(res["n_with"], ref["n_with"][keep])
        if weighted:
            # the kept rows' 2 x 2 tables are summed in the reference's sample order: the statistic is the oracle's bit for bit
            assert np.array_equal(res["stat"], ref["stat"][keep])
        assert np.allclose(res["stat"], ref["stat"][keep], rtol=CHI2_RTOL, atol=0)
        assert np.allclose(res["p"], ref["p"][keep], rtol=1e-12, atol=0)
        # string-level identity of what the reference prints
        assert [oracle.pstring(x) for x in res["p"]] == [oracle.pstring(x) for x in ref["p"][keep]]
        assert [oracle.round2(x) for x in res["stat"]] == [oracle.round2(x) for x in ref["stat"][keep]]


@pytest.mark.parametrize("n,m", [(3, 1), (5, 2), (30, 4097), (31, 127), (64, 6001), (64, 128 * 4 * 3 + 1), (33, 70_001)])
def test_eight_byte_rows_every_scan_form(ctx, oracle, n, m):
    """r04: up to 64 samples a row of the matrix is ONE u64 and a lane's 16-byte load holds two rows (the scans' G = 0
    instantiations).  Odd row counts (the last load is half a load), a single row, row counts around a whole number of wave
    steps; unit-weight chi2 in both of its forms, weighted chi2 and both Welch scans through every table form."""
    from phenotypeseeker_amd.engine import words_per_row
    assert words_per_row(n) == 1
    rng = np.random.default_rng(n * 1000 + m)
    bits = _random_matrix(rng, m, n, 1)
    base = rng.normal(0.0, 1.0, n)
    for r in range(0, m, 3):
        row = base + rng.normal(0, rng.uniform(0.3, 2.0), n) > rng.uniform(-0.5, 0.8)
        bits[r, 0] = np.uint64(sum(1 << int(i) for i in np.nonzero(row)[0]))
    bits[m - 1, 0] = np.uint64(sum(1 << int(i) for i in np.nonzero(base > 0)[0]))   # the last row is a hit
    ctx.set_presence(bits, n)
    assert ctx.presence_shape()[1] == 1
    assert np.array_equal(ctx.get_rows(np.arange(m, dtype=np.uint64)), bits)
    valid = rng.random(n) > 0.05
    valid[:2] = True
    ph01 = [(int(b > 0) if ok else "NA") for b, ok in zip(base, valid)]
    ph8 = np.array([(-1 if p == "NA" else p) for p in ph01], dtype=np.int8)
    vals = np.round(3.0 + 1.5 * base, 4)
    pheno = [float(v) if ok else "NA" for v, ok in zip(vals, valid)]
    weights = np.round(rng.uniform(0.2, 3.0, n), 6)
    saved = {k_: os.environ.get(k_) for k_ in ("PSK_CHI2_MODE", "PSK_NO_LUT", "PSK_LUT_F64")}
    try:
        for env in ({}, {"PSK_CHI2_MODE": "0"}, {"PSK_CHI2_MODE": "2"}, {"PSK_NO_LUT": "1"}, {"PSK_LUT_F64": "1"}):
            for k_ in saved:
                os.environ.pop(k_, None)
            os.environ.update(env)
            for w in (None, weights):
                for cut, omit, nk in ((0.05, True, m), (0.05, False, m), (1.5, True, 10)):
                    ref = oracle.chi2_scan(bits, ph01, w if w is not None else np.ones(n), n, 1, n, cut, omit, nk)
                    npass = ctx.chi2_scan(ph8, w, 1, n, cut, omit, nk)
                    res = ctx.get_results(npass)
                    keep = np.nonzero(ref["keep"])[0]
                    assert np.array_equal(res["row"], keep.astype(np.uint64)), (env, cut, omit)
                    assert np.array_equal(res["n_with"], ref["n_with"][keep])
                    if w is not None:
                        assert np.array_equal(res["stat"], ref["stat"][keep])
                    assert np.allclose(res["stat"], ref["stat"][keep], rtol=CHI2_RTOL, atol=0)
                    assert [oracle.pstring(x) for x in res["p"]] == [oracle.pstring(x) for x in ref["p"][keep]]
                if n < 6:
                    continue
                for cut, nk in ((0.05, 1), (0.9, 1), (0.05, m)):
                    ref = oracle.ttest_scan(bits, pheno, w if w is not None else np.ones(n), n, 2, n - 2, cut, nk)
                    npass = ctx.ttest_scan(vals, valid, w, 2, n - 2, cut, nk)
                    res = ctx.get_results(npass)
                    keep = np.nonzero(ref["keep"])[0]
                    assert np.array_equal(res["row"], keep.astype(np.uint64)), (env, cut, nk)
                    assert np.array_equal(res["stat"], ref["stat"][keep])
                    assert np.array_equal(res["mean_x"], ref["mean_x"][keep]) and np.array_equal(res["mean_y"], ref["mean_y"][keep])
    finally:
        for k_, v in saved.items():
            os.environ.pop(k_, None)
            if v is not None:
                os.environ[k_] = v


@pytest.mark.parametrize("n", [12, 64, 100, 256, 1024, 2048, 9000])
@pytest.mark.parametrize("weighted", [False, True])
def test_ttest_scan_vs_oracle(ctx, oracle, n, weighted):
    from phenotypeseeker_amd.engine import words_per_row
    rng = np.random.default_rng(77 + n + int(weighted))
    m = 3000
    wpr = words_per_row(n)
    bits = _random_matrix(rng, m, n, wpr)
    vals = np.round(rng.normal(3.0, 1.5, n), 4)
    valid = rng.random(n) > 0.06
    for r in range(0, m, 5):  # planted associations
        row = vals + rng.normal(0, 0.7, n) > 3.4
        bits[r] = 0
        for i in np.nonzero(row)[0]:
            bits[r, i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    weights = np.round(rng.uniform(0.2, 3.0, n), 6) if weighted else np.ones(n)
    pheno = [float(v) if ok else "NA" for v, ok in zip(vals, valid)]
    for (mn, mx, cut, nk) in [(2, n - 2, 0.05, 100), (2, n - 2, 0.9, 1), (3, n // 2, 0.05, m)]:
        ref = oracle.ttest_scan(bits, pheno, weights, n, mn, mx, cut, nk)
        ctx.set_presence(bits, n)
        npass = ctx.ttest_scan(vals, valid, weights if weighted else None, mn, mx, cut, nk)
        res = ctx.get_results(npass)
        keep = np.nonzero(ref["keep"])[0]
        # the candidates' moments are summed again in the reference's (= the oracle's) sample order by the second
        # kernel: the SAME rows, t and the two means bit for bit, and every printed field string-identical (r02 allowed
        # two rows to flip at the cut and compared values at 1e-8)
        assert np.array_equal(res["row"], keep.astype(np.uint64))
        assert np.array_equal(res["stat"], ref["stat"][keep])
        assert np.array_equal(res["mean_x"], ref["mean_x"][keep]) and np.array_equal(res["mean_y"], ref["mean_y"][keep])
        assert np.array_equal(res["n_with"], ref["n_with"][keep])
        assert np.allclose(res["p"], ref["p"][keep], rtol=1e-10, atol=1e-300)   # lgamma / exp / log of two libms
        differing = sum(("%.2E" % a) != ("%.2E" % b) for a, b in zip(res["p"], ref["p"][keep]))
        differing += sum(oracle.round2(a) != oracle.round2(b) for a, b in zip(res["stat"], ref["stat"][keep]))
        assert differing == 0


@pytest.mark.parametrize("n,offset,scale,wspread", [(1024, 0.0, 1.0, 0.0), (1024, 1e6, 1e-3, 2.5), (700, -3e3, 1e-12, 1.0), (384, 5.0, 1e9, 3.0),
                                                    (2048, 0.0, 1.0, 0.5)])
def test_moment_scans_miss_no_candidate_on_badly_scaled_data(ctx, oracle, n, offset, scale, wspread):
    """r03: the streaming kernels of the Welch and weighted chi2 scans pick their candidates from f32 sums over six-bit
    tables and an upper bound of the statistic built from rounding-error bounds; the second kernel decides exactly.
    So the kept rows must equal the oracle's whatever the phenotype's unit (offsets of 1e6 against a spread of 1e-3,
    values of 1e-12 or 1e9), with weights over six orders of magnitude, and with cuts that put many rows near the
    threshold.  (2,048 samples: the six-bit table does not fit the LDS, the f64 nibble table serves.)"""
    from phenotypeseeker_amd.engine import words_per_row
    rng = np.random.default_rng(int(n + wspread * 10))
    m = 6000
    wpr = words_per_row(n)
    bits = _random_matrix(rng, m, n, wpr)
    base = rng.normal(0.0, 1.0, n)
    vals = offset + scale * base
    valid = rng.random(n) > 0.03
    for r in range(0, m, 3):  # planted associations of every strength, so that rows sit on both sides of every cut
        row = base + rng.normal(0, rng.uniform(0.3, 3.0), n) > rng.uniform(-0.5, 1.0)
        bits[r] = 0
        for i in np.nonzero(row)[0]:
            bits[r, i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    weights = np.exp(rng.normal(0.0, wspread, n)) if wspread else np.ones(n)
    pheno = [float(v) if ok else "NA" for v, ok in zip(vals, valid)]
    ctx.set_presence(bits, n)
    for cut, nk in ((0.05, 1), (0.05, m), (1e-6, m), (0.9, 1)):
        ref = oracle.ttest_scan(bits, pheno, weights, n, 2, n - 2, cut, nk)
        npass = ctx.ttest_scan(vals, valid, weights if wspread else None, 2, n - 2, cut, nk)
        res = ctx.get_results(npass)
        keep = np.nonzero(ref["keep"])[0]
        assert np.array_equal(res["row"], keep.astype(np.uint64)), (cut, nk, npass, len(keep))
        assert np.array_equal(res["stat"], ref["stat"][keep])
    ph01 = [(int(b > 0) if ok else "NA") for b, ok in zip(base, valid)]
    ph8 = np.array([(-1 if p == "NA" else p) for p in ph01], dtype=np.int8)
    for cut, omit, nk in ((0.05, True, m), (0.05, False, m), (1e-9, False, m)):
        ref = oracle.chi2_scan(bits, ph01, weights, n, 2, n - 2, cut, omit, nk)
        npass = ctx.chi2_scan(ph8, weights if wspread else None, 2, n - 2, cut, omit, nk)
        res = ctx.get_results(npass)
        keep = np.nonzero(ref["keep"])[0]
        assert np.array_equal(res["row"], keep.astype(np.uint64)), (cut, omit)
        assert np.array_equal(res["stat"], ref["stat"][keep])


def test_count_dict_matches_gmer_counter(ctx, oracle):
    with open(os.path.join(GOLDEN, "gmer_counter.json")) as f:
        d = json.load(f)
    k = d["k"]
    words = [oracle.canonical_word(oracle.kmer_to_word(km), k) for km in d["kmers"]]
    for c in d["cases"]:
        fa = gzip.decompress(base64.b64decode(c["fasta_gz_b64"]))
        counts = ctx.count_dict(fa, k, words)
        body = [l.split("\t") for l in c["output"].splitlines()[2:]]
        assert [int(b[2]) for b in body] == counts.tolist()


def test_matrix_cannot_be_replaced_under_a_running_scan(ctx):
    """the calls that replace or compact the matrix refuse while a scan is in flight."""
    from phenotypeseeker_amd._lib import PskError
    n, m = 64, 200_000
    ctx.synth_presence(m, n, seed=3)
    ph = (np.arange(n) % 2).astype(np.int8)
    ctx.chi2_scan_begin(ph, None, 2, n - 2, 0.05, False, m)
    for call in (lambda: ctx.synth_presence(m, n, seed=4), lambda: ctx.build_presence(),
                 lambda: ctx.set_presence(np.zeros((4, 2), np.uint64), n), lambda: ctx.intersect_db(np.arange(5, dtype=np.uint64))):
        with pytest.raises(PskError):
            call()
    a = ctx.scan_end()
    assert a == ctx.chi2_scan(ph, None, 2, n - 2, 0.05, False, m)
    ctx.synth_presence(m, n, seed=4)        # fine again


def test_count_dict_with_large_dictionaries_and_in_batches(ctx, oracle, tmp_path):
    """`--n_kmers 0` / `--n_kmers 5000` models have more than 2048 k-mers (the LDS table's limit): beyond
    it the table is probed in global memory.  And the batched forms (what `prediction` calls: all samples in one go,
    from memory or from files) equal one call per sample; counts are occurrences on both strands (the oracle's list),
    duplicates of one canonical word share their count, absent words are 0."""
    from phenotypeseeker_amd.synth import GenomeSet
    rng = np.random.default_rng(5)
    for k in (13, 21):
        gs = GenomeSet(5, 80_000, seed=k, gene_len=300)
        datas = [gs.sample(i)[1] for i in range(5)] + [b"", b">x\nACGT\n"]
        lists = [oracle.count_kmers(d, k)[:2] for d in datas]
        present = np.unique(np.concatenate([lists[0][0][::11], lists[1][0][::13]]))
        absent = rng.integers(0, 1 << (2 * k), 3000).astype(np.uint64)
        for n_dict in (1, 700, 2048, 2049, 6000, 40_000):
            words = np.concatenate([present, absent])[:n_dict]
            words = rng.permutation(np.concatenate([words, words[:5]]))          # with duplicates
            want = np.zeros((len(datas), len(words)), np.uint32)
            for i, (w, f) in enumerate(lists):
                pos = np.searchsorted(w, words)
                hit = (pos < len(w)) & (w[np.minimum(pos, max(len(w) - 1, 0))] == words) if len(w) else np.zeros(len(words), bool)
                want[i, hit] = f[pos[hit]]
            one = np.stack([ctx.count_dict(d, k, words) for d in datas])
            assert np.array_equal(one, want), (k, n_dict)
            assert np.array_equal(ctx.count_dict_batch(datas, k, words, 3), want)
            paths = []
            for i, d in enumerate(datas):
                paths.append(os.path.join(tmp_path, "s%d.fa" % i))
                with open(paths[-1], "wb") as f:
                    f.write(d)
            assert np.array_equal(ctx.count_dict_files(paths, k, words, 2), want)


def test_full_size_fastq_sample_properties(ctx, oracle):
    """BASELINE config-5 sized sample (2 M reads x 150 bp, ~0.63 GB of FASTQ, 276 M windows): the window
    count is reads x (150 - k + 1), the counts add up to it, the words come back strictly ascending, and
    a 20000-read prefix equals the oracle word for word."""
    from phenotypeseeker_amd.synth import GenomeSet, fastq_reads
    reads, k = 2_000_000, 13
    gs = GenomeSet(2, 5_000_000, seed=99)
    data = fastq_reads(gs.codes(0), reads, 150, seed=[5, 0])
    ctx.begin(k, 1)
    nu, nt = ctx.count_kmers(0, data)
    assert nt == reads * (150 - k + 1)
    words, freqs = ctx.get_list(0, nu)
    assert int(freqs.astype(np.uint64).sum()) == nt and np.all(words[1:] > words[:-1])
    cut = 0
    for _ in range(4 * 20000):
        cut = data.index(b"\n", cut) + 1
    ow, of = oracle.count_kmers(data[:cut], k)[:2]
    ctx.begin(k, 1)
    nu2, _ = ctx.count_kmers(0, data[:cut])
    w2, f2 = ctx.get_list(0, nu2)
    assert np.array_equal(w2, ow) and np.array_equal(f2, of)


def test_full_size_scan_properties(ctx):
    """BASELINE config-2 sized matrix (2^25 rows x 256 samples): size-independent properties --
    the scan is idempotent, its survivors come back in ascending row order, flipping the
    phenotype labels leaves chi2 unchanged, and the NA-everything phenotype yields nothing."""
    n, m = 256, 1 << 25
    ctx.synth_presence(m, n, seed=9)
    ph = (np.arange(n) % 2).astype(np.int8)
    a = ctx.chi2_scan(ph, None, 2, n - 2, 0.05, False, m)
    ra = ctx.get_results(a)
    b = ctx.chi2_scan(ph, None, 2, n - 2, 0.05, False, m)
    rb = ctx.get_results(b)
    assert a == b and a > 0
    assert np.array_equal(ra["row"], rb["row"]) and np.array_equal(ra["stat"], rb["stat"])
    assert np.all(np.diff(ra["row"].astype(np.int64)) > 0)
    c = ctx.chi2_scan(1 - ph, None, 2, n - 2, 0.05, False, m)
    rc = ctx.get_results(c)
    assert np.array_equal(ra["row"], rc["row"]) and np.allclose(ra["stat"], rc["stat"], rtol=1e-12)
    assert ctx.chi2_scan(np.full(n, -1, np.int8), None, 2, n - 2, 0.05, True, m) == 0
    # a sample of survivors re-checked on the CPU by the closed 2x2 formula
    rows = ctx.get_rows(ra["row"][:200])
    m1 = np.zeros(rows.shape[1], dtype=np.uint64)
    for i in np.nonzero(ph == 1)[0]:
        m1[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    for j in range(len(rows)):
        aa = sum(bin(int(x & y)).count("1") for x, y in zip(rows[j], m1))
        tot = sum(bin(int(x)).count("1") for x in rows[j])
        cc = tot - aa
        A, B, C, D = aa, 128 - aa, cc, 128 - cc
        chi = 256.0 * (A * D - B * C) ** 2 / ((A + B) * (C + D) * (A + C) * (B + D))
        assert ra["stat"][j] == pytest.approx(chi, rel=1e-9)


def test_two_scans_in_flight_equal_the_one_call_scans(ctx):
    """psk_chi2_scan_begin twice, then psk_scan_end twice (two result sets): each ended scan's survivors equal the
    one-call scan's, ends come in launch order, unweighted and weighted (per-set staging of masks and weights),
    and the state errors of include/psk.h hold."""
    n, m = 200, 400_000
    ctx.synth_presence(m, n, seed=21)
    rng = np.random.default_rng(5)
    phs = [(np.arange(n) % 2).astype(np.int8), (rng.random(n) < 0.4).astype(np.int8),
           np.where(rng.random(n) < 0.1, -1, (np.arange(n) // 3) % 2).astype(np.int8)]
    for w in (None, rng.uniform(0.2, 3.0, n)):
        want = []
        for ph in phs:
            c = ctx.chi2_scan(ph, w, 2, n - 2, 0.05, False, m)
            want.append((c, ctx.get_results(c)))
        assert want[0][0] > 0
        ctx.chi2_scan_begin(phs[0], w, 2, n - 2, 0.05, False, m)
        ctx.chi2_scan_begin(phs[1], w, 2, n - 2, 0.05, False, m)
        with pytest.raises(RuntimeError):
            ctx.chi2_scan_begin(phs[2], w, 2, n - 2, 0.05, False, m)       # a third
        with pytest.raises(RuntimeError):
            ctx.chi2_scan(phs[2], w, 2, n - 2, 0.05, False, m)             # one-call scan while scans are in flight
        for i in range(3):
            c = ctx.scan_end()
            got = ctx.get_results(c)
            assert c == want[i][0]
            for key in ("row", "word", "stat", "p", "n_with"):
                assert np.array_equal(got[key], want[i][1][key]), (i, key)
            if i == 0:
                ctx.chi2_scan_begin(phs[2], w, 2, n - 2, 0.05, False, m)   # takes the set just read
                with pytest.raises(RuntimeError):
                    ctx.get_results(c)                                      # ... whose results are gone
        assert ctx.scan_end() == want[2][0]                                 # none in flight: the last count again
    # the two forms of the unweighted kernel give the same rows: the first scan of a matrix with the Bonferroni rule runs
    # the in-line form, it keeps > 0.1 % of these rows, so the repeat runs the queued form (pick_chi2_mode)
    ctx.synth_presence(m, n, seed=22)
    c1 = ctx.chi2_scan(phs[0], None, 2, n - 2, 0.05, False, m)
    r1 = ctx.get_results(c1)
    assert c1 > m // 1000
    c2 = ctx.chi2_scan(phs[0], None, 2, n - 2, 0.05, False, m)
    r2 = ctx.get_results(c2)
    assert c1 == c2 and all(np.array_equal(r1[key], r2[key]) for key in ("row", "stat", "p", "n_with"))
    # psk_begin with scans in flight: waits for them and starts from a clean scan state
    ctx.chi2_scan_begin(phs[0], None, 2, n - 2, 0.05, False, m)
    ctx.chi2_scan_begin(phs[1], None, 2, n - 2, 0.05, False, m)
    ctx.begin(13, n)
    with pytest.raises(RuntimeError):
        ctx.get_results(1)
    ctx.synth_presence(1000, n, seed=3)
    ctx.chi2_scan_begin(phs[0], None, 2, n - 2, 0.05, False, 1000)
    ctx.chi2_scan_begin(phs[0], None, 2, n - 2, 0.05, False, 1000)      # two again: the old ones no longer count
    assert ctx.scan_end() == ctx.scan_end()


def _pattern_sums(X, w):
    """coefficient mass per distinct column pattern (identical columns share it arbitrarily)"""
    pats = {}
    for j in range(X.shape[1]):
        pats.setdefault(X[:, j].tobytes(), 0.0)
        pats[X[:, j].tobytes()] += w[j]
    return pats


def test_l1_logreg_solver_reaches_liblinear_optimum(ctx):
    """a10 contract at north_star's tolerance: coefficients (per distinct column pattern where columns repeat), intercept
    and the linear predictor on the training rows within 1e-6 RELATIVE of the exact optimum of liblinear's objective.
    The comparator is the arbiter of oracle_model (active-set Newton, KKT residual < 1e-12 -- better than either
    approximate solver); the liblinear fixture is held to it on the CPU side (test_oracle_golden).  r02 compared at 1e-3
    because it ran the HIP solver at tol = 1e-8: at tol = 1e-12 the solver is 1e-8 from the optimum, the converged
    liblinear fixture 6e-7 (tools/a10_probe.py) -- the stopping tolerance was the limiting side, not the solver."""
    from oracle import oracle_model as OM
    z = np.load(os.path.join(GOLDEN, "model_kat.npz"))
    for tag in ("1", "2"):
        X, y = z["X" + tag], z["y" + tag]
        Cs = [float(c) for c in z["Cs"]]
        coef, icpt, iters = ctx.logreg_l1_fit(X, y, np.zeros(len(y), np.int32), Cs, [-1] * len(Cs), tol=1e-12,
                                              max_iter=5000)
        for ci, C in enumerate(Cs):
            assert iters[ci] < 5000, (tag, C)
            a = OM.logreg_l1_arbiter(X, y, C, z["logreg_coef" + tag][ci], float(z["logreg_icpt" + tag][ci]))
            assert a["kkt"] < 1e-12 and not a["rank_deficient"]
            sums = np.zeros(len(a["w_groups"]))
            np.add.at(sums, a["group"], coef[ci])                  # design 2 has no repeated column: the coefficients
            assert np.array_equal(sums == 0, a["w_groups"] == 0), (tag, C)      # the same support, exact zeros off it
            assert np.allclose(sums, a["w_groups"], rtol=1e-6, atol=0), (tag, C)
            assert icpt[ci] == pytest.approx(a["b"], rel=1e-6, abs=0)
            assert np.allclose(X @ coef[ci] + icpt[ci], a["linpred"], rtol=1e-6, atol=1e-12), (tag, C)
            obj = OM.logreg_l1_objective(X, y, coef[ci], icpt[ci], C)
            assert obj == pytest.approx(a["objective"], rel=1e-12)
            assert obj == pytest.approx(float(z["logreg_obj" + tag][ci]), rel=1e-9)          # and the liblinear fixture's


def test_l1_logreg_converges_on_near_duplicate_columns(ctx):
    """The (grid x fold) problem of a weighted 256-sample run: 138 distinct columns that differ from the gene
    pattern in a few samples.  Every fit must stop by liblinear's rule well before max_iter (a coefficient
    stuck at ~1e-17 once kept five of them spinning), and at the stop the rule must hold for the true
    gradient."""
    d = np.load(os.path.join(GOLDEN, "fit_near_duplicates.npz"))
    X, y, fold, fp, ff = d["X"], d["y"], d["fold"], d["fit_param"], d["fit_fold"]
    coef, icpt, iters = ctx.logreg_l1_fit(X, y, fold, fp, ff, tol=1e-4, max_iter=1000)
    assert iters.max() < 100, iters.tolist()
    ypm = 2.0 * y - 1.0
    for j in range(0, len(fp), 7):
        tr = fold != ff[j]
        A = np.hstack([X[tr], np.ones((tr.sum(), 1))])
        yt = ypm[tr]

        def viol(th):
            g = -fp[j] * (A.T @ (yt / (1.0 + np.exp(yt * (A @ th)))))
            return np.where(th > 0, np.abs(g + 1), np.where(th < 0, np.abs(g - 1),
                                                             np.maximum(0, np.maximum(-(g + 1), g - 1)))).sum()
        th = np.append(coef[j], icpt[j])
        eps = 1e-4 * max(min((yt > 0).sum(), (yt < 0).sum()), 1) / tr.sum()
        assert viol(th) <= 1.5 * eps * viol(np.zeros_like(th)) + 1e-9, (j, fp[j], iters[j])


def test_l1_logreg_mid_size_objectives_match_liblinear(ctx):
    """256 samples x 100 random columns (two feature slots per lane in the covariance-form QP) and x 150
    near-duplicate columns (three slots, packed Gram block, CG accelerator): the objective reached at a tight
    tolerance equals tightly converged liblinear's (tests/golden/model_mid_kat.npz), and the default
    tolerance stays within a percent of it."""
    from oracle import oracle_model as OM
    z = np.load(os.path.join(GOLDEN, "model_mid_kat.npz"))
    n = int(z["n"])
    Cs = [float(c) for c in z["Cs"]]
    for tag in ("a", "b"):
        X = np.unpackbits(z["X" + tag], axis=0)[:n].astype(np.float64)
        y = z["y" + tag]
        zero = np.zeros(n, np.int32)
        coef, icpt, iters = ctx.logreg_l1_fit(X, y, zero, Cs, [-1] * len(Cs), tol=1e-7, max_iter=300)
        for j, C in enumerate(Cs):
            obj = OM.logreg_l1_objective(X, y, coef[j], icpt[j], C)
            # the near-duplicate design is ill-conditioned: at tol 1e-7 the objective is within 1e-5 (2e-7 at 1e-9,
            # which takes 40 s)
            assert obj == pytest.approx(float(z["obj_" + tag][j]), rel=2e-6 if tag == "a" else 2e-5), (tag, C, int(iters[j]))
        coef, icpt, iters = ctx.logreg_l1_fit(X, y, zero, Cs, [-1] * len(Cs), tol=1e-4, max_iter=1000)
        assert iters.max() < 100
        for j, C in enumerate(Cs):
            obj = OM.logreg_l1_objective(X, y, coef[j], icpt[j], C)
            # (the near-duplicate design at C = 100 stops 0.4-0.7 % above the optimum, depending on the rounding of the
            # Gram block's sums: 17 or 18 Newton steps)
            assert float(z["obj_" + tag][j]) * (1 - 1e-9) <= obj <= float(z["obj_" + tag][j]) * 1.01, (tag, C)


def _l1_stop_rule_holds(X, ypm, fold, fp, ff, coef, icpt, iters, fits, tol=1e-4, slack=1.5):
    """liblinear's stopping rule, for the TRUE gradient at the returned point: ||violation||_1 <= tol min(#pos, #neg) / l x
    the violation at w = 0 (x slack: the solver tests it on its own accumulated quantities)."""
    for j in fits:
        tr = fold != ff[j]
        A = np.hstack([X[tr].astype(np.float64), np.ones((tr.sum(), 1))])
        yt = ypm[tr]

        def viol(th):
            g = -fp[j] * (A.T @ (yt / (1.0 + np.exp(yt * (A @ th)))))
            return np.where(th > 0, np.abs(g + 1), np.where(th < 0, np.abs(g - 1),
                                                             np.maximum(0, np.maximum(-(g + 1), g - 1)))).sum()
        eps = tol * max(min((yt > 0).sum(), (yt < 0).sum()), 1) / tr.sum()
        th = np.append(coef[j], icpt[j])
        assert viol(th) <= slack * eps * viol(np.zeros_like(th)) + 1e-9, (j, fp[j], iters[j])


def _l1_objectives(X, ypm, fold, fp, ff, coef, icpt):
    Z = X.astype(np.float64) @ coef.T + icpt[None, :]
    return np.array([np.abs(coef[j]).sum() + abs(icpt[j]) + fp[j] * np.logaddexp(0.0, -ypm[fold != ff[j]] * Z[fold != ff[j], j]).sum()
                     for j in range(len(fp))])


@pytest.mark.parametrize("n", [40, 130, 700, 2048, 3000, 4096])
def test_l1_logreg_three_forms_of_the_descent_agree(ctx, n, monkeypatch):
    """More distinct columns than the LDS Gram block holds (250 > 192).  Three forms of the same inner solver: the Gram
    matrix in global memory (the default: Q = X'DX on the bf16 matrix cores, a visit divided over four waves), the array
    form on four waves with the samples in registers (PSK_NO_GRAM_GLOBAL=1) and the one-wave LDS array form
    (PSK_NO_CD_REGS=1).  Each is deterministic, all stop by liblinear's rule, and their objectives agree far inside the
    stopping tolerance.  n = 3000 and 4096 run the 33..64-word instance of the kernel."""
    rng = np.random.default_rng(n)
    p = 250
    base = rng.random((n, 12)) < 0.4
    X = (base[:, rng.integers(0, 12, p)] ^ (rng.random((n, p)) < 0.08)).astype(np.float32)
    y = (base[:, 0] ^ (rng.random(n) < 0.15)).astype(np.int32)
    fold = (np.arange(n) % 3).astype(np.int32)
    fp = np.array([0.01, 0.01, 0.1, 1.0, 1.0], np.float64)
    ff = np.array([-1, 0, 1, 2, -1], np.int32)
    runs = {}
    for tag, env in (("gram-global", None), ("four-wave arrays", "PSK_NO_GRAM_GLOBAL"), ("one-wave arrays", "PSK_NO_CD_REGS")):
        if env:
            monkeypatch.setenv(env, "1")
        r = ctx.logreg_l1_fit(X, y, fold, fp, ff, tol=1e-4, max_iter=1000)
        r2 = ctx.logreg_l1_fit(X, y, fold, fp, ff, tol=1e-4, max_iter=1000)
        if env:
            monkeypatch.delenv(env)
        assert all(np.array_equal(u, v) for u, v in zip(r, r2)), tag      # four waves, one answer: deterministic
        assert r[2].max() < 200, (tag, r[2])
        runs[tag] = r
    a = runs["gram-global"]
    assert n < 700 or all((c != 0).sum() > 0 for c in a[0][2:])   # (a few dozen samples may leave the weak fits at zero)
    ypm = 2.0 * y - 1.0
    objs = {}
    for tag, r in runs.items():
        _l1_stop_rule_holds(X, ypm, fold, fp, ff, r[0], r[1], r[2], range(len(fp)))
        objs[tag] = _l1_objectives(X, ypm, fold, fp, ff, r[0], r[1])
    for tag in ("four-wave arrays", "one-wave arrays"):
        assert np.allclose(objs["gram-global"], objs[tag], rtol=1e-5, atol=0), tag


def test_l1_logreg_gram_global_form_reaches_the_same_optimum_at_a_tight_tolerance(ctx, monkeypatch):
    """The Gram matrix of the global form is f32 sums of bf16-split weights: an approximate Hessian, so its Newton steps
    converge linearly near the optimum (more of them than the array form's) -- to the SAME optimum: at tol = 1e-9 the
    objectives agree to 1e-13 and every coefficient to 1e-6 of the largest (measured: 2e-16 and 2e-9 ... 2e-8)."""
    n, p = 700, 250
    rng = np.random.default_rng(n + p)
    base = rng.random((n, 12)) < 0.4
    X = (base[:, rng.integers(0, 12, p)] ^ (rng.random((n, p)) < 0.08)).astype(np.float32)
    y = (base[:, 0] ^ (rng.random(n) < 0.15)).astype(np.int32)
    fold = (np.arange(n) % 3).astype(np.int32)
    fp = np.array([0.1, 1.0, 1.0], np.float64)
    ff = np.array([0, 1, -1], np.int32)
    a = ctx.logreg_l1_fit(X, y, fold, fp, ff, tol=1e-9, max_iter=3000)
    monkeypatch.setenv("PSK_NO_GRAM_GLOBAL", "1")
    b = ctx.logreg_l1_fit(X, y, fold, fp, ff, tol=1e-9, max_iter=3000)
    monkeypatch.delenv("PSK_NO_GRAM_GLOBAL")
    assert a[2].max() < 3000 and b[2].max() < 3000, (a[2], b[2])
    ypm = 2.0 * y - 1.0
    oa, ob = _l1_objectives(X, ypm, fold, fp, ff, a[0], a[1]), _l1_objectives(X, ypm, fold, fp, ff, b[0], b[1])
    assert np.allclose(oa, ob, rtol=1e-13, atol=0), oa / ob - 1
    for j in range(len(fp)):
        scale = max(np.abs(b[0][j]).max(), abs(b[1][j]))
        assert np.abs(a[0][j] - b[0][j]).max() <= 1e-6 * scale and abs(a[1][j] - b[1][j]) <= 1e-6 * scale, j
        assert np.array_equal(a[0][j] != 0, b[0][j] != 0), j      # the same support


def test_l1_logreg_accelerated_descent_ends_where_the_plain_one_does(ctx, monkeypatch):
    """The Gram-global form interrupts a crawling descent for conjugate-gradient steps on the free set (gg_polish).  Any
    point is a valid iterate, so the stopping rule and the optimum are those of the plain descent (PSK_CG_MAX=0): on an
    ill-conditioned design (600 near-duplicate columns of 40 factors, C up to 100) both stop by liblinear's rule; at
    tol = 1e-8 the objectives agree to 1e-8 (measured 3e-10; the coefficients themselves to 1e-4: flat directions), at the
    default tolerance the accelerated fits end at or below the plain ones (measured 0 ... -1.8 %)."""
    n, p = 500, 600
    rng = np.random.default_rng(n + p)
    base = rng.random((n, 40)) < 0.35
    X = (base[:, rng.integers(0, 40, p)] ^ (rng.random((n, p)) < 0.05)).astype(np.float32)
    y = ((base[:, 0] & base[:, 3]) ^ (rng.random(n) < 0.1)).astype(np.int32)
    fold = (np.arange(n) % 5).astype(np.int32)
    fp = np.array([1.0, 10.0, 100.0, 100.0])
    ff = np.array([-1, 2, -1, 3], np.int32)
    ypm = 2.0 * y - 1.0
    for tol, lo, hi in ((1e-4, -5e-2, 1e-3), (1e-8, -1e-8, 1e-8)):
        a = ctx.logreg_l1_fit(X, y, fold, fp, ff, tol=tol, max_iter=3000)
        monkeypatch.setenv("PSK_CG_MAX", "0")
        b = ctx.logreg_l1_fit(X, y, fold, fp, ff, tol=tol, max_iter=3000)
        monkeypatch.delenv("PSK_CG_MAX")
        assert a[2].max() < 3000 and b[2].max() < 3000, (a[2], b[2])
        _l1_stop_rule_holds(X, ypm, fold, fp, ff, a[0], a[1], a[2], range(len(fp)), tol=tol)
        _l1_stop_rule_holds(X, ypm, fold, fp, ff, b[0], b[1], b[2], range(len(fp)), tol=tol)
        rel = _l1_objectives(X, ypm, fold, fp, ff, a[0], a[1]) / _l1_objectives(X, ypm, fold, fp, ff, b[0], b[1]) - 1
        assert rel.min() > lo and rel.max() < hi, (tol, rel)


@pytest.mark.parametrize("n,p,nb,flip,noise", [(1500, 191, 20, 0.08, 0.02), (1100, 255, 40, 0.02, 0.02), (4096, 64, 8, 0.2, 0.3),
                                               (3000, 300, 20, 0.02, 0.02), (2048, 192, 8, 0.02, 0.1)])
def test_l1_logreg_gram_global_form_from_1024_samples_on(ctx, monkeypatch, n, p, nb, flip, noise):
    """From 1,024 samples on, designs of more than 64 distinct columns take the Gram-global form (below that the LDS Gram
    form keeps those of up to 192).  Random 0/1 designs around the borders (64 / 192 / 256 columns, 1,100 ... 4,096
    samples, near-duplicate columns of nb factors, clean to noisy labels), C = 0.01 ... 1000 with and without a held-out
    fold: every fit stops by liblinear's rule for the true gradient and ends at most 1e-3 above the objective of the form
    it replaced (measured: +4e-4 ... -1.6 %; 2-14 x faster)."""
    rng = np.random.default_rng(n + p)
    base = rng.random((n, nb)) < rng.uniform(0.1, 0.5)
    X = (base[:, rng.integers(0, nb, p)] ^ (rng.random((n, p)) < flip)).astype(np.float32)
    y = ((base[:, 0] & base[:, min(3, nb - 1)]) ^ (rng.random(n) < noise)).astype(np.int32)
    fold = (np.arange(n) % 4).astype(np.int32)
    fp = np.array([0.01, 0.1, 1.0, 10.0, 100.0, 1000.0])
    ff = np.array([-1, 0, 1, 2, 3, -1], np.int32)
    ypm = 2.0 * y - 1.0
    a = ctx.logreg_l1_fit(X, y, fold, fp, ff, tol=1e-4, max_iter=1000)
    monkeypatch.setenv("PSK_NO_GRAM_GLOBAL", "1")
    b = ctx.logreg_l1_fit(X, y, fold, fp, ff, tol=1e-4, max_iter=1000)
    monkeypatch.delenv("PSK_NO_GRAM_GLOBAL")
    assert np.isfinite(a[0]).all() and a[2].max() < 1000, a[2]
    _l1_stop_rule_holds(X, ypm, fold, fp, ff, a[0], a[1], a[2], range(len(fp)), tol=1e-4)
    rel = _l1_objectives(X, ypm, fold, fp, ff, a[0], a[1]) / _l1_objectives(X, ypm, fold, fp, ff, b[0], b[1]) - 1
    # C = 1000 at tol = 1e-4 is where liblinear's rule stops both forms far from the optimum and from each other: either may be the
    # lower one by a few per cent, and not the same one every run (r05: +3.4 % once in three runs of the suite, -1.6 % in r04) --
    # the bound is symmetric there; up to C = 100 the new form ends at most 1e-3 above the old
    assert rel[:-1].max() < 1e-3 and rel.max() < 5e-2 and rel.min() > -5e-2, rel


def test_l1_logreg_gram_global_form_on_the_2048_x_907_grid(ctx, monkeypatch):
    """the grid of a 2,048-genome run whose 1,000 selected k-mers have 907 distinct patterns (143 fits; at
    C >= 100 every coefficient ends non-zero, plain coordinate descent needs ~4,700 sweeps of 907 coordinates for the
    slowest fit).  The Gram form in global memory -- with its accelerator, conjugate-gradient steps on the free set -- must
    stop every fit by liblinear's rule, end no higher than the array form (which took 3.3 s) beyond what the rule leaves
    open, and do so well inside a second: 0.33 s measured (0.90 s without the accelerator); the bound leaves room for a
    loaded box."""
    import time
    d = np.load(os.path.join(GOLDEN, "fit2048_907.npz"))
    X = np.unpackbits(d["Xbits"], axis=1)[:, : int(d["p"])].astype(np.float32)
    y, fold, fp, ff = d["y"], d["fold"], d["fit_param"], d["fit_fold"]
    ypm = 2.0 * y - 1.0
    ctx.logreg_l1_fit(X[:, :50], y, fold, fp[:2], ff[:2], 1e-4, 50)       # code objects, buffers
    t0 = time.time()
    coef, icpt, iters = ctx.logreg_l1_fit(X, y, fold, fp, ff, float(d["tol"]), int(d["max_iter"]))
    wall = time.time() - t0
    assert iters.max() < 100, iters.max()
    assert wall < 0.8, wall
    again = ctx.logreg_l1_fit(X, y, fold, fp, ff, float(d["tol"]), int(d["max_iter"]))
    assert np.array_equal(coef, again[0]) and np.array_equal(icpt, again[1])
    _l1_stop_rule_holds(X, ypm, fold, fp, ff, coef, icpt, iters, range(0, len(fp), 9), tol=float(d["tol"]))
    monkeypatch.setenv("PSK_NO_GRAM_GLOBAL", "1")
    ref = ctx.logreg_l1_fit(X, y, fold, fp, ff, float(d["tol"]), int(d["max_iter"]))
    monkeypatch.delenv("PSK_NO_GRAM_GLOBAL")
    o_new, o_ref = _l1_objectives(X, ypm, fold, fp, ff, coef, icpt), _l1_objectives(X, ypm, fold, fp, ff, ref[0], ref[1])
    # both stop by the same rule -- a bound on the violation, not on the objective, which at tol = 1e-4 leaves a few per cent
    # open on these ill-conditioned fits: measured -3.3 % ... +1.7 % fit by fit, the accelerated form 1.8 % LOWER in total
    rel = o_new / o_ref - 1
    assert rel.max() < 3e-2 and rel.min() > -8e-2, (float(rel.max()), float(rel.min()))
    assert -6e-2 < o_new.sum() / o_ref.sum() - 1 < 5e-4


@pytest.mark.parametrize("form", ["gram-global", "four-wave arrays"])
@pytest.mark.parametrize("tag", ["g", "h", "i"])
def test_l1_logreg_large_forms_reach_the_certified_liblinear_optimum(ctx, monkeypatch, tag, form):
    """VERDICT r03 #1 -- a10's contract for the solver forms that carry every large fit.  tests/golden/model_large_kat.npz
    holds, for the 2,048 x 907 design of a 2,048-genome run ('g'), a 1,500 x 300 ('h') and a 300 x 400 ('i') near-duplicate
    design, scikit-learn's liblinear fits at C = 0.01 .. 100 on all samples and on two training folds, and beside each the
    exact optimum of liblinear's objective (the arbiter seeded with liblinear's point; its KKT certificate is re-checked
    from the stored numbers by tests/test_oracle_golden.py -- liblinear itself stops 6e-8 (C = 10) to 4e-4 (C = 100) above
    it in the objective).  The default form (Gram matrix in global memory, bf16-MFMA Hessian, CG accelerator) AND the
    four-wave array form (PSK_NO_GRAM_GLOBAL=1), run to a tight tolerance, must reach: the objective to 1e-8 relative
    (measured <= 3e-12), the linear predictor Xw + b on the training rows, the coefficient sum of every distinct column
    pattern and the intercept to 1e-6 of the largest ('h', 'i': measured <= 3e-8; 'g': see the bar below).
    Tolerances: liblinear's rule is relative to the violation at w = 0, so what it leaves in the coefficients grows with
    C and with the conditioning: 1e-10 leaves 1e-8 on 'h' but 4e-6 on 'i' and 5e-5 on 'g' at C = 100, hence 1e-12 there
    (near the floor of what the line search resolves in doubles a fit ends after three Newton steps inside its rounding noise:
    solver_l1_bits.h).  The array form without the accelerator needs more than 300 Newton steps of thousands of sweeps for 'g' at
    C >= 10 at such a tolerance (minutes per fit): there it is held to the optimum at the reference's tolerance, below."""
    from helpers import large_design
    z = np.load(os.path.join(GOLDEN, "model_large_kat.npz"))
    X, y, fold = large_design(z, tag)
    fp, ff = z["fit_C_" + tag].astype(np.float64), z["fit_held_" + tag].astype(np.int32)
    sel = np.arange(len(fp))
    tol = {"g": 1e-12, "h": 1e-10, "i": 1e-12}[tag]
    if form != "gram-global":
        monkeypatch.setenv("PSK_NO_GRAM_GLOBAL", "1")
        if tag == "g":
            sel = sel[fp <= 1.0]
        if tag == "i":
            tol = 1e-11
    # 'g' (907 columns, 2,048 samples): the bar on coefficients is 5e-6 (measured 1e-10 .. 2.8e-6 at C <= 10); at C = 100 its
    # training folds are all but separable along some direction -- the objective is flat there to 5e-10 while the linear
    # predictor still moves by 12 % (one fit: 4,700 Newton steps to meet the rule at 1e-12, then 26 % from the optimum in
    # its coefficients and 4.6e-10 in its objective; liblinear stops 3.4e-4 above it) -- no gradient rule pins coefficients
    # there: those three fits are held to the objective alone, and the step cap ends them
    max_iter = 400 if tag == "g" else 5000
    coef, icpt, iters = ctx.logreg_l1_fit(X, y, fold, fp[sel], ff[sel], tol=tol, max_iter=max_iter)
    assert iters[fp[sel] <= 10].max() < max_iter, iters.tolist()
    Xd, ypm = X.astype(np.float64), 2.0 * y - 1.0
    for q, j in enumerate(sel):
        bar = 5e-6 if tag == "g" else 1e-6
        tr = fold != ff[j]
        aw, ab, grp = z["arb_coef_" + tag][j], float(z["arb_icpt_" + tag][j]), z["arb_group_" + tag][j]
        lin, alin = Xd[tr] @ coef[q] + icpt[q], Xd[tr] @ aw + ab
        obj = np.abs(coef[q]).sum() + abs(icpt[q]) + fp[j] * np.logaddexp(0.0, -ypm[tr] * lin).sum()
        assert obj == pytest.approx(float(z["arb_obj_" + tag][j]), rel=1e-8), (tag, form, j, fp[j], ff[j], int(iters[q]))
        if tag == "g" and fp[j] > 10:
            continue
        assert np.abs(lin - alin).max() <= bar * max(np.abs(alin).max(), 1e-300), (tag, form, j, np.abs(lin - alin).max())
        sums, asums = np.zeros(grp.max() + 1), np.zeros(grp.max() + 1)
        np.add.at(sums, grp, coef[q])
        np.add.at(asums, grp, aw)
        scale = max(np.abs(asums).max(), abs(ab), 1e-300)
        assert np.abs(sums - asums).max() <= bar * scale, (tag, form, j, fp[j], np.abs(sums - asums).max() / scale)
        assert abs(icpt[q] - ab) <= bar * scale, (tag, form, j)
        # the same support: zero where the optimum is zero (to the same dust), non-zero where it is not
        assert np.all(np.abs(sums[asums == 0]) <= 1e-6 * scale) and np.all(sums[np.abs(asums) > 1e-5 * scale] != 0), (tag, form, j)


@pytest.mark.parametrize("form", ["gram-global", "four-wave arrays"])
def test_l1_logreg_large_forms_at_the_reference_tolerance(ctx, monkeypatch, form):
    """... and at the tolerance the reference runs at (tol = 1e-4, modeling.py:241 of the CLI): all 15 recorded fits of the
    2,048 x 907 design, both forms.  A stopping rule bounds the violation, not the objective: the fits end at or above the
    certified optimum (never below it -- a lower value would mean the wrong objective) and within 5 % of it (measured:
    the accelerated form <= 0.6 %, the plain array descent <= 2.9 %; liblinear's own fits at tol = 1e-6 stop 0.04 % above)."""
    from helpers import large_design
    z = np.load(os.path.join(GOLDEN, "model_large_kat.npz"))
    X, y, fold = large_design(z, "g")
    fp, ff = z["fit_C_g"].astype(np.float64), z["fit_held_g"].astype(np.int32)
    if form != "gram-global":
        monkeypatch.setenv("PSK_NO_GRAM_GLOBAL", "1")
    coef, icpt, iters = ctx.logreg_l1_fit(X, y, fold, fp, ff, tol=1e-4, max_iter=1000)
    assert iters.max() < 1000
    ypm = 2.0 * y - 1.0
    _l1_stop_rule_holds(X, ypm, fold, fp, ff, coef, icpt, iters, range(len(fp)), tol=1e-4)
    rel = _l1_objectives(X, ypm, fold, fp, ff, coef, icpt) / z["arb_obj_g"] - 1
    assert rel.min() > -1e-12 and rel.max() < 5e-2, (form, float(rel.min()), float(rel.max()))


@pytest.mark.parametrize("tag", ["g", "h", "i"])
def test_grid_search_on_large_designs_matches_sklearn_gridsearchcv(ctx, tag):
    """The grid search exactly as the reference configures it (modeling.py:1078-1085, :1208-1216, :1512-1524: 13 values of C,
    cv = min(min class, 10), accuracy, tol = 1e-4, max_iter = 1000) on the three large designs, against scikit-learn's
    GridSearchCV(LogisticRegression(penalty='l1', solver='liblinear')) recorded under three seeds of liblinear's coordinate
    order (the reference leaves random_state=None: scikit-learn's own scores move from run to run).  Folds identical.
    Candidates with C <= 1 -- where liblinear's fits are converged at this tolerance and where every one of the three designs
    has its best candidate --: every (candidate, fold) score within ONE test sample of a recorded run's ('i' at C = 1000: two).
    Candidates with C > 1 on the 907-column design: there scikit-learn's score is a property of how far from the optimum
    liblinear's rule stops it (0.04 % .. 3 % in the objective) -- its own seeds differ by up to 4 test samples per fold, and
    the closer a solver gets to the optimum of these over-fitted models the LOWER the held-out accuracy (ours at tol = 1e-8:
    0.83 .. 0.89 against 0.895 .. 0.907) -- ours end nearer the optimum (test_l1_logreg_large_forms_at_the_reference_
    tolerance) and score 0.5 .. 1.1 % lower: held to 2 % in the mean.  The same C is chosen whenever the recorded runs agree
    on it and their best two candidates are more than one test sample per fold apart (on 'g' scikit-learn has five
    candidates tied at 0.9165 and takes the first; ours has one of them a sample lower and takes the second)."""
    from helpers import large_design
    from phenotypeseeker_amd.model import GridSearch, L1LogisticRegression
    z = np.load(os.path.join(GOLDEN, "model_large_kat.npz"))
    X, y, fold = large_design(z, tag)
    grid = [float(c) for c in z["grid"]]
    cv = int(min(np.bincount(y).min(), 10))
    gs = GridSearch(L1LogisticRegression(tol=1e-4, max_iter=1000), "C", grid, cv).fit(X, y, ctx)
    assert np.array_equal(gs.test_folds_, fold)
    ref = z["gs_split_scores_" + tag]                       # [seed][candidate][fold]
    ours = np.array([gs.cv_results_["split%d_test_score" % f] for f in range(cv)]).T
    one = 1.0 / np.bincount(fold)                           # one test sample, per fold
    d = np.abs(ours[None] - ref).min(axis=0) / one[None, :]  # test samples to the nearest recorded run, per (candidate, fold)
    small = np.array(grid) <= 1.0 + 1e-12
    assert np.all(d[small] <= 1.0 + 1e-9), (tag, d[small].max())
    assert np.all(d[~small] <= (8.0 if tag == "g" else 2.0) + 1e-9), (tag, d[~small].max())
    rm, om = ref.mean(axis=2), ours.mean(axis=1)
    assert np.all(om >= rm.min(axis=0) - 0.02) and np.all(om <= rm.max(axis=0) + 0.02), (tag, om, rm)
    best = z["gs_best_C_" + tag]
    top2 = np.sort(rm, axis=1)[:, -2:]
    if np.all(best == best[0]) and np.all(top2[:, 1] - top2[:, 0] > one.mean()):
        assert gs.best_params_["C"] == pytest.approx(float(best[0]))
    # in any case one of the candidates scikit-learn has within one test sample per fold of its best
    near = np.nonzero(rm.max(axis=0) >= rm.max() - one.mean() - 1e-12)[0]
    assert any(gs.best_params_["C"] == pytest.approx(grid[c]) for c in near), (tag, gs.best_params_["C"], [grid[c] for c in near])


def test_solver_form_knobs_are_validated_and_every_one_of_them_runs(ctx, monkeypatch):
    """the PSK_* variables that pick solver forms are reachable from a user's shell.  Garbage and
    out-of-range values are refused (PSK_EINVAL with the variable's name) instead of atoi()'s silent zero; every knob at a
    legal value gives a fit that stops by liblinear's rule with the objective of the default form (1e-3: the forms differ
    in what the rule leaves open, not in the optimum)."""
    from phenotypeseeker_amd._lib import PskError
    rng = np.random.default_rng(5)
    n, p = 1100, 230
    base = rng.random((n, 10)) < 0.4
    X = (base[:, rng.integers(0, 10, p)] ^ (rng.random((n, p)) < 0.06)).astype(np.float32)
    y = (base[:, 0] ^ (rng.random(n) < 0.1)).astype(np.int32)
    fold = (np.arange(n) % 3).astype(np.int32)
    fp, ff = np.array([0.1, 1.0, 1.0]), np.array([-1, 0, 2], np.int32)
    ypm = 2.0 * y - 1.0
    ref = ctx.logreg_l1_fit(X, y, fold, fp, ff, tol=1e-6, max_iter=2000)
    o_ref = _l1_objectives(X, ypm, fold, fp, ff, ref[0], ref[1])
    for var, bad in (("PSK_GG_MIN_P1", ["abc", "0", "-3", "12x", "99999999"]), ("PSK_FORCE_WMREG", ["8", "48", "sixteen", "16"]),
                     ("PSK_CG_MAX", ["-1", "1e3", "100000"]), ("PSK_POLISH_REPS", ["many", "999999"]),
                     ("PSK_GG_POLISH_FROM", ["0", "-4", "5000", " "])):
        for val in bad:
            monkeypatch.setenv(var, val)
            with pytest.raises(PskError) as e:
                ctx.logreg_l1_fit(X, y, fold, fp, ff, tol=1e-4, max_iter=100)
            assert e.value.code == -1 and var in str(e.value), (var, val, str(e.value))   # (16 words hold 1,024 samples: fewer than the design's 1,100)
            monkeypatch.delenv(var)
    good = [("PSK_GG_MIN_P1", "100"), ("PSK_GG_MIN_P1", "500"), ("PSK_FORCE_WMREG", "32"), ("PSK_FORCE_WMREG", "64"), ("PSK_CG_MAX", "0"),
            ("PSK_CG_MAX", "4"), ("PSK_CG_MAX", "64"), ("PSK_POLISH_REPS", "0"), ("PSK_POLISH_REPS", "8"), ("PSK_POLISH_REPS", "-32"),
            ("PSK_GG_POLISH_FROM", "1"), ("PSK_GG_POLISH_FROM", "8"), ("PSK_GG_POLISH_FROM", "1000"), ("PSK_NO_GRAM", "1"),
            ("PSK_NO_GRAM_GLOBAL", "1"), ("PSK_NO_CD_REGS", "1"), ("PSK_NO_GRAM", "0")]
    for var, val in good:
        monkeypatch.setenv(var, val)
        r = ctx.logreg_l1_fit(X, y, fold, fp, ff, tol=1e-6, max_iter=2000)
        monkeypatch.delenv(var)
        assert r[2].max() < 2000, (var, val, r[2])
        _l1_stop_rule_holds(X, ypm, fold, fp, ff, r[0], r[1], r[2], range(len(fp)), tol=1e-6)
        assert np.allclose(_l1_objectives(X, ypm, fold, fp, ff, r[0], r[1]), o_ref, rtol=1e-3, atol=0), (var, val)
    # the Lasso's two: the residual forms walk the covariance form's path (test_lasso_covariance_form_walks_sklearns_path)
    yc = X[:, :5].astype(np.float64) @ rng.normal(0, 1, 5) + rng.normal(0, 0.3, n)
    a = ctx.lasso_fit(X, yc, fold, [0.01, 0.1], [-1, 1], tol=1e-4, max_iter=1000)
    for var in ("PSK_NO_LASSO_COV", "PSK_NO_LASSO_BITS"):
        monkeypatch.setenv(var, "1")
        b = ctx.lasso_fit(X, yc, fold, [0.01, 0.1], [-1, 1], tol=1e-4, max_iter=1000)
        monkeypatch.delenv(var)
        assert np.array_equal(a[2], b[2]) and np.allclose(a[0], b[0], rtol=1e-7, atol=1e-10), var


def _lasso_large():
    z = np.load(os.path.join(GOLDEN, "lasso_large_kat.npz"))
    d = np.load(os.path.join(GOLDEN, "fit2048_907.npz"))
    X = np.unpackbits(d["Xbits"], axis=1)[:1024, : int(d["p"])].astype(np.float32)
    return z, X


def test_lasso_covariance_form_walks_sklearns_path(ctx, monkeypatch):
    """VERDICT r03 #3.  tests/golden/lasso_large_kat.npz: scikit-learn's Lasso as the reference configures it (tol = 1e-4,
    max_iter = 1000) on a 1,024 x 907 design, 13 alphas on all samples and on two training folds.  The covariance form
    (solver_lasso.hip) is the same cyclic descent with the same stop -- the duality gap, evaluated when scikit-learn
    evaluates it -- so it must END WHERE SCIKIT-LEARN ENDS: the same number of sweeps for every fit (1 ... 894, and 1,000
    where the limit cuts the descent off: there the fixture is no optimum, only the point the walk has reached), the
    coefficients to 1e-6 of the largest one and the intercept to 1e-6.  The four-wave kernel on the samples
    (PSK_NO_LASSO_COV=1: the fallback beyond 1,024 columns) and the float kernel (PSK_NO_LASSO_BITS=1: --real_counts) walk
    the same path."""
    z, X = _lasso_large()
    fp, ff = z["fit_alpha"].astype(np.float64), z["fit_held"].astype(np.int32)
    for env in (None, "PSK_NO_LASSO_COV", "PSK_NO_LASSO_BITS"):
        if env:
            monkeypatch.setenv(env, "1")
        coef, icpt, iters = ctx.lasso_fit(X, z["y"], z["fold"].astype(np.int32), fp, ff, tol=1e-4, max_iter=1000)
        if env:
            monkeypatch.delenv(env)
        assert np.array_equal(iters, z["n_iter"]), (env, iters.tolist(), z["n_iter"].tolist())
        for j in range(len(fp)):
            scale = max(np.abs(z["coef"][j]).max(), 1e-300)
            assert np.abs(coef[j] - z["coef"][j]).max() <= 1e-6 * scale, (env, j, fp[j], int(iters[j]))
            assert np.array_equal(coef[j] != 0, z["coef"][j] != 0), (env, j)
            assert icpt[j] == pytest.approx(float(z["icpt"][j]), rel=1e-6, abs=1e-9), (env, j)


def test_lasso_grid_search_on_the_large_design_matches_sklearn(ctx):
    """... and the grid search around it (modeling.py:1078-1080, :1208-1216: cv = 10 contiguous folds, R^2): every
    (alpha, fold) score and the chosen alpha equal GridSearchCV's.  (model.GridSearch solves each distinct column pattern
    once; a pattern's copies take no weight in scikit-learn's walk either, beyond rounding dust.)"""
    from phenotypeseeker_amd.model import GridSearch, LassoRegression
    z, X = _lasso_large()
    gs = GridSearch(LassoRegression(tol=1e-4, max_iter=1000), "alpha", [float(a) for a in z["alphas"]], 10).fit(X, z["y"], ctx)
    assert np.array_equal(gs.test_folds_, z["fold"])
    ours = np.array([gs.cv_results_["split%d_test_score" % f] for f in range(10)]).T
    assert np.allclose(ours, z["gs_split_scores"], rtol=1e-5, atol=1e-7), np.abs(ours - z["gs_split_scores"]).max()
    assert gs.best_params_["alpha"] == pytest.approx(float(z["gs_best_alpha"]))


@pytest.mark.parametrize("n", [90, 1000, 2500])
def test_lasso_bit_packed_kernel_equals_the_float_kernel(ctx, n, monkeypatch):
    """0/1 designs take the four-wave bit-packed Lasso (masked sums over the samples that have the k-mer, the residual
    as r' + c); the float kernel (PSK_NO_LASSO_BITS, also what --real_counts uses) is the same cyclic descent: run to
    convergence the two agree on every fit of a (value, fold) grid, folds and NA-free rows included."""
    rng = np.random.default_rng(n)
    p = 120
    base = rng.random((n, 10)) < 0.4
    X = (base[:, rng.integers(0, 10, p)] ^ (rng.random((n, p)) < 0.1)).astype(np.float32)
    X[:, 7] = 1.0                      # a constant column: zero norm, skipped
    X[:, 8] = X[:, 9]                  # a duplicated one
    y = X[:, :6].astype(np.float64) @ rng.normal(0, 1, 6) + rng.normal(0, 0.3, n)
    fold = (np.arange(n) % 4).astype(np.int32)
    fp = np.array([0.3, 0.03, 0.03, 0.003, 0.003], np.float64)
    ff = np.array([-1, 0, 3, 1, -1], np.int32)
    a = ctx.lasso_fit(X, y, fold, fp, ff, tol=1e-12, max_iter=100000)
    monkeypatch.setenv("PSK_NO_LASSO_BITS", "1")
    b = ctx.lasso_fit(X, y, fold, fp, ff, tol=1e-12, max_iter=100000)
    assert np.allclose(a[0], b[0], rtol=1e-7, atol=1e-9) and np.allclose(a[1], b[1], rtol=1e-9, atol=1e-10)
    assert (a[0][1:] != 0).any(axis=1).all() and np.all(a[0][:, 7] == 0)


def test_grid_search_matches_sklearn_cv(ctx):
    from phenotypeseeker_amd.model import GridSearch, L1LogisticRegression, LassoRegression
    z = np.load(os.path.join(GOLDEN, "model_kat.npz"))
    X, y = z["X2"], z["y2"]
    Cs = [float(c) for c in z["Cs"]]
    cv = int(min(np.bincount(y).min(), 10))
    gs = GridSearch(L1LogisticRegression(tol=1e-7, max_iter=3000), "C", Cs, cv).fit(X, y, ctx)
    assert np.array_equal(gs.test_folds_, z["skf_folds2"])
    # accuracy on tiny folds is piecewise constant: allow one flipped test sample per candidate
    assert np.allclose(gs.cv_results_["mean_test_score"], z["gs_mean_score2"], atol=1.0 / len(y) + 1e-12)
    assert gs.best_params_["C"] == pytest.approx(float(z["gs_best_C2"]))
    yc = z["yc2"]
    alphas = [float(a) for a in z["alphas"]]
    gl = GridSearch(LassoRegression(tol=1e-13, max_iter=200000), "alpha", alphas, 10).fit(X, yc, ctx)
    assert np.allclose(gl.cv_results_["mean_test_score"], z["lasso_gs_mean_score2"], rtol=1e-6, atol=1e-8)
    assert gl.best_params_["alpha"] == pytest.approx(float(z["lasso_gs_best_alpha2"]))
    import pickle
    g2 = pickle.loads(pickle.dumps(gs))
    assert np.array_equal(g2.predict(X), gs.predict(X)) and g2.predict_proba(X).shape == (len(y), 2)


def test_ridge_solver_matches_sklearn(ctx):
    """f4 / `--penalty L2`, continuous: unique optimum, coefficients pinned (tests/golden/model_l2_kat.npz)."""
    from oracle import oracle_model as OM
    z = np.load(os.path.join(GOLDEN, "model_kat.npz"))
    g = np.load(os.path.join(GOLDEN, "model_l2_kat.npz"))
    alphas = [float(a) for a in z["alphas"]]
    for tag, X, y in (("1", z["X1"], g["yc1"]), ("2", z["X2"], z["yc2"])):
        coef, icpt, iters = ctx.ridge_fit(X, y, np.zeros(len(y), np.int32), alphas, [-1] * len(alphas))
        assert np.allclose(coef, g["ridge_coef" + tag], rtol=1e-6, atol=1e-8), (tag, iters)
        assert np.allclose(icpt, g["ridge_icpt" + tag], rtol=1e-7, atol=1e-9)
    # held-out folds: every (alpha, fold) fit equals the oracle's fit on the training rows
    X, y = z["X2"], z["yc2"]
    folds = OM.kfold(len(y), 10).astype(np.int32)
    fp = [a for a in alphas[::4] for _ in range(10)]
    ff = [f for _ in alphas[::4] for f in range(10)]
    coef, icpt, _ = ctx.ridge_fit(X, y, folds, fp, ff)
    for j in range(len(fp)):
        w, b = OM.ridge_fit(X[folds != ff[j]], y[folds != ff[j]], fp[j])
        assert np.allclose(coef[j], w, rtol=1e-6, atol=1e-8) and icpt[j] == pytest.approx(b, rel=1e-7, abs=1e-9)


def test_l2_logreg_solver_matches_sklearn(ctx):
    """f4 / `--penalty L2`, binary: free intercept (lbfgs & co.) and liblinear's penalised intercept."""
    from oracle import oracle_model as OM
    z = np.load(os.path.join(GOLDEN, "model_kat.npz"))
    g = np.load(os.path.join(GOLDEN, "model_l2_kat.npz"))
    Cs = [float(c) for c in z["Cs"]]
    for tag in ("1", "2"):
        X, y = z["X" + tag], z["y" + tag]
        for name, pen, rtol in (("free", False, 1e-6), ("liblinear", True, 2e-5)):
            coef, icpt, iters = ctx.logreg_l2_fit(X, y, np.zeros(len(y), np.int32), Cs, [-1] * len(Cs), tol=1e-12,
                                                  max_iter=200, penalise_intercept=pen)
            assert np.allclose(coef, g["l2_%s_coef%s" % (name, tag)], rtol=rtol, atol=rtol * 0.1), (tag, name, iters)
            assert np.allclose(icpt, g["l2_%s_icpt%s" % (name, tag)], rtol=rtol, atol=rtol * 0.1)
            assert iters.max() < 60
    X, y = z["X2"], z["y2"]
    cv = int(min(np.bincount(y).min(), 10))
    folds = OM.stratified_kfold(y, cv).astype(np.int32)
    fp = [c for c in Cs[::4] for _ in range(cv)]
    ff = [f for _ in Cs[::4] for f in range(cv)]
    coef, icpt, _ = ctx.logreg_l2_fit(X, y, folds, fp, ff, tol=1e-12, max_iter=200)
    for j in range(0, len(fp), 3):
        w, b = OM.logreg_l2_fit(X[folds != ff[j]], y[folds != ff[j]], fp[j])
        assert np.allclose(coef[j], w, rtol=1e-6, atol=1e-8) and icpt[j] == pytest.approx(b, rel=1e-6, abs=1e-8)


def test_l2_grid_search_matches_sklearn_cv(ctx):
    from phenotypeseeker_amd.model import GridSearch, L2LogisticRegression, RidgeRegression
    z = np.load(os.path.join(GOLDEN, "model_kat.npz"))
    g = np.load(os.path.join(GOLDEN, "model_l2_kat.npz"))
    X, y, yc = z["X2"], z["y2"], z["yc2"]
    alphas, Cs = [float(a) for a in z["alphas"]], [float(c) for c in z["Cs"]]
    gr = GridSearch(RidgeRegression(tol=1e-4, max_iter=1000), "alpha", alphas, 10).fit(X, yc, ctx)
    assert np.allclose(gr.cv_results_["mean_test_score"], g["ridge_gs_mean_score2"], rtol=1e-6, atol=1e-8)
    assert gr.best_params_["alpha"] == pytest.approx(float(g["ridge_gs_best_alpha2"]))
    assert gr.n_unique_columns_ == X.shape[1]  # no column de-duplication under an L2 penalty
    cv = int(min(np.bincount(y).min(), 10))
    gl = GridSearch(L2LogisticRegression(tol=1e-10, max_iter=200), "C", Cs, cv).fit(X, y, ctx)
    assert np.allclose(gl.cv_results_["mean_test_score"], g["l2_gs_mean_score2"], atol=1e-12)
    assert gl.best_params_["C"] == pytest.approx(float(g["l2_gs_best_C2"]))
    sk = gl.to_sklearn()
    assert np.array_equal(sk.predict(X), gl.predict(X)) and "penalty='l1'" not in repr(gl.best_estimator_)
    import pickle
    g2 = pickle.loads(pickle.dumps(gr))
    assert np.allclose(g2.predict(X), gr.predict(X))


def test_device_survivor_exchange_matches_get_results():
    """The multi-GPU hand-off (psk_export_survivors_async + ncclAllGather on libpsk.so's own communicator, then
    the list all-to-all through ncclSend / ncclRecv) as a one-rank RCCL group in a fresh process -- no torch in
    it: the gathered records equal psk_get_results / psk_get_rows."""
    import subprocess
    import sys
    from helpers import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_exchange_worker.py")], cwd=ROOT, timeout=600,
                       capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "exchange ok" in r.stdout


@pytest.mark.parametrize("k,n,length", [(31, 40, 3000), (32, 3, 5000), (27, 3000, 120), (26, 4096, 100)])
def test_presence_when_word_and_sample_do_not_fit_one_u64(ctx, oracle, k, n, length):
    """2k + ceil(log2 N) > 64 (k = 27..32 with many samples) takes the key + payload sort; k = 26 with
    4096 samples is the largest packed case.  Union and bit rows must equal the oracle's either way."""
    from phenotypeseeker_amd.synth import GenomeSet
    gs = GenomeSet(n, length, seed=k * 7 + n, gene_len=40, sub_rate=0.01)
    ctx.begin(k, n)
    lists = []
    for i in range(n):
        _, fa = gs.sample(i)
        nu, _ = ctx.count_kmers(i, fa)
        lists.append(oracle.count_kmers(fa, k)[0])
        assert nu == len(lists[-1])
    M = ctx.build_presence()
    uw = oracle.union(lists)
    assert M == len(uw) and np.array_equal(ctx.get_union(), uw)
    bits = ctx.get_rows(np.arange(M, dtype=np.uint64))
    assert np.array_equal(bits, oracle.presence_bits(lists, uw, wpr=bits.shape[1]))


@pytest.mark.parametrize("k,n,length", [(9, 1300, 300), (11, 2048, 400), (10, 4100, 150), (8, 9000, 120)])
def test_tiled_presence_build_with_thousands_of_samples(ctx, oracle, k, n, length, monkeypatch):
    """Rows of hundreds of bytes: a tile of the sort-free build then holds fewer rows (512 ... 64) than its workgroup
    has threads.  Union and every bit row against the oracle, and against the sort route on the same lists."""
    from phenotypeseeker_amd.synth import GenomeSet
    gs = GenomeSet(n, length, seed=k + n, gene_len=min(60, length // 3))
    datas = [gs.sample(i)[1] for i in range(n)]
    ctx.begin(k, n)
    nus = []
    for lo in range(0, n, 512):
        nu, _ = ctx.count_kmers_batch(lo, datas[lo:lo + 512], 4)
        nus += nu
    lists = [oracle.count_kmers(d, k)[0] for d in datas]
    assert nus == [len(w) for w in lists]
    uw = oracle.union(lists)
    m = ctx.build_presence()
    assert m == len(uw) and np.array_equal(ctx.get_union(), uw)
    wpr = ctx.presence_shape()[1]
    rows = ctx.get_rows(np.arange(m, dtype=np.uint64))
    assert np.array_equal(rows, oracle.presence_bits(lists, uw, wpr=wpr))
    monkeypatch.setenv("PSK_NO_TILED_PRESENCE", "1")       # the same lists through the sort route
    assert ctx.build_presence() == m
    assert np.array_equal(ctx.get_union(), uw) and np.array_equal(ctx.get_rows(np.arange(m, dtype=np.uint64)), rows)


def _collapse(clean):
    """runs of window breaks -> one, none in front (what the host machine writes)"""
    out = bytearray()
    for c in clean:
        if c == 10 and (not out or out[-1] == 10):
            continue
        out.append(c)
    return bytes(out)


def test_gpu_framing_equals_the_host_state_machine(ctx, oracle, monkeypatch, capfd):
    """record framing on the device (frame_gpu.hip).  On all 161 probed tokeniser cases, on FASTA
    with every awkward byte (headers inside lines, control bytes, IUPAC codes, '>' in sequence, CRLF, no final
    newline, text before the first record, NUL), on four-line FASTQ and (r06) on FASTQ that is NOT four lines per record the
    device's clean stream equals the host machine's up to collapsed breaks; and the lists counted with device framing equal
    those counted with host framing and glistmaker's (37 of the reference's 161 fixtures are such irregular FASTQ)."""
    from helpers import tokenizer_cases
    from phenotypeseeker_amd.engine import frame_sequence
    from phenotypeseeker_amd.synth import GenomeSet, fastq_reads
    rng = np.random.default_rng(11)
    gs = GenomeSet(3, 30_000, seed=8, gene_len=200, contigs=5)
    extra = [gs.sample(0)[1], gs.sample(1)[1].replace(b"\n", b"\r\n"), b"junk before\n" + gs.sample(2)[1][:-1],
             fastq_reads(gs.codes(0), 400, 100, seed=[1, 2]), fastq_reads(gs.codes(1), 50, 150, seed=[1, 3])[:-1],
             b">only header", b">h\n", b"@r\nACGT\n+\nIIII\n", b"@r\nACGTNNACGT\n+r\n@@@@@@@@@@\n@s\nGGGGCCCC\n+\n>>>>>>>>\n",
             b">a\nACGT>b\nGGGG\n>c\n\n\nTT\x01\x02TT\n", b">x\nACGT\x00ACGT\n", b"@r\nAC\nGT\n+\nIIII\n", b"@r\nACGT\n+\nIIII\n\n@s\nACGT\n+\nIIII\n"]
    body = bytearray(rng.integers(1, 256, 60_000, dtype=np.uint8).tobytes())
    extra.append(b">fuzz\n" + bytes(body))                           # every byte value but NUL, '>' and '\n' sprinkled in
    # r06: FASTQ that is NOT four lines per record is framed on the device too (the scan of line kinds) -- sequences and
    # qualities over several lines (the machine swallows the first byte of a continued sequence line), blank lines inside and
    # between records, quality lines that begin with '@', a FASTA record behind a FASTQ one, CRLF: more cases of that kind
    wrapped = [b"@r\nAC\nGT\n+\nIIII\n", b"@r\nACGT\n+\nIIII\n\n@s\nACGT\n+\nIIII\n",
               b"@r1\nACGTACGTAC\nGGGTTTAAAC\nCCA\n+r1\nIIIIIIIIII\nIIIIIIIIII\nIII\n@r2\nTTTTGGGGCCCCAAAA\n+\n@IIIIIIIIIIIIIII\n@r3\nACGTAACC\n+\nIIIIIIII\n",
               b"@r\r\nACGTACGT\r\nTTGGCCAA\r\n+\r\nIIIIIIII\r\nIIIIIIII\r\n@s\r\nGGGGCCCC\r\n+\r\nIIIIIIII\r\n",
               b"@r\nACGT\n\nTTGG\n\n\nCCAA\n+\nIIII\n\n\n@s\nGGGG\n+\n\nIIII\n@t\nAAAA\n+\nIIII\n",
               b"@r\n+ACGT\nTTTT\n+\nIIII\n@s\n\nACGT\n+\nIIII\n", b"@r\nACGT\n@notaheader\nGGGG\n+\nIIII\nIIII\n@s\nCCCCAAAA\n+\nIIIIIIII\n",
               b"@r\nACGTNNNNACGT\nNNNN\nACGTACGT\n+\n" + b"I" * 24 + b"\n>fasta_behind\nACGTACGTACGT\n@s\nTTTTTTTTGGGG\n+\nIIIIIIIIIIII\n",
               b"@r\nACGT", b"@r\nACGT\nAC", b"@r\nACGT\n+", b"@r\nACGT\n+\nII\nII", b"text first\nmore @r\nACGTACGT\nACGT\n+\nIIIIIIII\nIIII\n"]
    reads = fastq_reads(gs.codes(2), 300, 150, seed=[4, 5]).split(b"\n")
    big = bytearray()
    for q in range(0, len(reads) - 3, 4):     # every record's sequence and quality over lines of 60, a blank line behind every third
        h, sq, pl, ql = reads[q:q + 4]
        big += h + b"\n" + b"\n".join(sq[c:c + 60] for c in range(0, len(sq), 60)) + b"\n" + pl + b"\n" + b"\n".join(ql[c:c + 60] for c in range(0, len(ql), 60)) + b"\n"
        if (q // 4) % 3 == 2:
            big += b"\n"
    wrapped.append(bytes(big))                 # 60 KB: tiles, waves and threads that begin inside every kind of line
    extra += wrapped
    n_gpu = n_irregular = 0
    cases = [(d, k) for d, k, _ in tokenizer_cases()] + [(d, 13) for d in extra]
    for data, k in cases:
        want = frame_sequence(data)
        got = ctx.frame_sequence_gpu(data)
        assert got is not None, data[:80]      # nothing is handed to the host's state machine any more
        n_gpu += 1
        assert _collapse(got) == _collapse(want), data[:80]
        at, gt = data.find(b"@"), data.find(b">")
        if at >= 0 and (gt < 0 or at < gt):
            lines = data[at:].split(b"\n")
            lines = lines[:-1] if lines[-1] == b"" else lines
            n_irregular += any((j % 4 == 0 and not l.startswith(b"@")) or (j % 4 == 2 and not l.startswith(b"+")) for j, l in enumerate(lines))
    assert n_gpu > 170 and n_irregular >= 45, (n_gpu, n_irregular)
    # lists: device framing (default) against host framing and the reference's
    for data, k, ref in tokenizer_cases():
        if ref is None or not 1 <= k <= 32:
            continue
        ctx.begin(k, 1)
        nu, nt = ctx.count_kmers_batch(0, [data], 1)
        w, f = ctx.get_list(0, nu[0])
        assert oracle.list_bytes(k, w, f) == ref, data[:60]
    datas = extra + [b""]
    for k in (13, 16):
        ctx.begin(k, len(datas))
        monkeypatch.setenv("PSK_TRACE", "1")
        capfd.readouterr()
        nu, nt = ctx.count_kmers_batch(0, datas, 3)
        monkeypatch.delenv("PSK_TRACE")
        # the DEVICE route for the samples that are not four-line FASTQ, by the library's own account
        said = capfd.readouterr().err
        on_device = [int(l.split("sample ")[1].split(":")[0]) for l in said.splitlines() if "framed on the device by the scan of line kinds" in l]
        assert len(on_device) >= len(wrapped) - 4 and len(extra) - 1 in on_device, on_device     # (the truncated ones may be regular)
        lists = [ctx.get_list(i, nu[i]) for i in range(len(datas))]
        monkeypatch.setenv("PSK_HOST_FRAMING", "1")
        ctx.begin(k, len(datas))
        nu2, nt2 = ctx.count_kmers_batch(0, datas, 3)
        monkeypatch.delenv("PSK_HOST_FRAMING")
        assert list(nu) == list(nu2) and list(nt) == list(nt2)
        for i, d in enumerate(datas):
            a, b = ctx.get_list(i, nu2[i])
            ow, of, ont = oracle.count_kmers(d, k)
            assert np.array_equal(a, lists[i][0]) and np.array_equal(b, lists[i][1]), i
            assert np.array_equal(a, ow) and np.array_equal(b, of) and nt[i] == ont, i


@pytest.mark.parametrize("k", [11, 12, 13])
def test_dense_counting_paths_agree_with_the_oracle(ctx, oracle, k, monkeypatch):
    """k = 11..13 counts without a sort (dense_count.hip): the presence-bit pass for buckets with few keys, the
    counter-table pass for the ones it leaves (more than 2048 repeats in a bucket: tandem repeats, homopolymers), the
    table pass alone (PSK_DC_TABLE: what deep read sets take), 32-bit counters for buckets with 65,536 keys or more,
    and the sort route (PSK_NO_DENSE) -- every list, every count, and the matrix equal the oracle's."""
    from phenotypeseeker_amd.synth import GenomeSet
    rng = np.random.default_rng(k)
    gs = GenomeSet(3, 150_000, seed=31 + k, gene_len=300, gc=0.31, contigs=4)
    unit = bytes(rng.choice(list(b"ACGT"), size=37).tolist())
    datas = [gs.sample(i)[1] for i in range(3)]
    datas.append(b">tandem\n" + unit * 6000 + b"\n>polyA\n" + b"A" * 70_000 + b"\n" + datas[0])      # over-full buckets
    datas.append(b">deep\n" + (unit * 4 + b"N") * 3000 + bytes(rng.choice(list(b"ACGT"), size=400_000).tolist()) + b"\n")
    datas.append(b"")
    datas.append(b">short\nACGTACGTAC\n")
    ref = [oracle.count_kmers(d, k) for d in datas]
    lists = [r[0] for r in ref]
    uw = oracle.union(lists)

    def check(batch):
        ctx.begin(k, len(datas))
        if batch:
            nu, nt = ctx.count_kmers_batch(0, datas, 3)
        else:
            got = [ctx.count_kmers(i, d) for i, d in enumerate(datas)]
            nu, nt = [g[0] for g in got], [g[1] for g in got]
        assert list(nu) == [len(r[0]) for r in ref] and list(nt) == [r[2] for r in ref]
        q = np.concatenate([lists[3][::7], np.array([0, 1, (1 << (2 * k)) - 1], dtype=np.uint64)])
        want = np.zeros(len(q), np.uint32)
        pos = np.searchsorted(lists[3], q)
        hit = (pos < len(lists[3])) & (lists[3][np.minimum(pos, len(lists[3]) - 1)] == q)
        want[hit] = ref[3][1][pos[hit]]
        assert np.array_equal(ctx.lookup_counts(3, q), want)                    # before any list is materialised
        m = ctx.build_presence()
        assert m == len(uw) and np.array_equal(ctx.get_union(), uw)
        assert np.array_equal(ctx.get_rows(np.arange(m, dtype=np.uint64)), oracle.presence_bits(lists, uw, wpr=ctx.presence_shape()[1]))
        for i, (w, f, _) in enumerate(ref):
            gw, gf = ctx.get_list(i, nu[i])
            assert np.array_equal(gw, w) and np.array_equal(gf, f), i

    check(batch=True)
    check(batch=False)
    monkeypatch.setenv("PSK_DC_TABLE", "1")
    check(batch=True)
    monkeypatch.delenv("PSK_DC_TABLE")
    monkeypatch.setenv("PSK_NO_DENSE_PRESENCE", "1")      # dense lists, list-based matrix build
    check(batch=True)
    monkeypatch.delenv("PSK_NO_DENSE_PRESENCE")
    monkeypatch.setenv("PSK_NO_DENSE", "1")
    check(batch=True)


@pytest.mark.parametrize("k", [14, 15, 16, 17, 18, 21, 27, 31, 32])
def test_bucketed_sort_route_equals_the_radix_route_and_the_oracle(ctx, oracle, k, monkeypatch, capfd):
    """k = 14..32 (bucket_count.hip; r05: the 64-bit words of k >= 17 as well -- what `glistmaker -w <k>` is asked for at
    modeling.py:309-310 for any `-l` up to 32): once a list of the run has given the splitters the later samples are counted by
    partition + one LDS sort per bucket.  Same lists as the radix route (PSK_NO_BUCKET_SORT) and as the oracle: uniform
    and AT-rich genomes in one run (the splitters come from the first), a sample with a tandem repeat and a homopolymer
    (one word 70,000 times), an empty and a tiny one; whole space and under a slab filter; and with a bucket capacity of
    64 words (PSK_BS_CAP), which sends every later sample through the fall-back."""
    from phenotypeseeker_amd.synth import GenomeSet
    rng = np.random.default_rng(k)
    gs = GenomeSet(3, 220_000, seed=5 + k, gene_len=300)
    at = GenomeSet(2, 180_000, seed=9 + k, gene_len=300, gc=0.29, contigs=5)
    unit = bytes(rng.choice(list(b"ACGT"), size=41).tolist())
    datas = [gs.sample(0)[1], gs.sample(1)[1], at.sample(0)[1], gs.sample(2)[1], at.sample(1)[1],
             b">tandem\n" + unit * 5000 + b"\n>polyA\n" + b"A" * 70_000 + b"\n" + gs.sample(1)[1], b"", b">short\nACGTACGTACGTACGTACGT\n"]
    ref = [oracle.count_kmers(d, k) for d in datas]
    space = 1 << (2 * k)

    def run(lo, hi, batch):
        ctx.begin(k, len(datas), lo, hi if hi < space else 0)      # (0: no upper bound -- 4^32 does not fit the argument)
        if batch:
            nu, nt = ctx.count_kmers_batch(0, datas, 3)
        else:
            got = [ctx.count_kmers(i, d) for i, d in enumerate(datas)]
            nu, nt = [g[0] for g in got], [g[1] for g in got]
        out = []
        for i, (w, f, _) in enumerate(ref):
            sel = (w >= lo) & (w < (hi or space))
            gw, gf = ctx.get_list(i, nu[i])
            assert np.array_equal(gw, w[sel]) and np.array_equal(gf, f[sel]), (i, lo, hi, batch)
            out.append((gw, gf))
        return out

    monkeypatch.setenv("PSK_TRACE", "1")
    for lo, hi in ((0, 0), (space // 5, space // 2), (space - space // 3, space)):
        capfd.readouterr()
        run(lo, hi, True)
        # only the sample with 70,000 copies of one word may leave the bucketed route (genomes do not: the splitters fit them)
        assert capfd.readouterr().err.count("takes the fall-back") <= 1, (lo, hi)
        run(lo, hi, False)
    monkeypatch.setenv("PSK_BS_CAP", "64")
    run(0, 0, True)
    run(space // 5, space // 2, True)
    monkeypatch.delenv("PSK_BS_CAP")
    monkeypatch.setenv("PSK_NO_BUCKET_SORT", "1")
    run(0, 0, True)


def test_dense_counting_under_a_slab_filter(ctx, oracle):
    """The dense form with slab bounds that cut through buckets: the slabs' lists concatenate to the whole list and the
    slabs' matrices to the whole matrix."""
    from phenotypeseeker_amd.synth import GenomeSet
    k = 13
    gs = GenomeSet(6, 120_000, seed=77, gene_len=300)
    datas = [gs.sample(i)[1] for i in range(6)]
    ref = [oracle.count_kmers(d, k)[:2] for d in datas]
    uw = oracle.union([r[0] for r in ref])
    bounds = [0, 1_234_567, 1_234_567 + 40_000, 30_000_001, 0]
    got_w, got_f, got_u, got_rows = [[] for _ in datas], [[] for _ in datas], [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        ctx.begin(k, len(datas), lo, hi)
        nu, _ = ctx.count_kmers_batch(0, datas, 2)
        m = ctx.build_presence()
        got_u.append(ctx.get_union())
        got_rows.append(ctx.get_rows(np.arange(m, dtype=np.uint64)))
        for i in range(len(datas)):
            w, f = ctx.get_list(i, nu[i])
            got_w[i].append(w)
            got_f[i].append(f)
    for i, (w, f) in enumerate(ref):
        assert np.array_equal(np.concatenate(got_w[i]), w) and np.array_equal(np.concatenate(got_f[i]), f), i
    assert np.array_equal(np.concatenate(got_u), uw)
    assert np.array_equal(np.concatenate(got_rows), oracle.presence_bits([r[0] for r in ref], uw, wpr=ctx.presence_shape()[1]))


@pytest.mark.parametrize("k,n,length,chunk", [(16, 40, 30_000, 100_000), (31, 300, 900, 20_000), (13, 12, 60_000, 50_000)])
def test_sort_route_in_word_range_chunks(ctx, oracle, k, n, length, chunk, monkeypatch):
    """More (word, sample) pairs than one sort holds (2^32 per GPU; config 3 on two GPUs has 4.8 x 10^9 per rank): the
    slab's word range is cut at list quantiles into chunks that are sorted on their own (PSK_PAIR_CHUNK shrinks the
    chunk so that a small set takes a dozen of them).  Union and rows equal the one-chunk build and the oracle."""
    from phenotypeseeker_amd.synth import GenomeSet
    gs = GenomeSet(n, length, seed=k + n, gene_len=200, sub_rate=0.01)
    datas = [gs.sample(i)[1] for i in range(n)]
    datas[3] = b""
    ctx.begin(k, n)
    ctx.count_kmers_batch(0, datas, 4)
    monkeypatch.setenv("PSK_NO_TILED_PRESENCE", "1")
    monkeypatch.setenv("PSK_NO_MERGE_PRESENCE", "1")      # the sort route itself (r03: k <= 17 takes the streaming merge)
    m = ctx.build_presence()
    uw, rows = ctx.get_union(), ctx.get_rows(np.arange(m, dtype=np.uint64))
    lists = [oracle.count_kmers(d, k)[0] for d in datas]
    assert np.array_equal(uw, oracle.union(lists))
    assert np.array_equal(rows, oracle.presence_bits(lists, uw, wpr=ctx.presence_shape()[1]))
    monkeypatch.setenv("PSK_PAIR_CHUNK", str(chunk))
    assert sum(len(w) for w in lists) > 4 * chunk
    assert ctx.build_presence() == m
    assert np.array_equal(ctx.get_union(), uw) and np.array_equal(ctx.get_rows(np.arange(m, dtype=np.uint64)), rows)


@pytest.mark.parametrize("k,n,length,env", [
    (16, 40, 30_000, {}),                                                    # one wave, default tiles
    (16, 70, 20_000, {"PSK_MERGE_TILE_PAIRS": "64"}),                        # two waves, thousands of tiny tiles
    (14, 130, 9_000, {"PSK_MERGE_RCAP": "7", "PSK_MERGE_TILE_PAIRS": "500"}),  # padded column; tiles streamed in batches of 7 rows
    (15, 1100, 3_000, {"PSK_MERGE_TILE_PAIRS": "3000"}),                     # two sample groups (global bitmap by atomics)
    (16, 1100, 2_000, {"PSK_MERGE_RCAP": "40", "PSK_MERGE_RANGES": "3"}),    # two groups, batches, three long ranges
    (17, 24, 50_000, {"PSK_MERGE_RANGES": "1"}),                             # k = 17, the widest eligible space; one range
])
def test_merge_presence_build_equals_the_oracle_and_the_sort_route(ctx, oracle, k, n, length, env, monkeypatch):
    """r03: the presence build for k >= 14 is a streaming 64-way merge of the sorted lists per wave (presence_merge.hip)
    instead of a sort of all (word, sample) pairs.  Union and every row equal the oracle's and the sort route's, with
    empty samples, identical samples, a slab filter, tiles of every size, tiles streamed in several batches of rows,
    and more than 1,024 samples (two workgroups per tile range)."""
    from phenotypeseeker_amd.synth import GenomeSet
    gs = GenomeSet(n, length, seed=3 * k + n, gene_len=150, sub_rate=0.01)
    datas = [gs.sample(i)[1] for i in range(n)]
    datas[3] = b""
    datas[n - 1] = b""
    datas[5] = datas[4]
    lists = [oracle.count_kmers(d, k)[0] for d in datas]
    monkeypatch.setenv("PSK_NO_TILED_PRESENCE", "1")
    monkeypatch.setenv("PSK_TRACE", "1")
    for name, val in env.items():
        monkeypatch.setenv(name, val)
    for lo, hi in ((0, 0), (int(lists[0][len(lists[0]) // 3]) + 1, int(lists[0][2 * len(lists[0]) // 3]) + 7)):
        ctx.begin(k, n, lo, hi)
        for s0 in range(0, n, 64):
            ctx.count_kmers_batch(s0, datas[s0:s0 + 64], 4)
        kept = [w[(w >= lo) & ((w < hi) if hi else np.ones(len(w), bool))] for w in lists]
        monkeypatch.delenv("PSK_NO_MERGE_PRESENCE", raising=False)
        # r04: pass 2 replays the records pass 1 left -- the default from 16 M pairs on (r05: smaller builds skip the pool's
        # fixed gigabyte), asked for here by the knob
        monkeypatch.setenv("PSK_MERGE_REC_DIV", "12")
        m = ctx.build_presence()
        monkeypatch.delenv("PSK_MERGE_REC_DIV")
        uw, rows = ctx.get_union(), ctx.get_rows(np.arange(m, dtype=np.uint64))
        want = oracle.union(kept)
        assert m == len(want) and np.array_equal(uw, want), (lo, hi)
        assert np.array_equal(rows, oracle.presence_bits(kept, uw, wpr=ctx.presence_shape()[1]))
        # without records pass 2 merges the lists again (the default at this size, and PSK_MERGE_REC_DIV=0), and so it does
        # when the record pool overflows (regions of eight chunks)
        for envs in ({}, {"PSK_MERGE_REC_DIV": "0"}, {"PSK_MERGE_REC_DIV": "12", "PSK_MERGE_REC_REGION": "8"}):
            for name, val in envs.items():
                monkeypatch.setenv(name, val)
            assert ctx.build_presence() == m
            assert np.array_equal(ctx.get_union(), uw) and np.array_equal(ctx.get_rows(np.arange(m, dtype=np.uint64)), rows), envs
            for name in envs:
                monkeypatch.delenv(name)
        monkeypatch.setenv("PSK_NO_MERGE_PRESENCE", "1")
        assert ctx.build_presence() == m
        assert np.array_equal(ctx.get_union(), uw) and np.array_equal(ctx.get_rows(np.arange(m, dtype=np.uint64)), rows)


@pytest.mark.parametrize("k,n,length,env", [
    (18, 40, 30_000, {}),                                                    # one wave
    (21, 70, 20_000, {"PSK_MERGE_TILE_PAIRS": "64"}),                        # two waves, thousands of tiny tiles
    (27, 130, 9_000, {"PSK_MERGE_TILE_PAIRS": "500"}),                       # padded column
    (31, 1100, 3_000, {"PSK_MERGE_TILE_PAIRS": "3000"}),                     # two sample groups: two record streams per range
    (18, 1100, 3_000, {"PSK_MERGE_TILE_PAIRS": "3000"}),                     # ... with 32-bit cursors (ranges of 2^26 word values)
    (32, 1100, 2_000, {"PSK_MERGE_REC_DIV": "1"}),                           # the whole 64-bit space (a mutation costs 32 k-mers: the samples share too little for the default pool)
    (18, 300, 6_000, {"PSK_WIDE_MERGE_64": "1"}),                            # 64-bit cursors where the 32-bit ones would do (the first case)
    (19, 24, 50_000, {"PSK_MERGE_RANGES": "1", "PSK_MERGE_REC_DIV": "2"}),   # one range, one wave: every distinct word is a record
])
def test_wide_merge_presence_build_equals_the_oracle_and_the_sort_route(ctx, oracle, k, n, length, env, monkeypatch, capfd):
    """r05: the presence build beyond the value-bitmap spaces (k >= 18; `-l` takes up to 32, scripts/phenotypeseeker:89-92, and
    the reference's union / mapping, modeling.py:317-380, are the same glistcompare / glistquery calls at any k): the streaming
    merge leaves (word, ballot) records, the union is the sorted distinct record words, the rows are filled by replaying the
    records (presence_merge.hip build_presence_merge_wide).  Union and every row equal the oracle's and the sort route's, with
    empty samples, identical samples, a slab filter, more than 1,024 samples; a record pool that is too small (samples that
    share nothing, or PSK_MERGE_REC_REGION) hands the build to the sort route, with the same result."""
    from phenotypeseeker_amd.synth import GenomeSet
    gs = GenomeSet(n, length, seed=3 * k + n, gene_len=150, sub_rate=0.01)
    datas = [gs.sample(i)[1] for i in range(n)]
    datas[3] = b""
    datas[n - 1] = b""
    datas[5] = datas[4]
    lists = [oracle.count_kmers(d, k)[0] for d in datas]
    monkeypatch.setenv("PSK_TRACE", "1")
    for name, val in env.items():
        monkeypatch.setenv(name, val)
    for lo, hi in ((0, 0), (int(lists[0][len(lists[0]) // 3]) + 1, int(lists[0][2 * len(lists[0]) // 3]) + 7)):
        ctx.begin(k, n, lo, hi)
        for s0 in range(0, n, 64):
            ctx.count_kmers_batch(s0, datas[s0:s0 + 64], 4)
        kept = [w[(w >= lo) & ((w < hi) if hi else np.ones(len(w), bool))] for w in lists]
        capfd.readouterr()
        m = ctx.build_presence()
        err = capfd.readouterr().err
        assert "wide merge build:" in err, (lo, hi)      # the route under test ran
        if k >= 27 or "PSK_WIDE_MERGE_64" in env or "PSK_MERGE_RANGES" in env:
            assert "(64-bit cursors)" in err, (k, lo, hi)          # ranges of 2^32 word values and more
        elif k == 18 and n == 1100 and not (lo or hi):
            assert "(32-bit cursors)" in err, (k, lo, hi)          # words relative to each range's first bound (1,100 ranges of ~2^26 values)
        uw, rows = ctx.get_union(), ctx.get_rows(np.arange(m, dtype=np.uint64))
        want = oracle.union(kept)
        assert m == len(want) and np.array_equal(uw, want), (lo, hi)
        assert np.array_equal(rows, oracle.presence_bits(kept, uw, wpr=ctx.presence_shape()[1]))
        # a record region of eight chunks overflows: the sort route takes over; and the sort route asked for by the knob
        for envs, said in (({"PSK_MERGE_REC_REGION": "8"}, "overflowed"), ({"PSK_NO_WIDE_MERGE": "1"}, None)):
            for name, val in envs.items():
                monkeypatch.setenv(name, val)
            capfd.readouterr()
            assert ctx.build_presence() == m
            err = capfd.readouterr().err
            assert "wide merge build:" not in err and (said is None or said in err or n < 64), envs
            assert np.array_equal(ctx.get_union(), uw) and np.array_equal(ctx.get_rows(np.arange(m, dtype=np.uint64)), rows), envs
            for name in envs:
                monkeypatch.delenv(name)


def test_full_size_ingest_properties(ctx, oracle):
    """BASELINE config-2 sized ingest (256 x 5 Mbp, k = 13) checked through size-independent
    properties: every list is strictly ascending with sum(freq) = number of windows, the union is
    strictly ascending, every matrix row has at least one bit, the column sums of the matrix equal the
    per-sample list lengths, and a sample of rows/lists equals the oracle's."""
    from phenotypeseeker_amd.synth import GenomeSet
    n, L, k = 256, 5_000_000, 13
    gs = GenomeSet(n, L, seed=4242)
    ctx.begin(k, n)
    uniq = []
    for i in range(n):
        _, fa = gs.sample(i)
        nu, nt = ctx.count_kmers(i, fa)
        uniq.append(nu)
        assert nt == len(gs.codes(i)) - k + 1 if i % 64 == 0 else nt > L - k
    for i in (0, 77, 255):
        w, f = ctx.get_list(i, uniq[i])
        assert np.all(np.diff(w.astype(np.int64)) > 0) and int(f.sum()) > L - k
        if i == 77:
            ow, of, _ = oracle.count_kmers(gs.sample(i)[1], k)
            assert np.array_equal(w, ow) and np.array_equal(f, of)
    M = ctx.build_presence()
    assert max(uniq) <= M <= min(sum(uniq), 1 << (2 * k - 1) + (1 << (k - 1)))
    uw = ctx.get_union()
    assert np.all(np.diff(uw.astype(np.int64)) > 0)
    colsum = np.zeros(n, dtype=np.int64)
    chunk = 1 << 21
    for lo in range(0, M, chunk):
        rows = ctx.get_rows(np.arange(lo, min(lo + chunk, M), dtype=np.uint64))
        assert np.all(rows.any(axis=1))
        b = np.unpackbits(rows.view(np.uint8), axis=1, bitorder="little")[:, :n]
        colsum += b.sum(axis=0, dtype=np.int64)
    assert colsum.tolist() == uniq
    # rows of sample 77's first words carry bit 77
    w77, _ = ctx.get_list(77, uniq[77])
    idx = np.searchsorted(uw, w77[:1000])
    assert np.array_equal(uw[idx], w77[:1000])
    r = ctx.get_rows(idx.astype(np.uint64))
    assert np.all((r[:, 77 >> 6] >> np.uint64(77 & 63)) & np.uint64(1))


def test_abi_misuse_returns_errors_not_crashes(ctx):
    from phenotypeseeker_amd._lib import PskError
    from phenotypeseeker_amd.engine import PskContext
    with PskContext(0) as c:
        with pytest.raises(PskError, match="psk_begin"):
            c.count_kmers(0, b">r\nACGT\n")
        with pytest.raises(PskError):
            c.begin(0, 1)
        with pytest.raises(PskError):
            c.begin(33, 1)
        c.begin(13, 2)
        with pytest.raises(PskError, match="out of range"):
            c.count_kmers(5, b">r\nACGT\n")
        c.count_kmers(0, b">r\nACGTACGTACGTACGTACGT\n")
        with pytest.raises(PskError, match="has not been counted"):
            c.build_presence()
        with pytest.raises(PskError, match="no presence matrix"):
            c.chi2_scan(np.zeros(2, np.int8), None, 1, 2, 0.05, True, 1)
        c.count_kmers(1, b"")                       # empty sample is legal
        assert c.build_presence() == 2
        with pytest.raises(PskError, match="row index"):
            c.get_rows(np.array([99], dtype=np.uint64))
        with pytest.raises(PskError, match="no scan"):
            c.get_results(1)
        assert c.chi2_scan(np.array([1, 0], np.int8), None, 1, 2, 1.5, True, 2) >= 0
    with pytest.raises(PskError):
        PskContext(99)


def test_batch_counting_equals_single_calls(ctx, oracle):
    ds = load_dataset("ds_bonf")
    k, names = ds["meta"]["k"], ds["names"]
    datas = [ds["files"][n] for n in names]
    ctx.begin(k, len(names))
    nu, nt = ctx.count_kmers_batch(0, datas[:30], 3)
    nu2, nt2 = ctx.count_kmers_batch(30, datas[30:], 8)
    for i, nm in enumerate(names):
        m = ds["meta"]["lists"][nm]
        assert ((nu + nu2)[i], (nt + nt2)[i]) == (m["n_unique"], m["n_total"])
        w, f = ctx.get_list(i, m["n_unique"])
        assert hashlib.sha256(oracle.list_bytes(k, w, f)).hexdigest() == m["sha256"]
    assert ctx.build_presence() == ds["meta"]["n_union"]
    from phenotypeseeker_amd._lib import PskError
    with pytest.raises(PskError):
        ctx.count_kmers_batch(40, datas[:10], 2)   # runs past the declared sample count


@pytest.mark.parametrize("k,slab", [(13, False), (12, True), (16, False), (15, True)])
def test_grouped_counting_chains_equal_the_oracle_and_the_one_sample_chains(ctx, oracle, monkeypatch, k, slab):
    """The genomes of a batch go through the counting kernels in groups (r03: a sample dimension in the kernels of the dense
    chain, k <= 13, and of the bucketed sort, k = 14..16): groups of 8, of 3 and the one-sample chain (PSK_DC_GROUP) give the
    lists of the oracle bit for bit -- with a ragged last group, an empty sample and a windowless one in the middle, a FASTQ
    sample, MinHash sketches riding along, and under a slab filter."""
    from phenotypeseeker_amd.dist import slab_bounds
    from phenotypeseeker_amd.synth import GenomeSet, fastq_reads
    n_gen = 21
    gs = GenomeSet(n_gen, 150_000 if k >= 14 else 60_000, seed=100 + k)
    datas = [gs.sample(i)[1] for i in range(n_gen)]
    datas[5] = b""
    datas[11] = b">tiny\nACGTACGTAC\n"
    datas[17] = fastq_reads(gs.codes(3), 400, 150, seed=[k, 2])
    lo, hi = slab_bounds(k, 3, 1) if slab else (0, 0)
    ref = []
    for d in datas:
        w, f = oracle.count_kmers(d, k)[:2]
        sel = (w >= lo) & ((w < hi) if hi else np.ones(len(w), bool))
        ref.append((w[sel], f[sel]))
    sk = None
    for group in ("8", "3", "1"):
        monkeypatch.setenv("PSK_DC_GROUP", group)
        ctx.begin(k, len(datas), lo, hi)
        out = ctx.count_kmers_batch(0, datas, 4, sketch=(21, 64, 42))
        nu, nt = out[0], out[1]
        for i, (rw, rf) in enumerate(ref):
            w, f = ctx.get_list(i, nu[i])
            assert (nu[i], nt[i]) == (len(rw), int(rf.sum())), (group, i)
            assert np.array_equal(w, rw) and np.array_equal(f, rf), (group, i)
        if sk is None:
            sk = [np.array(h) for h in out[2]]
        else:
            assert all(np.array_equal(a, np.array(b)) for a, b in zip(sk, out[2])), group
    monkeypatch.delenv("PSK_DC_GROUP")


def test_batch_counting_with_slab_filter(ctx, oracle):
    """The multi-GPU ingest path: psk_count_kmers_batch under a slab filter (the kept-word count only exists
    on the device; launches are sized by the host's window count).  Slabs concatenate to the full lists,
    including samples that keep nothing, an empty sample and a FASTQ one."""
    from phenotypeseeker_amd.dist import slab_bounds
    from phenotypeseeker_amd.synth import GenomeSet, fastq_reads
    k = 13
    gs = GenomeSet(5, 60_000, seed=17)
    datas = [gs.sample(i)[1] for i in range(5)] + [b"", b">x\nACGTACGTACGTACGTACGTAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA\n",
                                                   fastq_reads(gs.codes(0), 300, 150, seed=[5, 1])]
    ref = [oracle.count_kmers(d, k)[:2] for d in datas]
    world = 4
    got_w = [[] for _ in datas]
    got_f = [[] for _ in datas]
    for rank in range(world):
        lo, hi = slab_bounds(k, world, rank)
        ctx.begin(k, len(datas), lo, hi)
        nu, nt = ctx.count_kmers_batch(0, datas, 3)
        for i in range(len(datas)):
            w, f = ctx.get_list(i, nu[i])
            assert int(f.sum()) == nt[i]
            if len(w):
                assert w.min() >= lo and (hi == 0 or w.max() < hi)
            got_w[i].append(w)
            got_f[i].append(f)
    for i, (rw, rf) in enumerate(ref):
        assert np.array_equal(np.concatenate(got_w[i]), rw), i
        assert np.array_equal(np.concatenate(got_f[i]), rf), i


def test_list_split_copy_and_install_reproduce_the_slab_lists(ctx, oracle):
    """The multi-GPU ingest's three calls (psk_lists_split / psk_copy_list_ranges / psk_set_lists_device): lists
    counted without a slab filter, cut at the slab bounds and installed into a slab context, are the lists -- and give
    the union and matrix -- that counting with the slab filter gives; a list that is not ascending inside the slab
    is refused."""
    import ctypes
    from phenotypeseeker_amd import dist as D
    from phenotypeseeker_amd.engine import PskContext
    from phenotypeseeker_amd.synth import GenomeSet
    hip = ctypes.CDLL("libamdhip64.so")
    k, n, world = 9, 6, 3
    gs = GenomeSet(n, 30_000, seed=5, gene_len=300)
    datas = [gs.sample(i)[1] for i in range(n)] + []
    datas[4] = b""                                          # an empty sample
    ctx.begin(k, n)
    nu, nt = ctx.count_kmers_batch(0, datas, 2)
    bounds = [D.slab_bounds(k, world, d)[0] for d in range(world)] + [0]
    cuts = ctx.lists_split(0, n, bounds)
    full = [ctx.get_list(i, nu[i]) for i in range(n)]
    for i in range(n):
        want = [int(np.searchsorted(full[i][0], np.uint64(b))) for b in bounds[:-1]] + [nu[i]]
        assert cuts[i].tolist() == want, i
    dw, df = ctypes.c_void_p(), ctypes.c_void_p()
    cap = max(sum(nu), 1)
    assert hip.hipMalloc(ctypes.byref(dw), ctypes.c_size_t(cap * 8)) == 0
    assert hip.hipMalloc(ctypes.byref(df), ctypes.c_size_t(cap * 4)) == 0
    try:
        with PskContext(0) as slab, PskContext(0) as ref:
            for d in range(world):
                lo, hi = D.slab_bounds(k, world, d)
                slab.begin(k, n, lo, hi)
                ref.begin(k, n, lo, hi)
                rnu, _ = ref.count_kmers_batch(0, datas, 2)
                cnt = [int(cuts[i, d + 1] - cuts[i, d]) for i in range(n)]
                order = [3, 0, 5, 1, 4, 2]                       # any order, as long as both calls agree
                ctx.copy_list_ranges(order, [int(cuts[i, d]) for i in order], [cnt[i] for i in order], dw.value, df.value)
                slab.set_lists_device(order, [cnt[i] for i in order], [nt[i] for i in order], dw.value, df.value)
                for i in range(n):
                    assert cnt[i] == rnu[i]
                    a, b = slab.get_list(i, cnt[i]), ref.get_list(i, cnt[i])
                    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (d, i)
                m1, m2 = slab.build_presence(), ref.build_presence()
                assert m1 == m2
                if m1:
                    assert np.array_equal(slab.get_union(), ref.get_union())
                    rows = np.arange(m1, dtype=np.uint64)
                    assert np.array_equal(slab.get_rows(rows), ref.get_rows(rows))
            # misuse: words outside the slab / not ascending
            lo, hi = D.slab_bounds(k, world, 1)
            slab.begin(k, n, lo, hi)
            ctx.copy_list_ranges([0], [0], [nu[0]], dw.value, df.value)    # the whole list: starts below the slab
            with pytest.raises(RuntimeError):
                slab.set_lists_device([0], [nu[0]], [0], dw.value, df.value)
            with pytest.raises(RuntimeError):
                slab.get_list(0, 0)                                        # ... and nothing of it was kept
            with pytest.raises(RuntimeError):
                ctx.copy_list_ranges([0], [nu[0]], [1], dw.value, df.value)   # beyond the end
    finally:
        hip.hipFree(dw)
        hip.hipFree(df)


def test_fastq_reads_and_gzip_input(ctx, oracle, tmp_path):
    """cfg-5 shape at test size: raw reads as FASTQ (constant quality line 'I...'), gzip-compressed on
    disk; the file reader inflates on the host, the list must equal the oracle's on the inflated bytes."""
    import gzip as _gz
    from phenotypeseeker_amd import formats
    from phenotypeseeker_amd.synth import GenomeSet, fastq_reads
    gs = GenomeSet(1, 300_000, seed=8)
    fq = fastq_reads(gs.codes(0), n_reads=20_000, read_len=150, seed=3)
    path = os.path.join(tmp_path, "reads.fastq.gz")
    with _gz.open(path, "wb", compresslevel=1) as f:
        f.write(fq)
    data = formats.read_sequence_file(path)
    assert data == fq
    for k in (13, 21):
        ow, of, ont = oracle.count_kmers(fq, k)
        assert ont == 20_000 * (150 - k + 1)
        ctx.begin(k, 1)
        nu, nt = ctx.count_kmers(0, data)
        w, f = ctx.get_list(0, nu)
        assert nt == ont and np.array_equal(w, ow) and np.array_equal(f, of)
    assert int(of.max()) > 3   # coverage 10x: counts well above 1 are exercised


def test_randomised_pipeline_against_oracle():
    """A few seconds of tests/_stress.py: random sample sets, k and slabs through batch counting, presence build
    (tiled and sort routes) and the chi2 scan, everything compared with the oracle for equality."""
    import subprocess
    import sys
    from helpers import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_stress.py"), "8", "7"], cwd=ROOT, timeout=600,
                       capture_output=True, text=True)
    assert r.returncode == 0 and "stress ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_counting_from_files_equals_counting_from_memory(ctx, tmp_path):
    """psk_count_kmers_files (the framing threads read the files) against psk_count_kmers_batch on the same bytes,
    with and without sketches; an unreadable path fails loudly."""
    from phenotypeseeker_amd._lib import PskError
    ds = load_dataset("ds_bonf")
    names = ds["names"][:12]
    paths = []
    for nm in names:
        p = os.path.join(tmp_path, nm + ".seq")
        with open(p, "wb") as f:
            f.write(ds["files"][nm])
        paths.append(p)
    empty = os.path.join(tmp_path, "empty.seq")
    open(empty, "wb").close()
    paths.append(empty)
    datas = [ds["files"][nm] for nm in names] + [b""]
    k = ds["meta"]["k"]
    ctx.begin(k, len(paths))
    nu, nt, sk = ctx.count_kmers_files(0, paths, 3, sketch=(21, 1000, 42))
    lists = [ctx.get_list(i, nu[i]) for i in range(len(paths))]
    ctx.begin(k, len(paths))
    nu2, nt2, sk2 = ctx.count_kmers_batch(0, datas, 3, sketch=(21, 1000, 42))
    assert (nu, nt) == (nu2, nt2)
    for i in range(len(paths)):
        w, f = ctx.get_list(i, nu2[i])
        assert np.array_equal(w, lists[i][0]) and np.array_equal(f, lists[i][1]) and np.array_equal(sk[i], sk2[i])
    assert ctx.count_kmers_files(0, paths[:3], 2)[0] == nu[:3]
    with pytest.raises(PskError, match="reading or framing"):
        ctx.count_kmers_files(0, [paths[0], os.path.join(tmp_path, "missing.fa")], 2)
