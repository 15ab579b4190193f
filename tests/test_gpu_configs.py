"""BASELINE.json configs 3, 4 and 5 exercised inside `pytest -m gpu`.  Config 2 has its
full-size tests in test_gpu_parity.py, config 1 its recipe (tools/cfg1_repro.py, test_gpu_e2e.py::test_cfg1_example_dataset).

Full-size workloads cannot be replayed through the oracle, so they are checked through size-independent properties and
a CPU re-derivation of a sample of rows; the workloads that fit (config 4 at 256 genomes) are compared with the oracle
row by row."""
import os

import numpy as np
import pytest

from helpers import read_results_tsv

pytestmark = pytest.mark.gpu


def _run(tmp, argv):
    from phenotypeseeker_amd.cli import build_parser
    os.chdir(tmp)
    args = build_parser().parse_args(argv)
    args.func(args)


def test_cfg4_continuous_weighted_cli_run_matches_the_oracle(tmp_path, oracle):
    """Config 4 as a workload: `modeling -w` on a continuous phenotype, 256 genomes (GSC weights from GPU MinHash
    sketches, non-integer) -- the weighted Welch rows the CLI writes against oracle.ttest_scan run with the run's own
    weights at 1e-8, and the Lasso model file."""
    import joblib
    from phenotypeseeker_amd import modeling as M
    from phenotypeseeker_amd.synth import GenomeSet
    n, L, k = 256, 60_000, 13
    gs = GenomeSet(n, L, seed=404, gene_len=400, sub_rate=0.004)
    rows, pheno = ["ID\tAddresses\tMIC"], []
    os.chdir(tmp_path)
    for i in range(n):
        name, fa = gs.sample(i)
        with open(name + ".fasta", "wb") as f:
            f.write(fa)
        v = "NA" if i in (7, 100) else repr(round(gs.continuous_phenotype(i), 4))
        pheno.append(v)
        rows.append("%s\t%s.fasta\t%s" % (name, name, v))
    with open("data.pheno", "w") as f:
        f.write("\n".join(rows) + "\n")
    _run(tmp_path, ["modeling", "data.pheno", "-w", "--pvalue", "0.05"])
    names = [gs.name(i) for i in range(n)]
    w = np.array([M.Input.samples[nm].weight for nm in names], dtype=np.float64)
    assert w.sum() == pytest.approx(n) and w.max() > 1.02 and w.min() < 0.98 and len(np.unique(np.round(w, 6))) > 50
    # the weights themselves (they were only ever fed back to the oracle): the run's sketches -> distances
    # (the formats of distances.mat, byte for byte what the run left behind) -> the ORACLE's neighbour joining and newick text
    # (orc_nj, nothing of the product in it) -> the GSC recursion that tests/golden/gsc_kat.json pins to the reference
    from oracle import oracle_weights as OW
    from phenotypeseeker_amd import weights as W
    labels, mat = W.distance_matrix(names, {nm: M.Input.samples[nm].sketch for nm in names})
    assert open("distances.mat").read() == W.distances_mat_text(labels, mat)
    newick = OW.nj_newick(labels, np.asarray(mat, dtype=np.float64))
    assert open("tree_newick.txt").read() == newick + "\n"
    want_w = W.gsc_weights(W.from_newick(newick))
    assert w.tolist() == [want_w[nm] for nm in names]
    head, got = read_results_tsv("t-test_results_MIC.tsv")
    assert head[:3] == ["k-mer", "t-test", "p-value"] and len(got) > 100
    wl = [oracle.count_kmers(gs.sample(i)[1], k)[0] for i in range(n)]
    uw = oracle.union(wl)
    bits = oracle.presence_bits(wl, uw)
    ph = [("NA" if p == "NA" else float(p)) for p in pheno]
    ref = oracle.ttest_scan(bits, ph, w, n, 2, n - 2, 0.05, len(uw))
    keep = np.nonzero(ref["keep"])[0]
    want = {oracle.word_to_kmer(uw[r], k): r for r in keep}
    got_k = {g[0] for g in got}
    # the second kernel of the Welch scan re-sums every candidate in the reference's sample order (r03): the SAME set of
    # rows, and every printed field -- round(t, 2), "%.2E" % p, the two rounded means, the count -- string-identical
    assert got_k == set(want)
    differing = 0
    for g in got:
        r = want[g[0]]
        exp = [str(oracle.round2(ref["stat"][r])), "%.2E" % ref["p"][r], str(oracle.round2(ref["mean_x"][r])),
               str(oracle.round2(ref["mean_y"][r])), str(int(ref["n_with"][r]))]
        differing += sum(a != b for a, b in zip(g[1:6], exp))
    assert differing == 0
    # ... and the unrounded statistics through the library, at 1e-8
    from phenotypeseeker_amd.engine import PskContext
    with PskContext(0) as ctx:
        ctx.begin(k, n)
        ctx.count_kmers_batch(0, [gs.sample(i)[1] for i in range(n)], 4)
        m = ctx.build_presence()
        assert m == len(uw)
        vals = np.array([0.0 if p == "NA" else float(p) for p in ph])
        valid = np.array([p != "NA" for p in ph], dtype=np.uint8)
        npass = ctx.ttest_scan(vals, valid, w, 2, n - 2, 0.05, m)
        res = ctx.get_results(npass)
    assert np.array_equal(res["row"].astype(np.int64), keep)
    for key in ("stat", "mean_x", "mean_y", "n_with"):
        assert np.array_equal(res[key], ref[key][keep]), key
    assert np.allclose(res["p"], ref["p"][keep], rtol=1e-10, atol=1e-300)
    pkg = joblib.load("linreg_model_MIC.pkl")
    assert pkg["pred_scale"] == "continuous" and type(pkg["model"].best_estimator_).__name__ == "Lasso"


def test_cfg4_full_size_cli_run_properties(tmp_path, oracle):
    """Config 4 at FULL size through the CLI: 1,024 genomes x 5 Mbp from FASTA files on disk, continuous phenotype,
    `modeling -w` (GPU MinHash sketches -> Mash distances -> NJ -> GSC weights -> weighted Welch scan -> Lasso).  The
    oracle cannot replay 1,024 x 5 Mbp, so: the weights sum to N and vary; the result file's rows are re-derived -- the
    library scans the same files with the run's weights, 200 of its survivors go through oracle.ttest_scan row by row
    (t, means, count equal bit for bit; p at 1e-10) and every one of those rows is in the CLI's TSV with the oracle's
    printed strings; rows the oracle rejects are absent; the model file is a Lasso GridSearchCV."""
    import joblib
    from phenotypeseeker_amd import modeling as M
    from phenotypeseeker_amd.engine import PskContext
    from phenotypeseeker_amd.synth import GenomeSet
    n, L, k = 1024, 5_000_000, 13
    gs = GenomeSet(n, L, seed=4242)
    rng = np.random.default_rng(7)
    os.chdir(tmp_path)
    rows, pheno, paths = ["ID\tAddresses\tMIC"], [], []
    for i in range(n):
        name, fa = gs.sample(i)
        paths.append(os.path.join(tmp_path, name + ".fasta"))
        with open(paths[-1], "wb") as f:
            f.write(fa)
        v = "NA" if i in (11, 500) else "%.4f" % (2.0 * gs.phenotype(i) + rng.normal(0, 0.5))
        pheno.append(v)
        rows.append("%s\t%s.fasta\t%s" % (name, name, v))
    with open("data.pheno", "w") as f:
        f.write("\n".join(rows) + "\n")
    _run(tmp_path, ["modeling", "data.pheno", "-w"])
    names = [gs.name(i) for i in range(n)]
    w = np.array([M.Input.samples[nm].weight for nm in names], dtype=np.float64)
    assert w.sum() == pytest.approx(n, rel=1e-9) and w.min() > 0 and len(np.unique(np.round(w, 6))) > 100
    # the weights themselves: the matrix the run left in distances.mat -> the ORACLE's neighbour joining (orc_nj: 1,024 leaves
    # in half a second) and newick text -> the GSC recursion pinned to the reference (tests/golden/gsc_kat.json)
    from oracle import oracle_weights as OW
    from phenotypeseeker_amd import weights as W
    dm = [l.split("\t") for l in open("distances.mat").read().split("\n")]
    assert [r[0] for r in dm] == names and all(len(r) == n + 1 for r in dm)
    mat = np.array([[float(x) for x in r[1:]] for r in dm])
    assert np.array_equal(mat, mat.T) and (np.diag(mat) == 0).all()
    newick = OW.nj_newick(names, mat)
    assert open("tree_newick.txt").read() == newick + "\n"
    want_w = W.gsc_weights(W.from_newick(newick))
    assert w.tolist() == [want_w[nm] for nm in names]
    head, got = read_results_tsv("t-test_results_MIC.tsv")
    assert head[:3] == ["k-mer", "t-test", "p-value"] and len(got) >= 500          # the 2-kbp gene's k-mers at least
    printed = {g[0]: g[1:6] for g in got}
    vals = np.array([0.0 if p == "NA" else float(p) for p in pheno])
    valid = np.array([p != "NA" for p in pheno], dtype=np.uint8)
    with PskContext(0) as ctx:
        ctx.begin(k, n)
        for s0 in range(0, n, 64):
            ctx.count_kmers_files(s0, paths[s0:s0 + 64], 8)
        m = ctx.build_presence()
        assert 30_000_000 < m <= 4 ** k // 2 + 2 ** k
        uw = ctx.get_union()
        npass = ctx.ttest_scan(vals, valid, w, 2, n - 2, 0.05, m)
        res = ctx.get_results(npass)
        assert npass == len(got) and np.all(res["row"][1:] > res["row"][:-1])
        pick = np.unique(np.linspace(0, npass - 1, 200).astype(np.int64))
        sel_rows = ctx.get_rows(res["row"][pick])
        # ... and 300 rows the scan rejected (every 100,003rd row of the matrix that is not a survivor)
        others = np.setdiff1d(np.arange(0, m, 100_003, dtype=np.uint64), res["row"])[:300]
        other_rows = ctx.get_rows(others)
    ph = [("NA" if p == "NA" else float(p)) for p in pheno]
    ref = oracle.ttest_scan(sel_rows, ph, w, n, 2, n - 2, 0.05, m)
    assert ref["keep"].all()
    for key in ("stat", "mean_x", "mean_y", "n_with"):
        assert np.array_equal(res[key][pick], ref[key]), key
    assert np.allclose(res["p"][pick], ref["p"], rtol=1e-10, atol=1e-300)
    differing = 0
    for j, r in enumerate(pick):
        g = printed[oracle.word_to_kmer(uw[int(res["row"][r])], k)]
        exp = [str(oracle.round2(ref["stat"][j])), "%.2E" % ref["p"][j], str(oracle.round2(ref["mean_x"][j])),
               str(oracle.round2(ref["mean_y"][j])), str(int(ref["n_with"][j]))]
        differing += sum(a != b for a, b in zip(g, exp))
    assert differing == 0
    assert not oracle.ttest_scan(other_rows, ph, w, n, 2, n - 2, 0.05, m)["keep"].any()
    pkg = joblib.load("linreg_model_MIC.pkl")
    assert pkg["pred_scale"] == "continuous" and type(pkg["model"].best_estimator_).__name__ == "Lasso"


def _popcount_columns(rows, n):
    """column sums of a bit matrix [m][wpr] u64 -> per-sample counts"""
    b = np.unpackbits(np.ascontiguousarray(rows, dtype="<u8").view(np.uint8), axis=1, bitorder="little")[:, :n]
    return b.sum(axis=0, dtype=np.int64)


def test_cfg3_full_size_slab_properties(oracle):
    """Config 3, one rank's share at full size: 2,048 x 5 Mbp, k = 16, slab 0 of 8 of the balanced (quantile-cut) word
    space.  Column sums of the matrix = list lengths inside the slab, rows strictly ascending and inside the slab, the
    scan is idempotent and the two forms of the kernel agree, and 200 survivors re-derived on the CPU from their rows."""
    from phenotypeseeker_amd import dist
    from phenotypeseeker_amd.engine import PskContext
    from phenotypeseeker_amd.synth import GenomeSet
    n, L, k, world = 2048, 5_000_000, 16, 8
    gs = GenomeSet(n, L, seed=12345)
    with PskContext(0) as ctx:
        ctx.begin(k, 1)
        nu0, _ = ctx.count_kmers(0, gs.sample(0)[1])
        pilot = ctx.get_list(0, nu0)[0]
        bounds = dist.quantile_bounds(dist.pilot_points(pilot), k, world)
        inside = int(np.searchsorted(pilot, np.uint64(bounds[1])))
        assert abs(inside / len(pilot) - 1.0 / world) < 0.01          # the cut is where an eighth of the list ends
        lo, hi = bounds[0], bounds[1]
        ctx.begin(k, n, lo, hi)
        nus = []
        for s0 in range(0, n, 64):
            nu, nt = ctx.count_kmers_batch(s0, [gs.sample(i)[1] for i in range(s0, min(s0 + 64, n))], 8)
            nus += nu
        m = ctx.build_presence()
        _, wpr, _ = ctx.presence_shape()
        assert wpr == 32 and 20_000_000 < m < 30_000_000      # an eighth of the 1.9 x 10^8 union rows (the uniform first slab holds 4.4 x 10^7)
        uw = ctx.get_union()
        assert np.all(uw[1:] > uw[:-1]) and uw[0] >= lo and uw[-1] < hi
        # column sums over a sixteenth of the rows at a time (the whole matrix is 11 GB)
        sums = np.zeros(n, dtype=np.int64)
        step = 1 << 20
        for r0 in range(0, m, step):
            rows = ctx.get_rows(np.arange(r0, min(r0 + step, m), dtype=np.uint64))
            assert rows.any(axis=1).all()
            sums += _popcount_columns(rows, n)
        assert sums.tolist() == list(nus)
        w0, _ = ctx.get_list(5, nus[5])
        ow = oracle.count_kmers(gs.sample(5)[1], k)[0]
        assert np.array_equal(w0, ow[(ow >= lo) & (ow < hi)])
        pheno = np.array([gs.phenotype(i) for i in range(n)], dtype=np.int8)
        m_global = 8 * m
        a = ctx.chi2_scan(pheno, None, 2, n - 2, 0.05, False, m_global)
        ra = ctx.get_results(a)
        b = ctx.chi2_scan(pheno, None, 2, n - 2, 0.05, False, m_global)
        rb = ctx.get_results(b)
        assert a == b and a >= 100 and all(np.array_equal(ra[key], rb[key]) for key in ra)
        os.environ["PSK_CHI2_MODE"] = "2"
        try:
            c = ctx.chi2_scan(pheno, None, 2, n - 2, 0.05, False, m_global)
            rc = ctx.get_results(c)
        finally:
            del os.environ["PSK_CHI2_MODE"]
        assert c == a and all(np.array_equal(ra[key], rc[key]) for key in ra)
        pick = np.unique(np.linspace(0, a - 1, 200).astype(np.int64))
        rows = ctx.get_rows(ra["row"][pick])
        ref = oracle.chi2_scan(rows, pheno.tolist(), np.ones(n), n, 2, n - 2, 0.05, False, m_global)
        assert ref["keep"].all()
        # p = exp(-1421 / 2) = 2.6e-309 is a subnormal double here: equal to the oracle's libm value to the bits a
        # subnormal has, compared at 1e-10
        assert np.array_equal(ref["stat"], ra["stat"][pick]) and np.allclose(ref["p"], ra["p"][pick], rtol=1e-10, atol=0)
        assert np.array_equal(ref["n_with"], ra["n_with"][pick])
        assert np.array_equal(ra["word"][pick], uw[ra["row"][pick].astype(np.int64)])


def _fastq_sample(codes, n_reads, read_len, seed, err=0.005):
    """cfg-5 reads with fixed-width names, built without a Python loop over the reads."""
    rng = np.random.default_rng(seed)
    L = len(codes)
    starts = rng.integers(0, L - read_len, n_reads)
    reads = codes[starts[:, None] + np.arange(read_len)[None, :]]
    mask = rng.random(reads.shape) < err
    reads[mask] = rng.integers(0, 4, int(mask.sum()), dtype=np.uint8)
    rec = np.empty((n_reads, 10 + read_len + 3 + read_len + 1), dtype=np.uint8)
    digits = np.array(list(b"0123456789"), dtype=np.uint8)
    idx = np.arange(n_reads)
    rec[:, 0] = ord("@")
    rec[:, 1] = ord("r")
    for d in range(7):
        rec[:, 2 + d] = digits[(idx // 10 ** (6 - d)) % 10]
    rec[:, 9] = 10
    rec[:, 10:10 + read_len] = np.frombuffer(b"ACGT", dtype=np.uint8)[reads]
    rec[:, 10 + read_len] = 10
    rec[:, 11 + read_len] = ord("+")
    rec[:, 12 + read_len] = 10
    rec[:, 13 + read_len:13 + 2 * read_len] = ord("I")
    rec[:, -1] = 10
    return rec.tobytes()


def test_cfg5_full_size_fastq_samples_properties(tmp_path, oracle):
    """Config 5 at full sample size: 8 samples x 2 M 150-bp reads (0.64 GB of FASTQ each, 276 M windows) through
    psk_count_kmers_files, then the presence matrix and a scan.  Window counts, strictly ascending lists whose counts
    add up, column sums = list lengths, the gene's k-mers found by the scan, a read prefix equal to the oracle."""
    from phenotypeseeker_amd.engine import PskContext
    from phenotypeseeker_amd.synth import GenomeSet
    n, reads, rl, k = 8, 2_000_000, 150, 13
    gs = GenomeSet(n, 5_000_000, seed=99)
    paths = []
    first = None
    for i in range(n):
        data = _fastq_sample(gs.codes(i), reads, rl, seed=[5, i])
        if i == 0:
            first = data[: 20000 * (13 + 2 * rl + 1)]
        paths.append(os.path.join(tmp_path, "s%d.fastq" % i))
        with open(paths[-1], "wb") as f:
            f.write(data)
        del data
    with PskContext(0) as ctx:
        ctx.begin(k, n)
        nu, nt = ctx.count_kmers_files(0, paths, 8)
        assert list(nt) == [reads * (rl - k + 1)] * n
        ow, of, _ = oracle.count_kmers(first, k)
        w0, f0 = ctx.get_list(0, nu[0])
        assert np.all(w0[1:] > w0[:-1]) and int(f0.astype(np.uint64).sum()) == nt[0]
        pos = np.searchsorted(w0, ow)
        assert np.array_equal(w0[pos], ow) and np.all(f0[pos] >= of)      # the prefix's words are there, at least as often
        m = ctx.build_presence()
        uw = ctx.get_union()
        assert m == len(uw) and np.all(uw[1:] > uw[:-1])
        sums = np.zeros(n, dtype=np.int64)
        for r0 in range(0, m, 1 << 21):
            sums += _popcount_columns(ctx.get_rows(np.arange(r0, min(r0 + (1 << 21), m), dtype=np.uint64)), n)
        assert sums.tolist() == list(nu)
        assert np.array_equal(ctx.lookup_counts(0, w0[::1001]), f0[::1001])
        pheno = np.array([gs.phenotype(i) for i in range(n)], dtype=np.int8)
        npass = ctx.chi2_scan(pheno, None, 2, n - 2, 0.05, True, m)
        res = ctx.get_results(npass)
        if npass:
            rows = ctx.get_rows(res["row"])
            ref = oracle.chi2_scan(rows, pheno.tolist(), np.ones(n), n, 2, n - 2, 0.05, True, m)
            assert ref["keep"].all() and np.array_equal(ref["stat"], res["stat"]) and np.array_equal(ref["n_with"], res["n_with"])
        some = ctx.get_rows(np.arange(0, m, 997, dtype=np.uint64))
        ref = oracle.chi2_scan(some, pheno.tolist(), np.ones(n), n, 2, n - 2, 0.05, True, m)
        kept = np.nonzero(ref["keep"])[0] * 997
        assert np.array_equal(np.intersect1d(res["row"].astype(np.int64), np.arange(0, m, 997)), kept)
        # reads at 60x cover the 2-kbp gene completely: its k-mers are rows, and they split the samples by carrier status
        gene_words = np.unique(oracle.count_kmers(b">g\n" + bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[gs.gene]) + b"\n", k)[0])
        carriers = sum(gs.has_gene(i) for i in range(n))
        gpos = np.searchsorted(uw, gene_words)
        assert np.array_equal(uw[gpos], gene_words)
        grow = ctx.get_rows(gpos.astype(np.uint64))
        assert (np.unpackbits(grow.view(np.uint8), axis=1, bitorder="little").sum(axis=1) >= carriers).all()


def test_cfg3_whole_dataset_on_one_gpu_equals_its_eight_slabs(oracle):
    """BASELINE config 3 at FULL size on one GPU (r03 ran it as a scratch script): 2,048 x 5 Mbp, k = 16 --
    10.2 G (word, sample) pairs, 188 M union rows, a 48-GB matrix -- and then the same dataset as the eight ranks of a node
    would hold it, slab by slab (bounds at the pilot's quantiles, every slab its own count / build / scan with the global
    Bonferroni denominator).  Size-independent properties: the slab unions are disjoint, ascending, and concatenate to the
    one-GPU union word for word; the pairs of the slabs add up to the pairs of the whole; the survivors of the eight slab
    scans, in slab order, are the survivors of the one-GPU scan -- same words, statistic, p, counts, bit for bit."""
    from phenotypeseeker_amd import dist
    from phenotypeseeker_amd.engine import PskContext
    from phenotypeseeker_amd.synth import GenomeSet
    n, L, k, world = 2048, 5_000_000, 16, 8
    gs = GenomeSet(n, L, seed=12345)
    fas = [gs.sample(i)[1] for i in range(n)]          # 10 GB of FASTA in host memory: nine passes over it
    pheno = np.array([gs.phenotype(i) for i in range(n)], dtype=np.int8)

    def run(ctx, lo, hi, m_global):
        ctx.begin(k, n, lo, hi)
        nus = []
        for s0 in range(0, n, 64):
            nu, _ = ctx.count_kmers_batch(s0, fas[s0:s0 + 64], 8)
            nus += nu
        m = ctx.build_presence()
        uw = ctx.get_union()
        assert len(uw) == m and np.all(uw[1:] > uw[:-1])
        res = None
        if m_global:
            c = ctx.chi2_scan(pheno, None, 2, n - 2, 0.05, False, m_global)
            res = ctx.get_results(c)
        return nus, uw, res

    with PskContext(0) as ctx:
        ctx.begin(k, 1)
        nu0, _ = ctx.count_kmers(0, fas[0])
        bounds = dist.quantile_bounds(dist.pilot_points(ctx.get_list(0, nu0)[0]), k, world)
        nus_all, uw_all, _ = run(ctx, 0, 0, 0)
        m_all = len(uw_all)
        assert 180_000_000 < m_all < 200_000_000 and ctx.presence_shape()[1] == 32
        c = ctx.chi2_scan(pheno, None, 2, n - 2, 0.05, False, m_all)
        res_all = ctx.get_results(c)
        assert c >= 1000
        keys = ("word", "stat", "p", "n_with")
        got = {key: [] for key in keys}
        pairs = np.zeros(n, dtype=np.int64)
        off = 0
        for s in range(world):
            nus, uw, res = run(ctx, bounds[s], bounds[s + 1], m_all)
            assert uw[0] >= bounds[s] and (bounds[s + 1] == 0 or uw[-1] < bounds[s + 1] or s == world - 1)
            assert np.array_equal(uw, uw_all[off:off + len(uw)]), s       # the slab's union IS that stretch of the whole
            assert abs(len(uw) / m_all - 1.0 / world) < 0.02              # balanced slabs
            off += len(uw)
            pairs += np.array(nus, dtype=np.int64)
            for key in keys:
                got[key].append(res[key])
        assert off == m_all and pairs.tolist() == list(nus_all)
        for key in keys:
            assert np.array_equal(np.concatenate(got[key]), res_all[key]), key


def test_cfg5_one_ranks_share_of_full_size_fastq_samples_through_the_list_exchange(oracle):
    """Config 5 at ONE RANK'S SHARE of the 8-GPU split (VERDICT r04 #6a; r03/r04 ran 32 samples counted in place): 64 of its 512
    samples at full size -- 2 M 150-bp reads, 0.64 GB of FASTQ each, generated and handed over four at a time (41 GB in all
    never sit in host memory at once) -- through the sample-parallel ingest of the sharded path: counted in a counting
    context, moved by dist.ListExchange over an RCCL communicator (one rank: ncclSend / ncclRecv to itself, the
    `--force-exchange` form) into the slab context, matrix built there.  Window counts, ascending lists whose counts add
    up, the exchanged lists equal to lists counted in place, column sums of the matrix = list lengths, the scan's
    survivors equal to the oracle's on a sample of the rows, the gene's k-mers present in every carrier."""
    import tempfile
    from phenotypeseeker_amd import dist
    from phenotypeseeker_amd.engine import PskContext, PskError
    from phenotypeseeker_amd.synth import GenomeSet
    n, reads, rl, k = 64, 2_000_000, 150, 13
    gs = GenomeSet(n, 5_000_000, seed=99)
    old = {v: os.environ.get(v) for v in ("PSK_RDZV_FILE", "PSK_DIST_TRANSPORT", "PSK_RDZV_DIR", "PSK_LAUNCH_NONCE")}
    os.environ["PSK_RDZV_FILE"] = os.path.join(tempfile.mkdtemp(prefix="psk_rdzv_"), "id")
    for v in ("PSK_DIST_TRANSPORT", "PSK_RDZV_DIR", "PSK_LAUNCH_NONCE"):
        os.environ.pop(v, None)
    g = dist.Group()
    g.world, g.rank, g.local_rank = 1, 0, 0
    try:
        g.init(force=True)
        assert g.backend == "rccl" and g.rccl_ranks == 1
        with PskContext(0) as cnt, PskContext(0) as ctx:
            cnt.begin(k, n)
            nu, nt = [], []
            for s0 in range(0, n, 4):
                batch = [_fastq_sample(gs.codes(i), reads, rl, seed=[5, i]) for i in range(s0, s0 + 4)]
                a, b = cnt.count_kmers_batch(s0, batch, 8)
                nu += list(a)
                nt += list(b)
                if s0 == 16:      # three samples counted in place as well, in the context the exchange fills
                    keep_batch = batch[1]
                del batch
            assert nt == [reads * (rl - k + 1)] * n
            ctx.begin(k, n)
            counted = {i: cnt.get_list(i, nu[i]) for i in (0, 17, 63)}     # (the exchange releases the counting context's lists)
            pairs = dist.ListExchange(g, k).run(cnt, ctx, n, nt)
            assert pairs == sum(nu)
            with pytest.raises(PskError, match="has not been counted"):
                cnt.get_list(0, nu[0])
            for i in (0, 17, 63):
                w0, f0 = ctx.get_list(i, nu[i])
                assert np.all(w0[1:] > w0[:-1]) and int(f0.astype(np.uint64).sum()) == nt[i]
                w1, f1 = counted[i]
                assert np.array_equal(w0, w1) and np.array_equal(f0, f1), i
            with PskContext(0) as solo:          # sample 17 counted in place: the list the exchange delivered
                solo.begin(k, 1)
                a, b = solo.count_kmers_batch(0, [keep_batch], 2)
                ws, fs = solo.get_list(0, a[0])
                w0, f0 = ctx.get_list(17, nu[17])
                assert a[0] == nu[17] and np.array_equal(ws, w0) and np.array_equal(fs, f0)
            del keep_batch
            m = ctx.build_presence()
            uw = ctx.get_union()
            assert m == len(uw) and np.all(uw[1:] > uw[:-1])
            sums = np.zeros(n, dtype=np.int64)
            for r0 in range(0, m, 1 << 21):
                sums += _popcount_columns(ctx.get_rows(np.arange(r0, min(r0 + (1 << 21), m), dtype=np.uint64)), n)
            assert sums.tolist() == list(nu)
            pheno = np.array([gs.phenotype(i) for i in range(n)], dtype=np.int8)
            npass = ctx.chi2_scan(pheno, None, 2, n - 2, 0.05, True, m)
            res = ctx.get_results(npass)
            some = ctx.get_rows(np.arange(0, m, 499, dtype=np.uint64))
            ref = oracle.chi2_scan(some, pheno.tolist(), np.ones(n), n, 2, n - 2, 0.05, True, m)
            kept = np.nonzero(ref["keep"])[0] * 499
            assert np.array_equal(np.intersect1d(res["row"].astype(np.int64), np.arange(0, m, 499)), kept)
            gene_words = np.unique(oracle.count_kmers(b">g\n" + bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[gs.gene]) + b"\n", k)[0])
            carriers = sum(gs.has_gene(i) for i in range(n))
            gpos = np.searchsorted(uw, gene_words)
            assert np.array_equal(uw[gpos], gene_words)
            grow = ctx.get_rows(gpos.astype(np.uint64))
            assert (np.unpackbits(grow.view(np.uint8), axis=1, bitorder="little").sum(axis=1) >= carriers).all()
    finally:
        g.close()
        for v, val in old.items():
            if val is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = val
