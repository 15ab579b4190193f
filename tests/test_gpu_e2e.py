"""GPU end-to-end tests (-m gpu): `phenotypeseeker modeling` and `prediction` on the golden
datasets, through the CLI parser and the C ABI, against the files the reference wrote."""
import csv
import os

import numpy as np
import pytest

from helpers import load_dataset, read_results_tsv
from test_host_modeling import _write_dataset

pytestmark = pytest.mark.gpu


def _run(tmp, argv):
    from phenotypeseeker_amd.cli import build_parser
    os.chdir(tmp)
    args = build_parser().parse_args(argv)
    args.func(args)


@pytest.mark.parametrize("tag,extra", [("ds_omitB", ["--omit_B_correction", "--n_kmers", "100"]), ("ds_bonf", []), ("ds_k21", ["-l", "21"])])
def test_modeling_and_prediction_end_to_end(tmp_path, tag, extra):
    import joblib
    ds = load_dataset(tag)
    _write_dataset(ds, str(tmp_path))
    _run(tmp_path, ["modeling", "data.pheno"] + extra)
    head, ref = read_results_tsv(os.path.join(ds["dir"], "chi2_results_Pheno.tsv"))
    head2, got = read_results_tsv("chi2_results_Pheno.tsv")
    assert head2 == head and sorted(got) == sorted(ref)
    assert [g[2] for g in got] == [r[2] for r in ref]
    limit = 100 if "--n_kmers" in extra else 1000
    _, ref_top = read_results_tsv(os.path.join(ds["dir"], "chi2_results_Pheno_top%d.tsv" % limit))
    _, got_top = read_results_tsv("chi2_results_Pheno_top%d.tsv" % limit)
    assert [g[2] for g in got_top] == [r[2] for r in ref_top]
    with open(os.path.join(ds["dir"], "Pheno_MLdf.csv")) as f:
        ref_csv = list(csv.reader(f))
    with open("Pheno_MLdf.csv") as f:
        got_csv = list(csv.reader(f))
    assert [r[0] for r in got_csv] == [r[0] for r in ref_csv] and [r[-2:] for r in got_csv] == [r[-2:] for r in ref_csv]
    if "--n_kmers" not in extra:
        assert sorted(got_csv[0][1:-2]) == sorted(ref_csv[0][1:-2])
    # model artefacts
    pkg = joblib.load("log_reg_model_Pheno.pkl")
    assert set(pkg) >= {"model", "kmers", "pca", "pred_scale"} and pkg["pred_scale"] == "binary"
    kmers = list(pkg["kmers"])
    assert kmers == got_csv[0][1:-2]
    X = np.array([[int(v) for v in r[1:-2]] for r in got_csv[1:]], dtype=np.float64)
    y = np.array([int(r[-1]) for r in got_csv[1:]])
    model = pkg["model"]
    assert model.best_params_["C"] in [1.0 / a for a in np.logspace(-3, 3, 13)]
    assert (model.predict(X) == y).mean() >= 0.9      # the planted gene separates the classes
    lines = open("k-mers_and_coefficients_in_log_reg_model_Pheno.txt").read().splitlines()
    assert lines[0] == "K-mer\tcoef._in_log_reg_model\tNo._of_samples_with_k-mer\tSamples_with_k-mer"
    assert [l.split("\t")[0] for l in lines[1:]] == kmers
    assert np.allclose([float(l.split("\t")[1]) for l in lines[1:]], model.best_estimator_.coef_[0])
    summary = open("summary_of_log_reg_analysis_Pheno.txt").read()
    for needle in ("Parameters:\nLogisticRegression(max_iter=1000, penalty='l1', solver='liblinear')",
                   "Grid scores (mean accuracy) on development set:", "Best parameters found on development set:",
                   "Model predictions on samples:", "Classification report:", "Confusion matrix:",
                   "### Outputting the model to a model file! ###"):
        assert needle in summary, needle
    assert "Func" in open("log.txt").read()

    # --jump_to modelling re-fits from <pheno>_MLdf.csv and must reproduce the model
    coef0 = model.best_estimator_.coef_.copy()
    _run(tmp_path, ["modeling", "data.pheno", "-jt", "modelling"] + extra)
    pkg2 = joblib.load("log_reg_model_Pheno.pkl")
    assert np.allclose(pkg2["model"].best_estimator_.coef_, coef0, atol=1e-9)
    assert list(pkg2["kmers"]) == kmers

    # prediction: same samples through psk_count_dict + the stored model
    with open("samples.txt", "w") as f:
        for line in open("data.pheno").read().splitlines()[1:]:
            f.write("\t".join(line.split()[:2]) + "\n")
    with open("phenos.txt", "w") as f:
        f.write("Pheno\tlog_reg_model_Pheno.pkl\n")
    _run(tmp_path, ["prediction", "samples.txt", "phenos.txt"])
    out = open("predictions_Pheno.txt").read().splitlines()
    assert out[0] == "Sample_ID\tpredicted_phenotype\tprobability_for_predicted_class"
    assert [l.split("\t")[0] for l in out[1:]] == ds["names"]
    non_na = {r[0]: i for i, r in enumerate(got_csv[1:])}
    pred = model.predict(X)
    proba = model.predict_proba(X)
    for l in out[1:]:
        name, p, pr = l.split("\t")
        if name in non_na:
            assert int(p) == pred[non_na[name]]
            assert pr == str(round(proba[non_na[name]][1], 2))

    # the .pkl is the reference's: scikit-learn objects behind a plain joblib.load (prediction.py:124-129), usable in
    # a process that cannot import this package at all
    import subprocess
    import sys
    np.save("X.npy", X)
    code = ("import sys, joblib, numpy as np\n"
            "sys.path = [p for p in sys.path if 'repo' not in p and p not in ('', '.')]\n"
            "pkg = joblib.load('log_reg_model_Pheno.pkl')\n"
            "assert 'phenotypeseeker_amd' not in sys.modules\n"
            "m = pkg['model']\n"
            "assert type(m).__module__.startswith('sklearn.') and type(m.best_estimator_).__name__ == 'LogisticRegression'\n"
            "X = np.load('X.npy')\n"
            "np.save('pred.npy', m.predict(X)); np.save('proba.npy', m.predict_proba(X))\n"
            "print(len(m.cv_results_['params']), m.best_params_['C'], list(pkg['kmers'])[:1])\n")
    env = {k_: v for k_, v in os.environ.items() if k_ != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.load("pred.npy"), pred) and np.allclose(np.load("proba.npy"), proba, rtol=1e-12, atol=0)
    assert r.stdout.split()[0] == "13"


def test_at_rich_multi_contig_set_matches_the_reference(tmp_path, oracle):
    """every other golden set is uniform ACGT.  60 x 1 Mbp at 29 % GC (the reference's example
    organism, C. difficile), six contigs per genome with assembler-style headers: the genomes are regenerated from
    the parameters in meta.json (sha256 checked), the lists must hash to glistmaker's .list files, the union to
    glistcompare's, one mapping to glistquery's text, and `modeling` with and without --omit_B_correction must write
    the reference's result tables (unmodified modeling.py through oracle/ref_shim.py, 37-39 s per run on 8 cores)."""
    import gzip
    import hashlib
    import json
    from helpers import GOLDEN
    from phenotypeseeker_amd.engine import PskContext
    from phenotypeseeker_amd.synth import GenomeSet
    gd = os.path.join(GOLDEN, "ds_atrich")
    with open(os.path.join(gd, "meta.json")) as f:
        meta = json.load(f)
    gs = GenomeSet(**meta["synth"])
    k = meta["k"]
    names = []
    for i in range(gs.n):
        name, fa = gs.sample(i)
        names.append(name)
        assert hashlib.sha256(fa).hexdigest() == meta["inputs_sha256"][name]
        with open(os.path.join(tmp_path, name + ".fasta"), "wb") as f:
            f.write(fa)
    with PskContext(0) as ctx:
        ctx.begin(k, gs.n)
        nu, nt = ctx.count_kmers_files(0, [os.path.join(tmp_path, n + ".fasta") for n in names], 4)
        for i, n in enumerate(names):
            m = meta["lists"][n]
            assert (nu[i], nt[i]) == (m["n_unique"], m["n_total"]), n
            if i % 6 == 0:
                w, f = ctx.get_list(i, nu[i])
                assert hashlib.sha256(oracle.list_bytes(k, w, f)).hexdigest() == m["sha256"], n
        assert ctx.build_presence() == meta["n_union"]
        uw = ctx.get_union()
        assert hashlib.sha256(uw.tobytes()).hexdigest() == meta["union_words_sha256"]
        ms = names.index(meta["mapped_sample"])
        counts = ctx.lookup_counts(ms, uw)
        from phenotypeseeker_amd import formats
        kmers = formats.words_to_kmers(uw, k)
        txt = "".join("%s\t%d\n" % kc for kc in zip(kmers, counts.tolist()))
        assert hashlib.sha256(txt.encode()).hexdigest() == meta["mapped_sha256"]
        # summed frequencies of the union (glistcompare -u adds them)
        tot = np.zeros(len(uw), dtype=np.uint64)
        for i in range(gs.n):
            tot += ctx.lookup_counts(i, uw)
        assert hashlib.sha256(tot.astype(np.uint32).tobytes()).hexdigest() == meta["union_freqs_sha256"]
    import shutil
    shutil.copy(os.path.join(gd, "data.pheno"), os.path.join(tmp_path, "data.pheno"))
    for run, info in meta["runs"].items():
        for fn in os.listdir(tmp_path):
            if fn.startswith("chi2_results_") or fn.endswith("_MLdf.csv"):
                os.remove(os.path.join(tmp_path, fn))
        _run(tmp_path, ["modeling", "data.pheno"] + info["flags"])
        with gzip.open(os.path.join(gd, run, "chi2_results_Pheno.tsv.gz"), "rt") as f:
            ref = [tuple(l.rstrip("\n").split("\t")) for l in f]
        head2, got = read_results_tsv("chi2_results_Pheno.tsv")
        assert tuple(head2) == ref[0] and sorted(got) == sorted(ref[1:]), run
        assert [g[2] for g in got] == [r[2] for r in ref[1:]], run               # the p-string order
        with gzip.open(os.path.join(gd, run, "chi2_results_Pheno_top1000.tsv.gz"), "rt") as f:
            ref_top = [l.rstrip("\n").split("\t") for l in f][1:]
        _, got_top = read_results_tsv("chi2_results_Pheno_top1000.tsv")
        assert [g[2] for g in got_top] == [r[2] for r in ref_top], run
        with gzip.open(os.path.join(gd, run, "Pheno_MLdf.csv.gz"), "rt") as f:
            ref_csv = list(csv.reader(f))
        with open("Pheno_MLdf.csv") as f:
            got_csv = list(csv.reader(f))
        assert [r[0] for r in got_csv] == [r[0] for r in ref_csv] and [r[-2:] for r in got_csv] == [r[-2:] for r in ref_csv]
        if len(ref) - 1 <= 1000:      # no tie class cut by the top-n boundary: the selected k-mers are the same set
            assert sorted(got_csv[0][1:-2]) == sorted(ref_csv[0][1:-2])


@pytest.mark.parametrize("route", ["default", "device", "r04"])
def test_gzip_inputs_with_and_without_the_suffix(tmp_path, route, monkeypatch):
    """Compressed inputs go to the library as they are (r05), whatever their names say (magic bytes decide): result tables and
    predictions equal the plain run's.  route: default -- this small a set is inflated by zlib on the library's host threads;
    device -- on the GPU (PSK_GZ_DEVICE_MIN_MB=0; csrc/gz_inflate.hip); r04 -- PSK_NO_GPU_GZ=1: the library refuses .gz files
    (PSK_EGZIP) and modeling.py / prediction.py inflate them and take the in-memory calls."""
    import gzip
    if route == "device":
        monkeypatch.setenv("PSK_GZ_DEVICE_MIN_MB", "0")
    elif route == "r04":
        monkeypatch.setenv("PSK_NO_GPU_GZ", "1")
    ds = load_dataset("ds_bonf")
    plain, packed = tmp_path / "plain", tmp_path / "packed"
    for d in (plain, packed):
        d.mkdir()
        _write_dataset(ds, str(d))
    rows = open(packed / "data.pheno").read().splitlines()
    out = [rows[0]]
    for j, line in enumerate(rows[1:]):
        name, fn, ph = line.split()
        raw = open(packed / fn, "rb").read()
        if j % 3 == 0:      # compressed, named .gz
            with gzip.open(packed / (fn + ".gz"), "wb") as f:
                f.write(raw)
            os.remove(packed / fn)
            fn += ".gz"
        elif j % 3 == 1:    # compressed, name unchanged
            with gzip.open(packed / fn, "wb") as f:
                f.write(raw)
        out.append("\t".join([name, fn, ph]))
    open(packed / "data.pheno", "w").write("\n".join(out) + "\n")
    _run(plain, ["modeling", "data.pheno"])
    _run(packed, ["modeling", "data.pheno"])
    for fn in ("chi2_results_Pheno.tsv", "Pheno_MLdf.csv", "k-mers_and_coefficients_in_log_reg_model_Pheno.txt"):
        assert (plain / fn).read_bytes() == (packed / fn).read_bytes(), fn
    # the same through prediction on the samples whose file name says nothing (plain, or gzip without the suffix: the
    # file call is refused and the run inflates them)
    for d in (plain, packed):
        lines = open(d / "data.pheno").read().splitlines()[1:]
        open(d / "samples.txt", "w").write("".join("\t".join(l.split()[:2]) + "\n" for j, l in enumerate(lines) if j % 3))
        open(d / "phenos.txt", "w").write("Pheno\tlog_reg_model_Pheno.pkl\n")
        _run(d, ["prediction", "samples.txt", "phenos.txt"])
    assert (plain / "predictions_Pheno.txt").read_bytes() == (packed / "predictions_Pheno.txt").read_bytes()
    assert len((plain / "predictions_Pheno.txt").read_text().splitlines()) > 20


def test_cfg1_example_dataset(tmp_path):
    """BASELINE config 1 (the reference's C. difficile example, /root/reference/example/test_PS_modeling.sh:12-25):
    the 174 MB tarball is not reachable offline, so this runs only where PSK_CFG1_TARBALL points at it --
    tools/cfg1_repro.py is the recipe (with --reference in the build container it runs the unmodified reference beside
    the product and compares the filtered k-mer lists)."""
    import subprocess
    import sys
    from helpers import ROOT
    tarball = os.environ.get("PSK_CFG1_TARBALL")
    if not tarball or not os.path.exists(tarball):
        pytest.skip("PS_modeling_example_files.tar.gz is not available offline (set PSK_CFG1_TARBALL)")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cfg1_repro.py"), tarball]
    if os.path.isdir("/root/reference/bin"):
        cmd.append("--reference")
    r = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=3600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "chi2_results_" in r.stdout


def test_continuous_phenotype_end_to_end(tmp_path, oracle):
    """Welch t-test + Lasso path: the written t-test rows equal the oracle's, the model explains
    the planted effect."""
    import joblib
    from phenotypeseeker_amd.synth import GenomeSet
    gs = GenomeSet(40, 8000, seed=41, gene_len=150)
    os.chdir(tmp_path)
    rows = ["ID\tAddresses\tMIC"]
    pheno = []
    for i in range(gs.n):
        name, fa = gs.sample(i)
        with open(name + ".fasta", "wb") as f:
            f.write(fa)
        v = "NA" if i == 7 else repr(round(gs.continuous_phenotype(i), 4))
        pheno.append(v)
        rows.append("%s\t%s.fasta\t%s" % (name, name, v))
    with open("data.pheno", "w") as f:
        f.write("\n".join(rows) + "\n")
    _run(tmp_path, ["modeling", "data.pheno", "--pvalue", "0.05"])
    head, got = read_results_tsv("t-test_results_MIC.tsv")
    assert head == ["k-mer", "t-test", "p-value", "+_group_mean", "-_group_mean", "num_samples_w_kmer", "samples_with_kmer"]
    assert len(got) > 20
    # oracle on the same data
    k, n = 13, gs.n
    wl = [oracle.count_kmers(gs.sample(i)[1], k)[0] for i in range(n)]
    uw = oracle.union(wl)
    bits = oracle.presence_bits(wl, uw)
    ph = [("NA" if p == "NA" else float(p)) for p in pheno]
    ref = oracle.ttest_scan(bits, ph, np.ones(n), n, 2, n - 2, 0.05, len(uw))
    keep = np.nonzero(ref["keep"])[0]
    want = {oracle.word_to_kmer(uw[r], k): (repr(oracle.round2(ref["stat"][r])), oracle.pstring(ref["p"][r]),
                                             repr(oracle.round2(ref["mean_x"][r])), repr(oracle.round2(ref["mean_y"][r])),
                                             str(int(ref["n_with"][r]))) for r in keep}
    assert {g[0] for g in got} == set(want)
    for g in got:
        assert tuple(g[1:6]) == want[g[0]], g[0]
    pkg = joblib.load("linreg_model_MIC.pkl")
    assert pkg["pred_scale"] == "continuous"
    assert "Parameters:\nLasso()" in open("summary_of_linreg_analysis_MIC.txt").read()


def test_weighted_modeling_end_to_end(tmp_path, oracle):
    """-w: GPU MinHash sketches -> Mash distances -> NJ -> GSC weights -> weighted chi2 scan.
    The weights the CLI used must equal tests/golden/gsc_kat.json's chain for this genome set -- real `mash`, the
    reference's own distance-matrix plumbing and GSC recursion (modeling.py:386-444, :461-503), the oracle's neighbour
    joining between them -- `distances.mat` must be the reference's file byte for byte, and the written rows must equal
    the oracle's weighted scan with those weights."""
    import json
    from helpers import GOLDEN
    from phenotypeseeker_amd import modeling as M
    with open(os.path.join(GOLDEN, "gsc_kat.json")) as f:
        kat = json.load(f)
    ds = load_dataset("ds_omitB")
    _write_dataset(ds, str(tmp_path))
    _run(tmp_path, ["modeling", "data.pheno", "-w", "--omit_B_correction", "--n_kmers", "100"])
    names, n, k = ds["names"], len(ds["names"]), ds["meta"]["k"]
    got_w = [M.Input.samples[nm].weight for nm in names]
    plumbing = next(r for r in kat["plumbing"] if r["tag"] == "ds_omitB")
    chain = kat["chains"]["ds_omitB"]
    assert chain["names"] == names
    for nm in names:      # the GPU sketches are mash's (sample 5 is FASTQ reads: `mash sketch -r` takes them as they are)
        assert M.Input.samples[nm].sketch == plumbing["hashes"][nm], nm
    assert open("distances.mat").read() == plumbing["distances_mat"]
    assert open("tree_newick.txt").read() == chain["newick"] + "\n"
    assert got_w == [chain["weights"][nm] for nm in names]
    assert sum(got_w) == pytest.approx(n) and max(got_w) > 1.0 > min(got_w)
    wl = [oracle.count_kmers(ds["files"][nm], k)[0] for nm in names]
    uw = oracle.union(wl)
    bits = oracle.presence_bits(wl, uw)
    ref = oracle.chi2_scan(bits, ds["pheno"], np.array(got_w), n, 2, n - 2, 0.05, True, len(uw))
    keep = np.nonzero(ref["keep"])[0]
    want = {oracle.word_to_kmer(uw[r], k): (oracle.round2(ref["stat"][r]), oracle.pstring(ref["p"][r])) for r in keep}
    head, got = read_results_tsv("chi2_results_Pheno.tsv")
    # the weighted cells of the kept rows are summed in the reference's order: the same rows and the same printed strings
    assert len(got) == len(want)
    for g in got:
        assert float(g[1]) == want[g[0]][0] and g[2] == want[g[0]][1], g[:3]
    with open("Pheno_MLdf.csv") as f:
        rows = list(csv.reader(f))
    assert np.allclose([float(r[-2]) for r in rows[1:]], [w for w, p in zip(got_w, ds["pheno"]) if p != "NA"])


def test_outer_cv_and_holdout_modes(tmp_path):
    """-cv1 (outer StratifiedKFold) and -ts/-tow (hold-out with random_state 55): sections of the
    summary, and the hold-out model equals a direct grid search on the same training rows."""
    import joblib
    from phenotypeseeker_amd import cv
    from phenotypeseeker_amd.engine import PskContext
    from phenotypeseeker_amd.model import GridSearch, L1LogisticRegression
    ds = load_dataset("ds_bonf")
    _write_dataset(ds, str(tmp_path))
    _run(tmp_path, ["modeling", "data.pheno", "-cv1", "3"])
    txt = open("summary_of_log_reg_analysis_Pheno.txt").read()
    for k in (1, 2, 3):
        assert "##### Train/test split nr.%d: #####" % k in txt
    assert "Mean performance metrics over all train splits:" in txt and "Mean performance metrics over all test splits:" in txt
    assert "### Outputting the last model to a model file! ###" in txt and txt.count("Test set:") == 3
    assert os.path.exists("log_reg_model_Pheno.pkl")

    _run(tmp_path, ["modeling", "data.pheno", "-jt", "modelling", "-ts", "0.25", "-tow"])
    txt = open("summary_of_log_reg_analysis_Pheno.txt").read()
    assert "Training set:" in txt and "Test set:" in txt
    assert "The final output model training on the whole dataset:" in txt
    assert "### Outputting the last model trained on whole data to a model file! ###" in txt

    _run(tmp_path, ["modeling", "data.pheno", "-jt", "modelling", "-ts", "0.25"])
    pkg = joblib.load("log_reg_model_Pheno.pkl")
    with open("Pheno_MLdf.csv") as f:
        rows = list(csv.reader(f))
    X = np.array([[int(v) for v in r[1:-2]] for r in rows[1:]], dtype=np.float64)
    y = np.array([int(r[-1]) for r in rows[1:]])
    tr, te = cv.train_test_split_indices(len(y), 0.25, y, 55)
    assert len(te) == int(np.ceil(0.25 * len(y)))
    inner = int(min(np.bincount(y[tr]).min(), 10))
    with PskContext(0) as ctx:
        ref = GridSearch(L1LogisticRegression(tol=1e-4, max_iter=1000), "C", [1.0 / a for a in np.logspace(-3, 3, 13)],
                         inner).fit(X[tr], y[tr], ctx)
    assert pkg["model"].best_params_ == ref.best_params_
    assert np.allclose(pkg["model"].best_estimator_.coef_, ref.best_estimator_.coef_, atol=1e-9)
    txt = open("summary_of_log_reg_analysis_Pheno.txt").read()
    assert "### Outputting the model to a file! ###" in txt
    # the listed training samples are the split's, in its order
    block = txt.split("Training set:")[1].split("Test set:")[0]
    listed = [l.split()[0] for l in block.splitlines() if l.startswith("S0")]
    assert listed == [rows[1 + i][0] for i in tr]


def test_kmerdb_real_counts_and_mpheno(tmp_path, oracle):
    """--kmerDB (glistcompare -i of modeling.py:367-372), -rc (counts instead of presence in the ML matrix,
    :693-695) and --mpheno (second phenotype column only) through the CLI, against the oracle."""
    from phenotypeseeker_amd.synth import GenomeSet
    gs = GenomeSet(24, 6000, seed=91, gene_len=120)
    k, n = 13, gs.n
    os.chdir(tmp_path)
    rows = ["ID\tAddresses\tdummy\tPheno"]
    files = []
    for i in range(n):
        name, fa = gs.sample(i)
        if i % 3 == 0:   # duplicate a stretch so that some k-mers occur twice in the sample
            body = fa.split(b"\n", 1)[1].replace(b"\n", b"")
            fa = fa + b">dup\n" + body[2980:3100] + b"\n"
        files.append(fa)
        with open(name + ".fasta", "wb") as f:
            f.write(fa)
        rows.append("%s\t%s.fasta\t%d\t%d" % (name, name, (i // 3) % 2, gs.phenotype(i)))
    with open("data.pheno", "w") as f:
        f.write("\n".join(rows) + "\n")
    # database = the planted gene +- flanks of the ancestor
    anc = gs.ancestor
    db_codes = np.concatenate([anc[2500:3000], gs.gene, anc[3000:3300]])
    from phenotypeseeker_amd.synth import wrap_fasta
    with open("db.fasta", "wb") as f:
        f.write(wrap_fasta("db", db_codes))
    _run(tmp_path, ["modeling", "data.pheno", "--mpheno", "2", "-rc", "--kmerDB", "db.fasta", "--omit_B_correction",
                    "--pvalue", "0.5"])
    assert not os.path.exists("chi2_results_dummy.tsv")
    head, got = read_results_tsv("chi2_results_Pheno.tsv")
    lists = [oracle.count_kmers(fa, k) for fa in files]
    dbw = oracle.count_kmers(open("db.fasta", "rb").read(), k)[0]
    uw = oracle.intersect(oracle.union([l[0] for l in lists]), dbw)
    bits = oracle.presence_bits([l[0] for l in lists], uw)
    pheno = [gs.phenotype(i) for i in range(n)]
    ref = oracle.chi2_scan(bits, pheno, np.ones(n), n, 2, n - 2, 0.5, True, len(uw))
    keep = np.nonzero(ref["keep"])[0]
    want = {oracle.word_to_kmer(uw[r], k): (repr(oracle.round2(ref["stat"][r])), oracle.pstring(ref["p"][r])) for r in keep}
    assert {g[0] for g in got} == set(want) and len(want) > 30
    for g in got:
        assert (g[1], g[2]) == want[g[0]]
    with open("Pheno_MLdf.csv") as f:
        ml = list(csv.reader(f))
    cols = ml[0][1:-2]
    saw_multi = False
    for j, kmer in enumerate(cols):
        w = oracle.canonical_word(oracle.kmer_to_word(kmer), k)
        for i in range(n):
            cnt = oracle.map_counts(lists[i][0], lists[i][1], np.array([w], dtype=np.uint64))[0]
            assert int(ml[1 + i][1 + j]) == cnt
            saw_multi |= cnt > 1
    assert saw_multi


def test_l2_penalty_end_to_end(tmp_path):
    """`--penalty L2` (set_model, modeling.py:1001-1002, :1015-1019): Ridge for a continuous phenotype, L2
    logistic regression (lbfgs objective by default, liblinear's with -ls) for a binary one; the stored model
    is the unique optimum of its objective on the written <pheno>_MLdf.csv."""
    import joblib
    from oracle import oracle_model as OM
    from phenotypeseeker_amd.synth import GenomeSet
    gs = GenomeSet(40, 8000, seed=41, gene_len=150)
    os.chdir(tmp_path)
    for col, val in (("MIC", lambda i: repr(round(gs.continuous_phenotype(i), 4))), ("Pheno", lambda i: str(gs.phenotype(i)))):
        rows = ["ID\tAddresses\t" + col]
        for i in range(gs.n):
            name, fa = gs.sample(i)
            with open(name + ".fasta", "wb") as f:
                f.write(fa)
            rows.append("%s\t%s.fasta\t%s" % (name, name, val(i)))
        with open("data_%s.pheno" % col, "w") as f:
            f.write("\n".join(rows) + "\n")

    def design(csv_name):
        with open(csv_name) as f:
            rows = list(csv.reader(f))
        return (np.array([[float(v) for v in r[1:-2]] for r in rows[1:]]), np.array([float(r[-1]) for r in rows[1:]]))

    _run(tmp_path, ["modeling", "data_MIC.pheno", "--pvalue", "0.05", "--penalty", "L2", "--n_kmers", "60"])
    pkg = joblib.load("linreg_model_MIC.pkl")
    X, y = design("MIC_MLdf.csv")
    m = pkg["model"]
    w, b = OM.ridge_fit(X, y, m.best_params_["alpha"])
    assert np.allclose(m.best_estimator_.coef_, w, rtol=1e-6, atol=1e-8) and m.best_estimator_.intercept_ == pytest.approx(b, rel=1e-7)
    assert "Parameters:\nRidge(" in open("summary_of_linreg_analysis_MIC.txt").read()

    for extra, pen in (([], False), (["-ls", "liblinear"], True)):
        _run(tmp_path, ["modeling", "data_Pheno.pheno", "--omit_B_correction", "--penalty", "L2", "--n_kmers", "60",
                        "--tolerance", "1e-10"] + extra)
        pkg = joblib.load("log_reg_model_Pheno.pkl")
        X, y = design("Pheno_MLdf.csv")
        m = pkg["model"]
        w, b = OM.logreg_l2_fit(X, y.astype(int), m.best_params_["C"], penalise_intercept=pen)
        assert np.allclose(m.best_estimator_.coef_[0], w, rtol=1e-5, atol=1e-7), extra
        assert m.best_estimator_.intercept_[0] == pytest.approx(b, rel=1e-5, abs=1e-7)
        assert (m.predict(X) == y).mean() >= 0.9
        summary = open("summary_of_log_reg_analysis_Pheno.txt").read()
        assert "Parameters:\nLogisticRegression(max_iter=1000" in summary and "penalty='l1'" not in summary


@pytest.mark.parametrize("ranks,flags,ingest", [(2, [], "exchange"), (2, [], "redundant"), (3, ["-w", "--omit_B_correction", "--n_kmers", "100"], "exchange"),
                                                (2, ["-l", "21"], "exchange"), (3, ["-l", "31", "--omit_B_correction"], "redundant"),
                                                (4, [], "redundant"), (8, [], "exchange")])      # (byte-identical for G in {1, 2, 4, 8})
def test_multi_rank_modeling_writes_the_same_files(tmp_path, ranks, flags, ingest):
    """SURVEY.md 8(e) invariant on the real pipeline: `PSK_GPUS=<ranks> phenotypeseeker modeling` -- the CLI starts its
    own ranks (launch.py; no outside launcher) -- with several ranks (all on the one visible GPU, collectives through the gloo transport of tests/: PSK_SHARE_GPU /
    PSK_DIST_TRANSPORT -- RCCL refuses two ranks on one device) and balanced, quantile-cut slabs produces
    byte-identical result tables and the same model as the one-rank run.  Ingest either way: every sample counted
    on one rank and the slab ranges of the lists exchanged (dist.ListExchange, the default), or every rank counting
    every sample with its slab filter (PSK_REDUNDANT_INGEST=1); -w adds the sketch all-gather."""
    import subprocess
    import sys
    import joblib
    from helpers import ROOT
    ds = load_dataset("ds_bonf")
    one, two = tmp_path / "one", tmp_path / "two"
    for d in (one, two):
        d.mkdir()
        _write_dataset(ds, str(d))
    _run(one, ["modeling", "data.pheno"] + flags)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PSK_SHARE_GPU="1", PSK_DIST_TRANSPORT="_gloo_transport:GlooTransport",
               PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1",
               PSK_REDUNDANT_INGEST="1" if ingest == "redundant" else "0")
    env.update(PSK_GPUS=str(ranks), MASTER_PORT="29619")
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(var, None)
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "phenotypeseeker"), "modeling", "data.pheno"] + flags
    r = subprocess.run(cmd, env=env, cwd=str(two), timeout=600, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    top = "chi2_results_Pheno_top%d.tsv" % (100 if "--n_kmers" in flags else 1000)
    for name in ("chi2_results_Pheno.tsv", top, "Pheno_MLdf.csv", "k-mers_and_coefficients_in_log_reg_model_Pheno.txt"):
        assert (one / name).read_bytes() == (two / name).read_bytes(), name
    a, b = joblib.load(str(one / "log_reg_model_Pheno.pkl")), joblib.load(str(two / "log_reg_model_Pheno.pkl"))
    assert list(a["kmers"]) == list(b["kmers"])
    assert np.array_equal(a["model"].best_estimator_.coef_, b["model"].best_estimator_.coef_)


def test_several_phenotypes_in_one_run_equal_one_run_each(tmp_path):
    """The phenotypes of one `modeling` run are scanned as a pipeline (the scan of phenotype j + 1 is launched before
    the survivors of j are read: two result sets in the context); every phenotype's files must equal those of a run
    on that phenotype alone (--mpheno)."""
    ds = load_dataset("ds_bonf")
    both = tmp_path / "all"
    both.mkdir()
    _write_dataset(ds, str(both))
    lines = open(both / "data.pheno").read().splitlines()
    rng = np.random.default_rng(3)
    head = lines[0].split("\t") + ["Flip", "Third", "Rand"]
    rows = []
    for i, l in enumerate(lines[1:]):
        f = l.split("\t")
        v = int(f[2]) if f[2] != "NA" else i % 2
        rows.append(f + [str(1 - v), str(v if i % 5 else 1 - v), "NA" if i % 7 == 0 else str(int(rng.random() < 0.5))])
    txt = "\n".join(["\t".join(head)] + ["\t".join(r) for r in rows]) + "\n"
    open(both / "data.pheno", "w").write(txt)
    _run(both, ["modeling", "data.pheno", "--pvalue", "0.5", "--omit_B_correction"])
    for col, name in enumerate(head[2:], start=1):
        one = tmp_path / ("only%d" % col)
        one.mkdir()
        _write_dataset(ds, str(one))
        open(one / "data.pheno", "w").write(txt)
        _run(one, ["modeling", "data.pheno", "--pvalue", "0.5", "--omit_B_correction", "--mpheno", str(col)])
        for fn in ("chi2_results_%s.tsv" % name, "%s_MLdf.csv" % name,
                   "k-mers_and_coefficients_in_log_reg_model_%s.txt" % name):
            if os.path.exists(one / fn) or os.path.exists(both / fn):
                assert open(one / fn).read() == open(both / fn).read(), fn
        assert os.path.exists(both / ("chi2_results_%s.tsv" % name))


@pytest.mark.parametrize("tag,pheno,short", [("ds_omitB", "Pheno", "log_reg"), ("ds_bonf", "Pheno", "log_reg"), ("ds_cont", "MIC", "linreg")])
def test_prediction_on_a_committed_model_file_equals_the_references_prediction(tmp_path, tag, pheno, short):
    """The .pkl contract in the reverse direction: tests/golden/product_pkl/<set>/*.pkl is a model file THIS
    package wrote on a GPU box; predictions_<pheno>.txt beside it was written by the REFERENCE's prediction.py (185-204) after
    a plain joblib.load of that file, bin/gmer_counter counting the k-mers (oracle/gen_golden.py::
    gen_prediction_of_product_pkl).  `phenotypeseeker prediction` of this package on the same file and samples must write
    the same bytes."""
    import shutil
    from helpers import GOLDEN
    gd = os.path.join(GOLDEN, "product_pkl", tag)
    ds = load_dataset("ds_omitB" if tag == "ds_cont" else tag)
    _write_dataset(ds, str(tmp_path))
    for fn in ("samples.txt", "phenos.txt", "%s_model_%s.pkl" % (short, pheno)):
        shutil.copy(os.path.join(gd, fn), str(tmp_path))
    _run(tmp_path, ["prediction", "samples.txt", "phenos.txt"])
    fn = "predictions_%s.txt" % pheno
    assert open(fn).read() == open(os.path.join(gd, fn)).read()


@pytest.mark.parametrize("sub,extra", [("whole", []), ("holdout", ["-ts", "0.25"]), ("outer_cv", ["-cv1", "3"])])
def test_regressor_files_equal_the_reference_written_ones(tmp_path, sub, extra):
    """a11, regressor branch: the reference ran `modeling data.pheno -jt modelling` (+ -ts 0.25 /
    -cv1 3) on tests/golden/ds_cont/MIC_MLdf.csv (oracle/gen_golden.py::gen_model_files); scikit-learn's Lasso is
    deterministic, so THIS package's run of the same command must write the same summary -- grid scores (nan where a
    10-fold split of 18 samples leaves one-sample folds), best alpha, per-sample predictions, MSE / R^2 / Spearman /
    Pearson / one-dilution lines, every layout line -- and the same coefficient table: text identical, numbers within 1e-6
    relative (north_star's bound for regression coefficients; in practice the last digit of the printed doubles)."""
    import shutil
    from helpers import GOLDEN, assert_text_equal_up_to_numbers
    gd = os.path.join(GOLDEN, "ds_cont")
    for fn in ("data.pheno", "MIC_MLdf.csv"):
        shutil.copy(os.path.join(gd, fn), str(tmp_path))
    _run(tmp_path, ["modeling", "data.pheno", "-jt", "modelling"] + extra)
    fn = "summary_of_linreg_analysis_MIC.txt"
    assert_text_equal_up_to_numbers(open(fn).read(), open(os.path.join(gd, sub, fn)).read(), rel=1e-6, abs_=1e-9, what=sub + " summary")
    fn = "k-mers_and_coefficients_in_linreg_model_MIC.txt"
    got = [l.split("\t") for l in open(fn).read().splitlines()]
    want = [l.split("\t") for l in open(os.path.join(gd, sub, fn)).read().splitlines()]
    assert got[0] == want[0] and len(got) == len(want)
    assert [(g[0], g[2], g[3]) for g in got[1:]] == [(w[0], w[2], w[3]) for w in want[1:]]
    assert np.allclose([float(g[1]) for g in got[1:]], [float(w[1]) for w in want[1:]], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("tag,extra", [("ds_omitB", ["--omit_B_correction", "--n_kmers", "100"]), ("ds_bonf", [])])
def test_classifier_files_against_the_reference_written_ones(tmp_path, tag, extra):
    """a11, classifier branch: the whole pipeline on the golden genome sets against the summary / coefficient table the
    unmodified reference wrote for them (tests/golden/<set>/model/).  liblinear runs unseeded and unconverged there (SURVEY
    Q6: two reference runs differ in grid scores, C and coefficients), so those NUMBERS are masked -- every line must still
    be the same line: same text, same number of numeric fields, the grid's parameter dicts, sample ids and actual
    phenotypes identical, the chosen C one of the grid's; the coefficient table names the same k-mers with the same carrier
    counts and sample lists.  (That the writers reproduce the reference's bytes when the numbers ARE the same is
    tests/test_host_modeling.py::test_summary_and_coefficient_files_equal_the_reference_byte_for_byte.)"""
    from helpers import mask_numbers
    ds = load_dataset(tag)
    _write_dataset(ds, str(tmp_path))
    _run(tmp_path, ["modeling", "data.pheno"] + extra)
    ref = os.path.join(ds["dir"], "model")
    fn = "summary_of_log_reg_analysis_Pheno.txt"
    g, w = open(fn).read().splitlines(), open(os.path.join(ref, fn)).read().splitlines()
    assert len(g) == len(w)
    in_predictions = False
    for a, b in zip(g, w):
        assert mask_numbers(a) == mask_numbers(b), (a, b)
        if a.startswith("Sample_ID "):
            in_predictions = True
        elif in_predictions and not a.strip():
            in_predictions = False
        elif in_predictions:
            assert a.split()[:2] == b.split()[:2], (a, b)               # sample id and actual phenotype
        if " for {" in a:
            assert a.split(" for ")[1] == b.split(" for ")[1], (a, b)    # the parameter dict of a grid-score line
        if a.startswith("C : "):
            assert any(abs(float(a[4:]) - 1.0 / al) <= 1e-12 / al for al in np.logspace(-3, 3, 13)), a
    fn = "k-mers_and_coefficients_in_log_reg_model_Pheno.txt"
    got = [l.split("\t") for l in open(fn).read().splitlines()]
    want = [l.split("\t") for l in open(os.path.join(ref, fn)).read().splitlines()]
    assert got[0] == want[0] and len(got) == len(want)
    if tag == "ds_bonf":         # (no cut inside a p-value tie class: the same k-mers; ds_omitB cuts 100 out of a larger class)
        assert sorted((x[0], x[2], x[3]) for x in got[1:]) == sorted((x[0], x[2], x[3]) for x in want[1:])
    else:
        both = {x[0]: (x[2], x[3]) for x in want[1:]}
        assert all(both[x[0]] == (x[2], x[3]) for x in got[1:] if x[0] in both)
