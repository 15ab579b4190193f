"""CPU tests of the host mirror (phenotypeseeker_amd.modeling) against the reference's own
output files in tests/golden/: data.pheno parsing, option defaults, and the selection stage
(get_ML_df: p-value-string ordering, TSV / MLdf.csv writers) fed with rows from the oracle."""
import csv
import os

import numpy as np
import pytest

from helpers import load_dataset, read_results_tsv


ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_dataset(ds, tmp):
    for name, data in ds["files"].items():
        fn = [l.split()[1] for l in open(os.path.join(ds["dir"], "data.pheno")).read().splitlines()[1:]
              if l.split()[0] == name][0]
        with open(os.path.join(tmp, fn), "wb") as f:
            f.write(data)
    with open(os.path.join(ds["dir"], "data.pheno")) as f:
        txt = f.read()
    with open(os.path.join(tmp, "data.pheno"), "w") as f:
        f.write(txt)


def _args(extra=()):
    from phenotypeseeker_amd.cli import build_parser
    return build_parser().parse_args(["modeling", "data.pheno"] + list(extra))


def _setup(tmp_path, tag, extra=()):
    from phenotypeseeker_amd import modeling as M
    ds = load_dataset(tag)
    _write_dataset(ds, str(tmp_path))
    os.chdir(tmp_path)
    a = _args(extra)
    M.Input.reset()
    M.Input.get_input_data(a.inputfile, a.take_logs, a.mpheno)
    M.Input.Input_args(a.alphas, a.alpha_min, a.alpha_max, a.n_alphas, a.gammas, a.gamma_min, a.gamma_max, a.n_gammas,
                       a.min, a.max, a.kmer_length, a.cutoff, a.num_threads, a.pvalue, a.n_kmers, a.binary_classifier,
                       a.regressor, a.penalty, a.max_iter, a.tolerance, a.l1_ratio, a.n_splits_cv_outer, a.kernel,
                       a.n_iter, a.n_splits_cv_inner, a.testset_size, a.train_on_whole, a.logreg_solver, a.jump_to,
                       a.pca, a.real_counts, a.omit_B_correction, a.kmerDB)
    return M, ds


def test_input_parsing_and_defaults(tmp_path):
    M, ds = _setup(tmp_path, "ds_omitB", ["--omit_B_correction", "--n_kmers", "100"])
    assert list(M.Input.samples) == ds["names"]
    assert M.Samples.no_samples == 20 and M.phenotypes.pred_scale == "binary"
    ph = M.Input.phenotypes_to_analyse["Pheno"]
    assert ph.no_samples == 18  # two NA rows
    assert [s.phenotypes["Pheno"] for s in M.Input.samples.values()] == ds["pheno"]
    assert (M.Samples.min_samples, M.Samples.max_samples) == (2, 18)  # --min 0 -> 2, --max 0 -> N - 2
    assert np.allclose(M.phenotypes.alphas, np.logspace(-3, 3, 13))
    assert M.phenotypes.penalty == "L1" and M.phenotypes.max_iter == 1000.0 and M.phenotypes.kmer_limit == 100


def test_continuous_detection_and_logs(tmp_path):
    from phenotypeseeker_amd import modeling as M
    os.chdir(tmp_path)
    with open("d.pheno", "w") as f:
        f.write("ID\tAddr\tMIC\tbin\nA\ta.fa\t0.25\t1\nB\tb.fa\tNA\t0\nC\tc.fa\t8\tNA\n\n")
    M.Input.reset()
    M.Input.get_input_data("d.pheno", True, [1])
    assert M.phenotypes.pred_scale == "continuous"
    assert list(M.Input.phenotypes_to_analyse) == ["MIC"]
    vals = [s.phenotypes["MIC"] for s in M.Input.samples.values()]
    assert vals == [-2.0, "NA", 3.0]
    assert M.Input.phenotypes_to_analyse["MIC"].no_samples == 2


def test_rejects_options_outside_the_hot_path(tmp_path):
    with pytest.raises(SystemExit):
        _setup(tmp_path, "ds_bonf", ["-bc", "SVM"])
    with pytest.raises(SystemExit):
        _setup(tmp_path, "ds_bonf", ["--penalty", "L1+L2"])
    with pytest.raises(SystemExit):
        _setup(tmp_path, "ds_bonf", ["--penalty", "L2", "-ls", "newton"])
    M, _ = _setup(tmp_path, "ds_bonf", ["--penalty", "L2"])
    assert M.phenotypes.penalty == "L2" and M.phenotypes.logreg_solver == "lbfgs"


@pytest.mark.parametrize("tag,extra", [("ds_omitB", ["--omit_B_correction", "--n_kmers", "100"]), ("ds_bonf", []), ("ds_k21", ["-l", "21"])])
def test_selection_stage_matches_reference_files(tmp_path, oracle, tag, extra):
    M, ds = _setup(tmp_path, tag, extra)
    k, names, n = ds["meta"]["k"], ds["names"], len(ds["names"])
    wl = [oracle.count_kmers(ds["files"][nm], k)[0] for nm in names]
    uw = oracle.union(wl)
    bits = oracle.presence_bits(wl, uw)
    res = oracle.chi2_scan(bits, ds["pheno"], np.ones(n), n, 2, n - 2, 0.05, "--omit_B_correction" in extra, len(uw))
    keep = np.nonzero(res["keep"])[0]
    pres = np.array([[(int(bits[r, i >> 6]) >> (i & 63)) & 1 for i in range(n)] for r in keep], dtype=np.uint8)
    ph = M.Input.phenotypes_to_analyse["Pheno"]
    ph.rows = {"kmer": [oracle.word_to_kmer(uw[r], k) for r in keep], "stat": res["stat"][keep], "p": res["p"][keep],
               "mean_x": np.zeros(len(keep)), "mean_y": np.zeros(len(keep)), "n_with": res["n_with"][keep],
               "presence": pres, "vector": pres}
    ph.get_ML_df()
    head, ref = read_results_tsv(os.path.join(ds["dir"], "chi2_results_Pheno.tsv"))
    head2, got = read_results_tsv("chi2_results_Pheno.tsv")
    assert head2 == head
    assert sorted(got) == sorted(ref)                      # same rows, byte for byte
    assert [g[2] for g in got] == [r[2] for r in ref]      # same p-string order (ties broken by k-mer here)
    top = "chi2_results_Pheno_top%d.tsv" % M.phenotypes.kmer_limit
    _, ref_top = read_results_tsv(os.path.join(ds["dir"], top))
    _, got_top = read_results_tsv(top)
    assert [g[2] for g in got_top] == [r[2] for r in ref_top]
    with open(os.path.join(ds["dir"], "Pheno_MLdf.csv")) as f:
        ref_csv = list(csv.reader(f))
    with open("Pheno_MLdf.csv") as f:
        got_csv = list(csv.reader(f))
    assert [r[0] for r in got_csv] == [r[0] for r in ref_csv]
    assert [r[-2:] for r in got_csv] == [r[-2:] for r in ref_csv]
    ref_cols = {ref_csv[0][j]: [r[j] for r in ref_csv[1:]] for j in range(1, len(ref_csv[0]) - 2)}
    got_cols = {got_csv[0][j]: [r[j] for r in got_csv[1:]] for j in range(1, len(got_csv[0]) - 2)}
    if len(ref) <= M.phenotypes.kmer_limit:               # no cut inside a tie class: identical column set
        assert got_cols == ref_cols
    else:
        assert len(got_cols) == len(ref_cols)
        for kmer in set(got_cols) & set(ref_cols):
            assert got_cols[kmer] == ref_cols[kmer]


def test_list_file_round_trip(tmp_path, oracle):
    from phenotypeseeker_amd import formats
    ds = load_dataset("ds_omitB")
    k = ds["meta"]["k"]
    first = ds["names"][0]
    ref_path = os.path.join(ds["dir"], "%s_0_%d.list" % (first, k))
    kk, w, f = formats.read_list(ref_path)
    assert kk == k
    out = os.path.join(tmp_path, "x.list")
    formats.write_list(out, k, w, f)
    assert open(out, "rb").read() == open(ref_path, "rb").read()
    assert formats.words_to_kmers(w[:5], k) == [oracle.word_to_kmer(x, k) for x in w[:5]]
    assert [formats.canonical(formats.kmer_to_word(s), k) for s in formats.words_to_kmers(w[:50], k)] == w[:50].tolist()


def test_metrics_match_sklearn():
    sk = pytest.importorskip("sklearn.metrics")
    from phenotypeseeker_amd import metrics as Mx
    rng = np.random.default_rng(0)
    for _ in range(20):
        n = int(rng.integers(10, 60))
        y = (rng.random(n) < 0.5).astype(int)
        p = (rng.random(n) < 0.5).astype(int)
        s = rng.random(n).round(1)
        if y.sum() in (0, n):
            continue
        assert Mx.f1(y, p) == pytest.approx(sk.f1_score(y, p))
        assert Mx.roc_auc(y, s) == pytest.approx(sk.roc_auc_score(y, s))
        assert Mx.average_precision(y, s) == pytest.approx(sk.average_precision_score(y, s))
        assert Mx.matthews(y, p) == pytest.approx(sk.matthews_corrcoef(y, p))
        assert Mx.cohen_kappa(y, p) == pytest.approx(sk.cohen_kappa_score(y, p))
        assert Mx.classification_report(y, p) == sk.classification_report(y, p, target_names=["sensitive", "resistant"])


def test_model_files_are_scikit_learn_pickles_written_without_importing_it(tmp_path):
    """the .pkl holds scikit-learn objects (the reference's loader is a plain joblib.load,
    /root/reference/PhenotypeSeeker/prediction.py:124-129).  skpickle writes them from recorded class templates
    without importing scikit-learn; a process that has scikit-learn but not this package loads the file and predicts
    exactly what to_sklearn()'s objects (built through the real constructors) predict."""
    import subprocess
    import sys
    import numpy as np
    from phenotypeseeker_amd import model as M, skpickle
    rng = np.random.default_rng(2)
    X = (rng.random((30, 7)) < 0.4).astype(np.float64)
    files = {}
    for kind in ("logistic", "lasso", "ridge"):
        if kind == "logistic":
            est = M.L1LogisticRegression(C=10.0, tol=1e-4, max_iter=1000)
            est.coef_, est.intercept_ = rng.normal(size=(1, 7)), rng.normal(size=1)
            gs = M.GridSearch(M.L1LogisticRegression(tol=1e-4, max_iter=1000), "C", [0.1, 10.0], 3)
            gs.best_params_ = {"C": 10.0}
        else:
            cls = M.LassoRegression if kind == "lasso" else M.RidgeRegression
            est = cls(alpha=0.5)
            est.coef_, est.intercept_ = rng.normal(size=7), float(rng.normal())
            gs = M.GridSearch(cls(), "alpha", [2.0, 0.5], 3)
            gs.best_params_ = {"alpha": 0.5}
        est.n_features_in_ = 7
        gs.best_estimator_, gs.best_index_, gs.best_score_, gs.n_splits_ = est, 1, 0.75, 3
        gs.cv_results_ = {"mean_test_score": np.array([0.5, 0.75]), "std_test_score": np.array([0.1, 0.2]),
                          "params": [{gs.param_name: v} for v in gs.param_grid[gs.param_name]]}
        shell = gs.to_sklearn_shell()
        assert shell is not None, "no template for the installed scikit-learn: run tools/make_sklearn_shells.py"
        blob = skpickle.dumps({"model": shell, "kmers": np.array(["ACGT", "TTTT"], dtype=object), "pca": False, "pred_scale": "binary"})
        files[kind] = os.path.join(tmp_path, kind + ".pkl")
        with open(files[kind], "wb") as f:
            f.write(blob)
        np.save(os.path.join(tmp_path, kind + "_want.npy"), gs.predict(X))
        if kind == "logistic":
            np.save(os.path.join(tmp_path, "proba_want.npy"), gs.predict_proba(X))
    np.save(os.path.join(tmp_path, "X.npy"), X)
    code = ("import sys, joblib, numpy as np, warnings\n"
            "warnings.simplefilter('error')\n"
            "sys.path = [p for p in sys.path if 'repo' not in p and p not in ('', '.')]\n"
            "X = np.load('X.npy')\n"
            "for kind, cls in (('logistic', 'LogisticRegression'), ('lasso', 'Lasso'), ('ridge', 'Ridge')):\n"
            "    pkg = joblib.load(kind + '.pkl')\n"
            "    m = pkg['model']\n"
            "    assert type(m).__module__ == 'sklearn.model_selection._search' and type(m.best_estimator_).__name__ == cls\n"
            "    assert np.allclose(m.predict(X), np.load(kind + '_want.npy'), rtol=1e-12, atol=1e-12)\n"
            "    assert list(pkg['kmers']) == ['ACGT', 'TTTT'] and pkg['pca'] is False\n"
            "    assert m.cv_results_['params'][1] == m.best_params_ and m.get_params()['cv'] == 3\n"
            "    repr(m); m.best_estimator_.get_params()\n"
            "    if kind == 'logistic':\n"
            "        assert np.allclose(m.predict_proba(X), np.load('proba_want.npy'), rtol=1e-12)\n"
            "assert 'phenotypeseeker_amd' not in sys.modules\n")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    # and the writer itself never pulled scikit-learn in
    probe = ("import sys; sys.path.insert(0, %r); from phenotypeseeker_amd import skpickle, model; "
             "assert skpickle.template('Lasso') is not None; assert 'sklearn' not in sys.modules" % ROOT_DIR)
    assert subprocess.run([sys.executable, "-c", probe], timeout=120).returncode == 0


def test_prediction_reads_linear_models_without_scikit_learn(tmp_path):
    """`prediction` reads a plain-pickle model file of a linear estimator through stub classes and applies it with
    scikit-learn's own expressions: the same labels, probabilities and regression values as joblib.load + scikit-learn,
    and scikit-learn is never imported.  Files it cannot take (joblib-wrapped arrays, as the reference writes them)
    return None, and the ordinary loader serves them."""
    import subprocess
    import sys
    import joblib
    import numpy as np
    from phenotypeseeker_amd import model as M, skpickle
    rng = np.random.default_rng(4)
    X = (rng.random((40, 9)) < 0.4).astype(np.float64)
    for kind in ("logistic", "lasso", "ridge"):
        if kind == "logistic":
            est = M.L1LogisticRegression(C=10.0, tol=1e-4, max_iter=1000)
            est.coef_, est.intercept_ = rng.normal(size=(1, 9)), rng.normal(size=1)
            gs = M.GridSearch(M.L1LogisticRegression(tol=1e-4, max_iter=1000), "C", [0.1, 10.0], 3)
            gs.best_params_ = {"C": 10.0}
        else:
            cls = M.LassoRegression if kind == "lasso" else M.RidgeRegression
            est = cls(alpha=0.5)
            est.coef_, est.intercept_ = rng.normal(size=9), float(rng.normal())
            gs = M.GridSearch(cls(), "alpha", [2.0, 0.5], 3)
            gs.best_params_ = {"alpha": 0.5}
        est.n_features_in_ = 9
        gs.best_estimator_, gs.best_index_, gs.best_score_, gs.n_splits_ = est, 1, 0.75, 3
        gs.cv_results_ = {"mean_test_score": np.array([0.5, 0.75]), "std_test_score": np.array([0.1, 0.2]),
                          "params": [{gs.param_name: v} for v in gs.param_grid[gs.param_name]]}
        path = os.path.join(tmp_path, kind + ".pkl")
        with open(path, "wb") as f:
            f.write(skpickle.dumps({"model": gs.to_sklearn_shell(), "kmers": np.array(["ACGT", "TTTT"], dtype=object), "pca": False,
                                    "pred_scale": "binary" if kind == "logistic" else "continuous"}))
        fast, ref = skpickle.load_linear_package(path), joblib.load(path)
        assert fast is not None and list(fast["kmers"]) == ["ACGT", "TTTT"] and fast["pred_scale"] == ref["pred_scale"]
        assert np.array_equal(fast["model"].predict(X), ref["model"].predict(X))
        if kind == "logistic":
            assert np.allclose(fast["model"].predict_proba(X), ref["model"].predict_proba(X), rtol=1e-14)
            assert [str(round(p[1], 2)) for p in fast["model"].predict_proba(X)] == [str(round(p[1], 2)) for p in ref["model"].predict_proba(X)]
        # what joblib.dump writes (arrays outside the pickle stream, as the reference's files are): not for the fast reader
        jpath = os.path.join(tmp_path, kind + "_joblib.pkl")
        joblib.dump(ref, jpath)
        assert skpickle.load_linear_package(jpath) is None or np.array_equal(skpickle.load_linear_package(jpath)["model"].predict(X),
                                                                             ref["model"].predict(X))
    probe = ("import sys; sys.path.insert(0, %r); from phenotypeseeker_amd import skpickle; "
             "assert skpickle.load_linear_package(%r) is not None; assert 'sklearn' not in sys.modules and 'joblib' not in sys.modules"
             % (ROOT_DIR, os.path.join(str(tmp_path), "logistic.pkl")))
    assert subprocess.run([sys.executable, "-c", probe], timeout=120).returncode == 0


def test_linear_package_loader_leaves_what_it_cannot_reproduce_to_scikit_learn(tmp_path):
    """ADVICE r02 on skpickle.load_linear_package: a binary LogisticRegression saved with multi_class='multinomial' has
    softmax([-d, d]) = expit(2 d) probabilities in scikit-learn -- the stub loader must return None for it (joblib.load +
    scikit-learn then apply the model); a joblib-wrapped file (arrays stored raw behind NumpyArrayWrapper) is refused at
    its first joblib class instead of reading array bytes as opcodes; and the probabilities of the models it does take
    are scipy's expit without overflow warnings at scores of -800."""
    import pickle
    import warnings
    import joblib
    import numpy as np
    from sklearn.linear_model import LogisticRegression
    from phenotypeseeker_amd import skpickle
    rng = np.random.default_rng(4)
    X = (rng.random((40, 5)) < 0.5).astype(np.float64)
    y = (X[:, 0] + X[:, 1] > 0.5).astype(int)
    est = LogisticRegression(penalty="l1", solver="liblinear", C=10.0).fit(X, y)
    pkg = {"model": est, "kmers": np.array(list("ACGTA"), dtype=object), "pca": False, "pred_scale": "binary"}
    plain = os.path.join(tmp_path, "plain.pkl")
    with open(plain, "wb") as f:
        pickle.dump(pkg, f, protocol=4)
    fast = skpickle.load_linear_package(plain)
    assert fast is not None
    big = np.array([[800.0, 0, 0, 0, 0], [-800.0, 0, 0, 0, 0]]) / max(abs(est.coef_[0, 0]), 1e-3)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        pr = fast["model"].predict_proba(np.vstack([X, big]))
    assert np.allclose(pr[:40], est.predict_proba(X), rtol=1e-12, atol=0) and np.all(np.isfinite(pr))
    est.multi_class = "multinomial"          # what an older scikit-learn records for LogisticRegression(multi_class='multinomial')
    multi = os.path.join(tmp_path, "multi.pkl")
    with open(multi, "wb") as f:
        pickle.dump(pkg, f, protocol=4)
    assert skpickle.load_linear_package(multi) is None
    del est.multi_class
    wrapped = os.path.join(tmp_path, "wrapped.pkl")
    pkg["kmers"] = np.arange(100_000)         # a large numeric array: joblib stores it raw behind a NumpyArrayWrapper
    joblib.dump(pkg, wrapped)
    assert skpickle.load_linear_package(wrapped) is None


class _RefModel:
    """The reference's fitted GridSearchCV (tests/golden/<set>/model/*.pkl, written by the unmodified modeling.py) behind
    the estimator boundary: scikit-learn >= 1.x returns Python floats from score() where the writers (like the reference's,
    modeling.py:1316-1356) call .round() on numpy ones -- oracle/ref_shim.py wraps the same call for the reference."""

    def __init__(self, fitted):
        self._m = fitted

    def __getattr__(self, name):
        return getattr(self._m, name)

    def score(self, X, y):
        return np.float64(self._m.score(X, y))


_MODEL_CASES = [
    ("ds_omitB", "model", "Pheno", "log_reg", ["--omit_B_correction", "--n_kmers", "100"]),
    ("ds_bonf", "model", "Pheno", "log_reg", []),
    ("ds_cont", "whole", "MIC", "linreg", []),
    ("ds_cont", "holdout", "MIC", "linreg", ["-ts", "0.25"]),
]


@pytest.mark.parametrize("tag,sub,pheno,short,extra", _MODEL_CASES)
def test_summary_and_coefficient_files_equal_the_reference_byte_for_byte(tmp_path, monkeypatch, tag, sub, pheno, short, extra):
    """a11 formats.  The reference's own fitted model (the .pkl it wrote) is handed to THIS
    package's writers in the fit's place: summary_of_<model>_analysis_<pheno>.txt (modeling.py:1219-1412) and
    k-mers_and_coefficients_in_<model>_model_<pheno>.txt (:1414-1455) must then be the reference's files byte for byte --
    classifier and regressor branch, whole-set and hold-out layout -- since nothing but formatting is left to differ."""
    import joblib
    from helpers import GOLDEN
    from phenotypeseeker_amd import modeling as M
    gd = os.path.join(GOLDEN, tag)
    os.chdir(tmp_path)
    for fn in ("data.pheno", pheno + "_MLdf.csv"):
        with open(os.path.join(gd, fn)) as f, open(fn, "w") as g:
            g.write(f.read())
    a = _args(["-jt", "modelling"] + extra)
    M.Input.reset()
    M.Input.get_input_data(a.inputfile, a.take_logs, a.mpheno)
    M.Input.Input_args(a.alphas, a.alpha_min, a.alpha_max, a.n_alphas, a.gammas, a.gamma_min, a.gamma_max, a.n_gammas,
                       a.min, a.max, a.kmer_length, a.cutoff, a.num_threads, a.pvalue, a.n_kmers, a.binary_classifier,
                       a.regressor, a.penalty, a.max_iter, a.tolerance, a.l1_ratio, a.n_splits_cv_outer, a.kernel,
                       a.n_iter, a.n_splits_cv_inner, a.testset_size, a.train_on_whole, a.logreg_solver, a.jump_to,
                       a.pca, a.real_counts, a.omit_B_correction, a.kmerDB)
    ref_dir = os.path.join(gd, sub)
    pkg = joblib.load(os.path.join(ref_dir, "%s_model_%s.pkl" % (short, pheno)))
    ph = M.Input.phenotypes_to_analyse[pheno]
    fits = []

    def fit_is_the_reference_model(self, ctx, X, y):
        fits.append(X.shape)
        self.model = pkg["model"].estimator
        self.model_fitted = _RefModel(pkg["model"])
    monkeypatch.setattr(M.phenotypes, "_fit", fit_is_the_reference_model)
    monkeypatch.setenv("PSK_NATIVE_PKL", "1")
    ph.machine_learning_modelling(None)
    assert len(fits) == 1
    for fn in ("summary_of_%s_analysis_%s.txt" % (short, pheno), "k-mers_and_coefficients_in_%s_model_%s.txt" % (short, pheno)):
        got, want = open(fn).read(), open(os.path.join(ref_dir, fn)).read()
        if got != want:
            # ONE line may differ, and only where the reference itself is not reproducible: "Average precision" ranks the
            # samples by predict_proba, whose BLAS sum over liblinear's ~100 non-zero near-duplicate coefficients differs in the
            # last bit between a C- and a Fortran-ordered X (the reference hands over a DataFrame whose layout depends on how
            # it was built); scores that are equal in exact arithmetic then tie or not.  Both layouts' values are computed
            # here with scikit-learn: the reference's line and this package's must each be one of them.
            from sklearn.metrics import average_precision_score
            g, w = got.splitlines(), want.splitlines()
            assert len(g) == len(w), fn
            diff = [i for i in range(len(g)) if g[i] != w[i]]
            assert diff and all(g[i].startswith("Average precision: ") for i in diff), (fn, [(g[i], w[i]) for i in diff][:3])
            X = np.asarray(ph.ML["X"], dtype=np.float64)
            y = np.asarray(ph.ML["phenotype"], dtype=np.int64)
            assert len(diff) == 1 and X.shape[0] == len(y)          # (whole-set layout: one report)
            ap = {"Average precision: %s" % np.float64(average_precision_score(y, pkg["model"].predict_proba(lay(X))[:, 1])).round(2)
                  for lay in (np.ascontiguousarray, np.asfortranarray)}
            assert len(ap) == 2 and g[diff[0]] in ap and w[diff[0]] in ap, (ap, g[diff[0]], w[diff[0]])
    assert list(pkg["kmers"]) == list(ph.ML["kmers"]) and pkg["pred_scale"] == ph.pred_scale and pkg["pca"] is False
