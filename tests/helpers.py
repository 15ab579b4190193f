"""Shared helpers for the parity tests: golden-dataset loading and TSV parsing."""
import base64
import gzip
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_dataset(tag):
    """Returns dict(meta, names, files{name: bytes}, pheno[list of 1/0/'NA'], dir)."""
    d = os.path.join(GOLDEN, tag)
    with open(os.path.join(d, "meta.json")) as f:
        meta = json.load(f)
    names, files, pheno = [], {}, []
    with open(os.path.join(d, "data.pheno")) as f:
        f.readline()
        for line in f:
            if not line.strip():
                continue
            name, fn, ph = line.split()
            names.append(name)
            with gzip.open(os.path.join(d, fn + ".gz"), "rb") as g:
                files[name] = g.read()
            pheno.append("NA" if ph == "NA" else int(ph))
    return {"meta": meta, "names": names, "files": files, "pheno": pheno, "dir": d}


def read_results_tsv(path):
    """chi2_results_*.tsv -> list of (kmer, stat_text, p_text, n_text, names_text) in file order."""
    rows = []
    with open(path) as f:
        header = f.readline().rstrip("\n").split("\t")
        for line in f:
            rows.append(tuple(line.rstrip("\n").split("\t")))
    return header, rows


def tokenizer_cases():
    with open(os.path.join(GOLDEN, "tokenizer_cases.json")) as f:
        cases = json.load(f)["cases"]
    out = []
    for c in cases:
        out.append((base64.b64decode(c["input_b64"]), c["k"],
                    None if c["list_b64"] is None else base64.b64decode(c["list_b64"])))
    return out


def fmt_float_like_pandas(x):
    """How pandas.to_csv prints the object-dtype float the reference stores (repr of float)."""
    return repr(float(x))


def large_design(z, tag):
    """A design of tests/golden/model_large_kat.npz (oracle/gen_golden.py::gen_model_large_kat): (X float32 [n][p], y int32,
    fold int32).  'g' is the recorded 2,048 x 907 design of fit2048_907.npz, 'h' and 'i' are stored bit-packed."""
    import numpy as np
    n = int(z["n_" + tag])
    if tag == "g":
        d = np.load(os.path.join(GOLDEN, "fit2048_907.npz"))
        X = np.unpackbits(d["Xbits"], axis=1)[:, : int(d["p"])]
        y = d["y"]
    else:
        X = np.unpackbits(z["X_" + tag], axis=0)[:n]
        y = z["y_" + tag]
    return X.astype(np.float32), y.astype(np.int32), z["fold_" + tag].astype(np.int32)


def _num(tok):
    t = tok.strip("(),:;[]{}'")
    if t.startswith("+/-"):          # "(+/-0.234)" of a grid-score line
        t = t[3:]
    try:
        return float(t)
    except ValueError:
        return None


def assert_text_equal_up_to_numbers(got, want, rel=1e-6, abs_=1e-9, what=""):
    """Line by line, token by token: text tokens identical, numeric tokens within rel / abs_ (nan == nan)."""
    import math
    g, w = got.splitlines(), want.splitlines()
    assert len(g) == len(w), "%s: %d lines against %d" % (what, len(g), len(w))
    for i, (a, b) in enumerate(zip(g, w)):
        if a == b:
            continue
        ta, tb = a.split(), b.split()
        assert len(ta) == len(tb), "%s line %d: %r != %r" % (what, i + 1, a, b)
        for x, y in zip(ta, tb):
            if x == y:
                continue
            fx, fy = _num(x), _num(y)
            assert fx is not None and fy is not None, "%s line %d: %r != %r" % (what, i + 1, a, b)
            if math.isnan(fx) and math.isnan(fy):
                continue
            assert abs(fx - fy) <= abs_ + rel * abs(fy), "%s line %d: %r != %r (%s vs %s)" % (what, i + 1, a, b, x, y)


def mask_numbers(line):
    """Every numeric token of a line replaced by '#': what is left of a summary line when liblinear's unseeded,
    unconverged coefficients decide its numbers."""
    return " ".join("#" if _num(t) is not None else t for t in line.split())
