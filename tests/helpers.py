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
