"""Consumes the capture-when-available pins of tools/pin_thirdparty.py (statsmodels' weighted Welch test at
non-integer weights; Biopython NJ + newick writer + ete3 + GSC weights): this repo's restatements against the real
libraries' outputs.  The files exist only once somebody has run the script where those libraries import -- they are not
installed in the build container or on the GPU box -- so each test skips while its file is absent (the same pattern as
tools/cfg1_repro.py / test_cfg1_example_dataset).  The plumbing itself is exercised on every run by the self-check at the
bottom, which feeds the same code a file written from this repo's own restatement."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN

PINS = os.environ.get("PSK_PINS_DIR", GOLDEN)


def _load(name, where=None):
    path = os.path.join(where or PINS, name)
    if not os.path.exists(path):
        pytest.skip("%s not captured yet: run tools/pin_thirdparty.py where statsmodels / Biopython / ete3 import" % name)
    with open(path) as f:
        return json.load(f)


def check_welch(d, oracle):
    """oracle.ttest_scan on a one-row matrix per case (the k-mer's samples first, then the others -- the order in which
    modeling.py:743-757 appends them) against statsmodels' t, p and the two weighted means.  The HIP scan is
    bit-identical to this oracle function (tests/test_gpu_parity.py::test_ttest_scan_vs_oracle)."""
    assert len(d["cases"]) >= 10
    for c in d["cases"]:
        x, y, xw, yw = (np.array(c[k_], dtype=np.float64) for k_ in ("x", "y", "xw", "yw"))
        n = len(x) + len(y)
        wpr = (n + 63) // 64
        bits = np.zeros((1, wpr), dtype=np.uint64)
        for i in range(len(x)):
            bits[0, i >> 6] |= np.uint64(1) << np.uint64(i & 63)
        ref = oracle.ttest_scan(bits, np.concatenate([x, y]).tolist(), np.concatenate([xw, yw]), n, 1, n, 2.0, 1)
        assert ref["n_with"][0] == len(x)
        if np.isfinite(c["t"]):
            assert ref["stat"][0] == pytest.approx(c["t"], rel=1e-12, abs=1e-300), c
            assert ref["p"][0] == pytest.approx(c["p"], rel=1e-9, abs=1e-300), c
        assert ref["mean_x"][0] == pytest.approx(c["mean_x"], rel=1e-14) and ref["mean_y"][0] == pytest.approx(c["mean_y"], rel=1e-14)


def check_nj(d):
    """weights.nj -> newick_round_trip -> to_newick equals the string Biopython wrote (tie-breaking, rooting, child
    order, "%1.5f" branch lengths), and gsc_weights over it equals the weights computed over ete3's parse of it."""
    from phenotypeseeker_amd import weights as W
    assert len(d["cases"]) >= 10
    for c in d["cases"]:
        names, n = c["names"], len(c["names"])
        mat = np.zeros((n, n))
        for i, row in enumerate(c["lower"]):
            for j, v in enumerate(row):
                mat[i, j] = mat[j, i] = v
        tree = W.newick_round_trip(W.nj(list(names), mat.tolist()))
        assert W.to_newick(tree) == c["newick"], names
        got = W.gsc_weights(tree)
        assert set(got) == set(c["weights"])
        for nm, v in c["weights"].items():
            assert got[nm] == pytest.approx(v, rel=1e-12), nm


def test_weighted_welch_against_statsmodels(oracle):
    check_welch(_load("welch_w_kat.json"), oracle)


def test_nj_newick_gsc_against_biopython_and_ete3():
    check_nj(_load("nj_gsc_kat.json"))


def test_pin_plumbing_on_self_generated_files(tmp_path, oracle):
    """The two checks above run here on files in the captured FORMAT whose outputs come from this repo's own restatements
    (so they say nothing about parity): the case generators of tools/pin_thirdparty.py, the JSON layout and the
    comparison code are exercised on every run, and a wrong restatement of the format would show up before the real
    files ever arrive."""
    import importlib.util
    from helpers import ROOT
    from phenotypeseeker_amd import weights as W
    spec = importlib.util.spec_from_file_location("pin_thirdparty", os.path.join(ROOT, "tools", "pin_thirdparty.py"))
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    cases = []
    for x, y, xw, yw in pin.welch_cases():
        n = len(x) + len(y)
        wpr = (n + 63) // 64
        bits = np.zeros((1, wpr), dtype=np.uint64)
        for i in range(len(x)):
            bits[0, i >> 6] |= np.uint64(1) << np.uint64(i & 63)
        r = oracle.ttest_scan(bits, np.concatenate([x, y]).tolist(), np.concatenate([xw, yw]), n, 1, n, 2.0, 1)
        cases.append({"x": x.tolist(), "y": y.tolist(), "xw": xw.tolist(), "yw": yw.tolist(), "t": float(r["stat"][0]),
                      "p": float(r["p"][0]), "df": 0.0, "mean_x": float(r["mean_x"][0]), "mean_y": float(r["mean_y"][0])})
    with open(tmp_path / "welch_w_kat.json", "w") as f:
        json.dump({"source": "self", "cases": cases}, f)
    check_welch(_load("welch_w_kat.json", str(tmp_path)), oracle)
    out = []
    for names, d in pin.nj_cases():
        tree = W.newick_round_trip(W.nj(list(names), d.tolist()))
        out.append({"names": list(names), "lower": [[float(d[i][j]) for j in range(i + 1)] for i in range(len(names))],
                    "newick": W.to_newick(tree), "weights": W.gsc_weights(tree)})
    with open(tmp_path / "nj_gsc_kat.json", "w") as f:
        json.dump({"source": "self", "cases": out}, f)
    check_nj(_load("nj_gsc_kat.json", str(tmp_path)))
    # weights of a tree sum to the number of leaves ("mean1")
    assert all(sum(c["weights"].values()) == pytest.approx(len(c["names"]), rel=1e-9) for c in out)
    found, missing = pin.probe()
    assert set(found) | {m.split()[0] for m in missing} == {"statsmodels", "Bio", "ete3"}
