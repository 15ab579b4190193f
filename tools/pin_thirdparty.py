#!/usr/bin/env python3
"""Capture-when-available pins for the third-party arithmetic of the `-w` / continuous path (BASELINE config 4) that is
NOT installed in the build container or on the GPU box (missing #3):

  * statsmodels  `ttest_ind(x, y, usevar='unequal', weights=(xw, yw))` with NON-INTEGER weights -- the call of
                 /root/reference/PhenotypeSeeker/modeling.py:44,734 -- and `np.average(..., weights=)` (:735-736)
  * Biopython    `DistanceTreeConstructor().nj(_DistanceMatrix(names, lower_triangle))`, `Bio.Phylo.write(tree, f,
                 "phyloxml")`, `Bio.Phylo.convert(..., "phyloxml", ..., "newick")`       (modeling.py:447-458)
  * ete3         `Tree(newick, format=1)` and the traversal the GSC recursion runs on      (modeling.py:461-503)

Run it wherever those three import (any machine: no GPU, no reference checkout needed):

    python tools/pin_thirdparty.py            # writes tests/golden/welch_w_kat.json and tests/golden/nj_gsc_kat.json

The inputs are generated here from fixed seeds and stored IN the files, so the tests (tests/test_thirdparty_pins.py)
need nothing but the files: they hold this repo's restatements (oracle/psk_oracle.c orc_ttest_row, the formula the HIP
scan is bit-identical to; phenotypeseeker_amd/weights.py nj / to_newick / gsc_weights) to the captured outputs, and
skip while the files are absent.  The day the files are committed, SURVEY.md 8(c)'s "parity unpinned" for a7's
non-integer weights and for f2's tree step flips to "pinned".  The GSC arithmetic itself is the reference's own code
(not third-party); here it runs over real ete3 nodes so that child order and rooting are the library's.
With --check the files are not written: the script only says what is importable."""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def probe():
    found, missing = {}, []
    for mod in ("statsmodels", "Bio", "ete3"):
        try:
            m = __import__(mod)
            found[mod] = getattr(m, "__version__", "?")
        except Exception as e:  # noqa: BLE001
            missing.append("%s (%s)" % (mod, type(e).__name__))
    return found, missing


def welch_cases():
    """(x, y, xw, yw): group sizes from 2 to 1,200, weights like GSC weights (mean ~1, non-integer), values like MICs."""
    rng = np.random.default_rng(20260003)
    cases = []
    for nx, ny in ((2, 2), (2, 9), (3, 3), (5, 40), (17, 23), (64, 64), (100, 7), (250, 300), (1200, 848)):
        for spread in (0.05, 0.6):
            x = np.round(rng.normal(3.0, 1.5, nx), 4)
            y = np.round(rng.normal(2.6, 1.1, ny), 4)
            xw = np.round(np.exp(rng.normal(0.0, spread, nx)), 6)
            yw = np.round(np.exp(rng.normal(0.0, spread, ny)), 6)
            cases.append((x, y, xw, yw))
    # equal values inside a group (zero variance on one side), and very unequal weights
    cases.append((np.array([2.0, 2.0, 2.0]), np.array([1.0, 3.5, 2.25, 8.0]), np.array([0.5, 1.25, 2.0]), np.array([1.1, 0.9, 1.0, 0.3])))
    cases.append((np.array([0.125, 8.0, 4.0, 2.0]), np.array([1.0, 1.5, 64.0]), np.array([1e-3, 5.0, 0.7, 1.3]), np.array([2.5, 0.01, 0.4])))
    return cases


def capture_welch():
    from statsmodels.stats.weightstats import ttest_ind
    import statsmodels
    out = []
    for x, y, xw, yw in welch_cases():
        t, p, df = ttest_ind(x, y, usevar="unequal", weights=(xw, yw))
        out.append({"x": x.tolist(), "y": y.tolist(), "xw": xw.tolist(), "yw": yw.tolist(), "t": float(t), "p": float(p),
                    "df": float(df), "mean_x": float(np.average(x, weights=xw)), "mean_y": float(np.average(y, weights=yw))})
    return {"source": "statsmodels.stats.weightstats.ttest_ind(x, y, usevar='unequal', weights=(xw, yw)) + numpy.average "
                      "(modeling.py:734-736)", "statsmodels": statsmodels.__version__, "numpy": np.__version__, "cases": out}


def nj_cases():
    """(names, full symmetric matrix): additive trees, random matrices with the 6-significant-digit values `mash dist`
    prints, ties, identical samples (distance 0), 3 ... 60 leaves; names as the reference has them (data.pheno order)."""
    rng = np.random.default_rng(20260004)
    cases = []
    for n in (3, 4, 5, 8, 13, 24, 60):
        for kind in ("mash", "ties", "additive"):
            names = ["S%04d" % i for i in rng.permutation(n)]
            if kind == "additive":
                pos = np.sort(rng.random(n))
                d = np.abs(pos[:, None] - pos[None, :]) + 0.01 * (1 - np.eye(n))
            else:
                d = rng.random((n, n)) * 0.05
                d = (d + d.T) / 2
            if kind == "ties":
                d = np.round(d, 2)
                if n > 4:
                    d[1, 0] = d[0, 1] = 0.0   # two identical samples
            d = np.array([[float("%g" % v) for v in row] for row in d])
            np.fill_diagonal(d, 0.0)
            cases.append((names, d))
    return cases


def _gsc_over_ete3(newick):
    """clip_branch_lengths / set_branch_sum / set_node_weight / mean1 of modeling.py:461-503, over real ete3 nodes."""
    from ete3 import Tree
    tree = Tree(newick, format=1)
    for nd in tree.traverse("levelorder"):
        nd.dist = min(max(nd.dist, 1e-9), 1e9)

    def branch_sum(nd):
        total = 0
        for ch in nd.get_children():
            branch_sum(ch)
            total += ch.BranchSum
            total += ch.dist
        nd.BranchSum = total

    def node_weight(nd):
        nd.NodeWeight = 1.0 if nd.up is None else nd.up.NodeWeight * (nd.dist + nd.BranchSum) / nd.up.BranchSum
        for ch in nd.get_children():
            node_weight(ch)
    branch_sum(tree)
    node_weight(tree)
    w = {leaf.name: leaf.NodeWeight for leaf in tree.iter_leaves()}
    return {k: v * len(w) for k, v in w.items()}


def capture_nj():
    import Bio
    import Bio.Phylo
    import ete3
    from Bio.Phylo.TreeConstruction import DistanceTreeConstructor, _DistanceMatrix
    out = []
    for names, d in nj_cases():
        lower = [[float(d[i][j]) for j in range(i + 1)] for i in range(len(names))]
        tree = DistanceTreeConstructor().nj(_DistanceMatrix(list(names), [row[:] for row in lower]))
        with tempfile.TemporaryDirectory() as tmp:
            xml, nwk = os.path.join(tmp, "tree_xml.txt"), os.path.join(tmp, "tree_newick.txt")
            with open(xml, "w+") as f:
                Bio.Phylo.write(tree, f, "phyloxml")
            with open(nwk, "w+") as f:
                Bio.Phylo.convert(xml, "phyloxml", f, "newick")
            newick = open(nwk).read().strip()
        out.append({"names": list(names), "lower": lower, "newick": newick, "weights": _gsc_over_ete3(newick)})
    return {"source": "Bio.Phylo.TreeConstruction.DistanceTreeConstructor().nj -> Bio.Phylo.write(phyloxml) -> "
                      "Bio.Phylo.convert(newick) -> ete3.Tree(format=1) -> GSC recursion (modeling.py:447-503)",
            "biopython": Bio.__version__, "ete3": ete3.__version__, "cases": out}


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--check", action="store_true", help="only report which libraries import")
    ap.add_argument("--out", default=GOLD)
    args = ap.parse_args()
    found, missing = probe()
    print("importable: %s" % (", ".join("%s %s" % kv for kv in found.items()) or "none"))
    if missing:
        print("missing:    %s" % ", ".join(missing))
    if args.check:
        return 0 if not missing else 3
    wrote = []
    if "statsmodels" in found:
        with open(os.path.join(args.out, "welch_w_kat.json"), "w") as f:
            json.dump(capture_welch(), f)
        wrote.append("welch_w_kat.json")
    if "Bio" in found and "ete3" in found:
        with open(os.path.join(args.out, "nj_gsc_kat.json"), "w") as f:
            json.dump(capture_nj(), f)
        wrote.append("nj_gsc_kat.json")
    print("wrote: %s" % (", ".join(wrote) or "nothing (no library to capture from)"))
    return 0 if wrote else 3


if __name__ == "__main__":
    sys.exit(main())
