#!/usr/bin/env python3
"""Runs THIS package's `phenotypeseeker modeling` (+ `prediction` on the same samples) on the golden genome sets ON THE GPU BOX
and keeps what it wrote under gpurun_out/product_run/<set>/: the .pkl, the summary, the coefficient table, predictions_*.txt.
oracle/gen_golden.py::gen_prediction_of_product_pkl then hands those .pkl files to the REFERENCE's prediction.py in the build
container (the reverse direction of the .pkl contract) and commits its predictions_*.txt next to them.
usage (GPU box, from the repo root):  python3 tools/capture_product_run.py"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import GOLDEN, load_dataset  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "product_run")
CLI = [sys.executable, os.path.join(ROOT, "scripts", "phenotypeseeker")]
KEEP = ("summary_of_", "k-mers_and_coefficients_", "predictions_", "chi2_results_", "t-test_results_")


def run(argv, cwd):
    r = subprocess.run(CLI + argv, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode:
        sys.exit("%s failed (%d): %s" % (argv, r.returncode, r.stderr.decode(errors="replace")[-2000:]))


def keep(tmp, tag):
    d = os.path.join(OUT, tag)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    for fn in sorted(os.listdir(tmp)):
        if fn.startswith(KEEP) or fn.endswith((".pkl", "_MLdf.csv")) or fn in ("samples.txt", "phenos.txt", "log.txt"):
            shutil.copy(os.path.join(tmp, fn), os.path.join(d, fn))
    print(tag, sorted(os.listdir(d)), flush=True)


for tag, flags, pheno, short in (("ds_omitB", ["--omit_B_correction", "--n_kmers", "100"], "Pheno", "log_reg"),
                                 ("ds_bonf", [], "Pheno", "log_reg"), ("ds_cont", ["-jt", "modelling"], "MIC", "linreg")):
    tmp = tempfile.mkdtemp(prefix="psk_cap_")
    src = "ds_omitB" if tag == "ds_cont" else tag           # ds_cont: ds_omitB's genomes under a continuous phenotype
    ds = load_dataset(src)
    fns = {l.split()[0]: l.split()[1] for l in open(os.path.join(ds["dir"], "data.pheno")).read().splitlines()[1:]}
    for name, data in ds["files"].items():
        with open(os.path.join(tmp, fns[name]), "wb") as f:
            f.write(data)
    for fn in ["data.pheno"] + (["MIC_MLdf.csv"] if tag == "ds_cont" else []):
        shutil.copy(os.path.join(GOLDEN, tag, fn), tmp)
    run(["modeling", "data.pheno"] + flags, tmp)
    with open(os.path.join(tmp, "samples.txt"), "w") as f:
        for name in ds["names"]:
            f.write("%s\t%s\n" % (name, fns[name]))
    with open(os.path.join(tmp, "phenos.txt"), "w") as f:
        f.write("%s\t%s_model_%s.pkl\n" % (pheno, short, pheno))
    run(["prediction", "samples.txt", "phenos.txt"], tmp)
    keep(tmp, tag)
    shutil.rmtree(tmp, ignore_errors=True)
