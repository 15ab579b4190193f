#!/usr/bin/env python3
"""BASELINE config 1: the reference's own example (/root/reference/example/test_PS_modeling.sh:12-25) -- the
C. difficile azithromycin set, `phenotypeseeker modeling PS_modeling_example_files/data.pheno`, default parameters.

The tarball (http://bioinfo.ut.ee/PhenotypeSeeker/PS_modeling_example_files.tar.gz, 174 MB) cannot be fetched in the
build container or on the GPU box (no network), so this is the recipe to run where it is at hand:

    tools/cfg1_repro.py /path/to/PS_modeling_example_files.tar.gz [--omit_B_correction] [--reference]

It unpacks the tarball, runs this package's `modeling` on it (needs the GPU) and prints the sha256 of every result
table -- the filtered k-mer list north_star asks to reproduce is chi2_results_<pheno>.tsv.  With --reference (only
where /root/reference and its bundled binaries exist: the build container) the unmodified reference is run on the same
files through oracle/ref_shim.py and the tables are compared: same rows, same p-value strings in the same order.
PSK_CFG1_TARBALL names the tarball for tests/test_gpu_e2e.py::test_cfg1_example_dataset, which skips without it.

The GPU and the reference never meet in one place here (the GPU box has no /root/reference, the build container no GPU),
so the two halves can run apart:
    GPU box:          tools/cfg1_repro.py T.tar.gz --save-product DIR      (this package's tables are kept in DIR)
    build container:  tools/cfg1_repro.py T.tar.gz --reference --product-from DIR
r05: executed end to end that way on a stand-in tarball of the same layout (tools/make_cfg1_standin.py: the 60 AT-rich
1-Mbp genomes of tests/golden/ds_atrich under PS_modeling_example_files/, data.pheno addressing them relative to the
directory the tarball is unpacked in) -- profiles/r05_cfg1_standin.txt.
"""
import argparse
import hashlib
import os
import shutil
import subprocess
import sys
import tarfile
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TABLES = ("chi2_results_", "_MLdf.csv")


def unpack(tarball, where):
    with tarfile.open(tarball) as t:
        t.extractall(where)
    for dirpath, _, files in os.walk(where):
        if "data.pheno" in files:
            return dirpath
    raise SystemExit("no data.pheno inside %s" % tarball)


def absolutise(pheno_path, data_dir, out_path):
    """data.pheno lists the FASTA files relative to the directory the reference is run from; rewrite them absolute."""
    rows = []
    with open(pheno_path) as f:
        for ln, line in enumerate(f):
            fields = line.rstrip("\n").split("\t") if "\t" in line else line.split()
            if ln and len(fields) > 1 and not os.path.isabs(fields[1]):
                for base in (os.path.dirname(data_dir), data_dir):
                    cand = os.path.join(base, fields[1])
                    if os.path.exists(cand):
                        fields[1] = cand
                        break
            rows.append("\t".join(fields))
    with open(out_path, "w") as f:
        f.write("\n".join(rows) + "\n")


def tables(d):
    out = {}
    for fn in sorted(os.listdir(d)):
        if fn.startswith(TABLES[0]) or fn.endswith(TABLES[1]):
            with open(os.path.join(d, fn), "rb") as f:
                out[fn] = f.read()
    return out


def rows_of(blob):
    lines = blob.decode().splitlines()
    return lines[0], [tuple(l.split("\t")) for l in lines[1:]]


def run_product(data_pheno, workdir, flags):
    from phenotypeseeker_amd.cli import build_parser
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        args = build_parser().parse_args(["modeling", data_pheno] + flags)
        t0 = time.time()
        args.func(args)
        return time.time() - t0
    finally:
        os.chdir(cwd)


def run_reference(data_pheno, workdir, flags, nt=8):
    env = dict(os.environ, PATH="/root/reference/bin:" + os.environ.get("PATH", ""))
    t0 = time.time()
    subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_shim.py"), "modeling", data_pheno, "-nt", str(nt)] + flags,
                   cwd=workdir, env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return time.time() - t0


def compare(ours, theirs):
    """Same tables, same rows, same p-value strings in the same order (ties inside one p-value string are ordered
    by the reference's unstable sort and its -nt; see DESIGN.md 'Exactness strategy')."""
    ok = True
    for fn in sorted(set(ours) | set(theirs)):
        if fn not in ours or fn not in theirs:
            print("MISSING %s in %s" % (fn, "product" if fn not in ours else "reference"))
            ok = False
            continue
        if fn.endswith(".csv"):
            a, b = ours[fn].decode().splitlines(), theirs[fn].decode().splitlines()
            same = [r.split(",")[0] for r in a] == [r.split(",")[0] for r in b] and \
                   [r.split(",")[-2:] for r in a] == [r.split(",")[-2:] for r in b]
        else:
            (ha, ra), (hb, rb) = rows_of(ours[fn]), rows_of(theirs[fn])
            same = ha == hb and [r[2] for r in ra] == [r[2] for r in rb] and \
                   (sorted(ra) == sorted(rb) or "_top" in fn)
        print("%-50s %s  product sha256 %s  reference sha256 %s" % (fn, "EQUAL" if same else "DIFFERENT",
              hashlib.sha256(ours[fn]).hexdigest()[:16], hashlib.sha256(theirs[fn]).hexdigest()[:16]))
        ok &= same
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tarball")
    ap.add_argument("--omit_B_correction", action="store_true")
    ap.add_argument("--reference", action="store_true", help="also run /root/reference through oracle/ref_shim.py and compare")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--save-product", metavar="DIR", help="keep this package's result tables in DIR (GPU half of a split run)")
    ap.add_argument("--product-from", metavar="DIR", help="take this package's tables from DIR instead of running it (reference half)")
    a = ap.parse_args()
    flags = ["--omit_B_correction"] if a.omit_B_correction else []
    tmp = tempfile.mkdtemp(prefix="psk_cfg1_")
    try:
        data_dir = unpack(a.tarball, os.path.join(tmp, "in"))
        ours_dir, ref_dir = os.path.join(tmp, "ours"), os.path.join(tmp, "ref")
        os.makedirs(ours_dir)
        absolutise(os.path.join(data_dir, "data.pheno"), data_dir, os.path.join(tmp, "data.pheno"))
        if a.product_from:
            ours = tables(a.product_from)
            print("product: %d tables taken from %s" % (len(ours), a.product_from))
        else:
            wall = run_product(os.path.join(tmp, "data.pheno"), ours_dir, flags)
            ours = tables(ours_dir)
            print("product: %.2f s, %d tables" % (wall, len(ours)))
        if a.save_product:
            os.makedirs(a.save_product, exist_ok=True)
            for fn, blob in ours.items():
                with open(os.path.join(a.save_product, fn), "wb") as f:
                    f.write(blob)
        for fn, blob in ours.items():
            print("  %-50s %8d rows  sha256 %s" % (fn, blob.count(b"\n") - 1, hashlib.sha256(blob).hexdigest()))
        if a.reference:
            if not os.path.isdir("/root/reference/bin"):
                raise SystemExit("--reference needs /root/reference (the build container)")
            os.makedirs(ref_dir)
            wall = run_reference(os.path.join(tmp, "data.pheno"), ref_dir, flags)
            print("reference: %.1f s" % wall)
            if not compare(ours, tables(ref_dir)):
                raise SystemExit(1)
            print("config 1 reproduced")
    finally:
        if a.keep:
            print("kept", tmp)
        else:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
