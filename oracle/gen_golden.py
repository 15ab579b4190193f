"""Generates tests/golden/ from the reference itself.  RUNS IN THE BUILD CONTAINER ONLY
(needs /root/reference; the GPU box never sees it -- only the committed fixtures travel).

What is captured (data only: inputs and the reference's outputs; no reference source):
  tokenizer_cases.json   crafted + fuzzed FASTA/FASTQ byte strings and the .list file that
                         /root/reference/bin/glistmaker 4.2.3 wrote for each (base64)
  ds_omitB/, ds_bonf/    small synthetic genome sets (FASTA, gz) + data.pheno and, from the
                         reference run through oracle/ref_shim.py: per-sample .list digests,
                         the union k-mer words, one glistquery mapping, chi2_results TSVs,
                         <pheno>_MLdf.csv
  chi2_kat.json          phenotypes.conduct_chi_squared_test called directly on random rows
                         (weights, NA, filters, Bonferroni) -> its return value
  welch_kat.json         scipy.stats.ttest_ind(equal_var=False) on random rows, unit weights
                         and integer frequency weights (replicated observations)
  model_kat.npz/json     sklearn liblinear L1 logistic regression and Lasso, converged
                         (tol 1e-10), plus StratifiedKFold/KFold splits and CV scores
  gmer_counter.json      /root/reference/bin/gmer_counter outputs for the prediction path
  mash.json              /root/reference/bin/mash sketches (hash lists) and `mash dist` table (-w path)

Usage:  python oracle/gen_golden.py   (takes ~1 min)
"""
import base64
import gzip
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REFBIN = "/root/reference/bin"
ENV = dict(os.environ, PATH=REFBIN + ":" + os.environ["PATH"])

from phenotypeseeker_amd.synth import GenomeSet, fastq_reads  # noqa: E402


def sh(cmd, cwd, stdout=None):
    return subprocess.run(cmd, shell=True, cwd=cwd, env=ENV, stdout=stdout or subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)


def b64(b):
    return base64.b64encode(b).decode()


def glistmaker_list(data, k, tmp):
    """Run the reference binary on `data`; return the .list bytes or None."""
    p = os.path.join(tmp, "in.dat")
    with open(p, "wb") as f:
        f.write(data)
    out = os.path.join(tmp, "out_%d.list" % k)
    if os.path.exists(out):
        os.remove(out)
    sh("glistmaker in.dat -o out -w %d" % k, tmp)
    if not os.path.exists(out):
        return None
    with open(out, "rb") as f:
        return f.read()


# ------------------------------------------------------------------------------------------
def gen_tokenizer_cases():
    crafted = [
        (b">r1 desc\nACGTACGTAC\nGTNACGTTGCA\n\nacgtu\r\nGGGCC\n>r2\nAC-GT\n", 5),
        (b">r1\nACGTA>CGTAC\nACGTACC\n", 5),
        (b"ACGTACGTAC\n>r\nCCCCCC", 5),
        (b">r\nACGTA CGTAC\nAC\tGTACC\n", 5),
        (b">r\nACGTAC\n@notheader\nGTACC\n", 5),
        (b">r\nACGTAC\n  >x\nGTACCA\n", 5),
        (b"\n\n>r\nACGTAC\n", 5),
        (b"\n>r\nACGTAC\n", 5),
        (b"XX@q\nACGTAC\n+\nIIIIII\n", 5),
        (b"XX>r\nACGTAC\n", 5),
        (b">r\nACGTA\x00CGTAC\nACGTAC\n", 5),
        (b">r\nACGTAC\x0bCGT\x01AC\n", 5),
        (b">r1\n>r2\nACGTAC\n", 5),
        (b">r1", 5),
        (b"", 5),
        (b">r1\nACGTAC\n>", 5),
        (b">r1\nACG\n\n\nTAC\n", 5),
        (b">r\nACGT\n", 5),
        (b">pal\nACGTACGT\n", 4),
        (b">pal\nAATTAATT\n", 4),
        (b">k1\nACGTNACGT\n", 1),
        (b">k32\n" + b"ACGTTGCAAGCTTAGCCGATCGATTAGCAGCTAGCTAGGATCCAAGTC" + b"\n", 32),
        (b">k31\n" + b"ACGTTGCAAGCTTAGCCGATCGATTAGCAGCTAGCTAGGATCCAAGTC" + b"\n", 31),
        (b">allT\n" + b"T" * 40 + b"\n", 13),
        (b">allA\n" + b"A" * 40 + b"\n", 16),
        (b"@q1\nACGTACGTAC\n+\n@CGTACGTAC\n@q2\nTTTTTGGGGG\n+q2\nACGTACGTAC\n", 5),
        (b"@q1\nACGTA\nCGTAC\n+\nIIIII\nIIIII\n@q2\nTTTTTGGGGG\n+\nIIIIIIIIII\n", 5),
        (b"@q1\nACGTACGTAC\n+\nIIIIIIIIII\n\n@q2\nTTTTTGGGGG\n+\nIIIIIIIIII", 5),
        (b"@q1\nACGTAC\n+\nIIIIII\n\n\n@q2\nTTTTTG\n+\nIIIIII\n", 5),
        (b"@q1\nACGTAC\n+\nIIIIII\nXX\n@q2\nTTTTTG\n+\nIIIIII\n", 5),
        (b"@q1\nACGTAC\n+\n\n@q2\nTTTTTG\n+\nIIIIII\n", 5),
        (b"@q1\n\n+\n\n@q2\nTTTTTG\n+\nIIIIII\n", 5),
        (b"@q1\r\nACGTAC\r\n+\r\nIIIIII\r\n@q2\r\nTTTTTG\r\n+\r\nIIIIII\r\n", 5),
        (b"@q1\nACGTAC\n+\nII\n@III\n@q2\nTTTTTG\n+\nIIIIII\n", 5),
        (b"@q\nACGTAC\n+\nIIIIII\n>r\nCCCCCC\n", 5),
        (b"@r\nACGTAC>ACGTAC\n+\nIIIIII\n", 5),
        (b">r\nACGTAC\n@q\nCCCCCC\n+\nGGGGGG\n", 5),
        (b"@q\nACGTAC\n@q2\nCCCCCC\n", 5),
        (b"@q1\nACG\n\nTAC\n+\nIIIIII\n", 5),
        (b"@q\nACGTACG", 5),
        (b"@q\nACGTAC\n+", 5),
    ]
    rnd = random.Random(20260101)
    fuzz = []
    for _ in range(120):
        k = rnd.choice([1, 2, 3, 5, 7, 13, 16, 31, 32])
        mode = rnd.choice(["fa", "fq", "mix"])
        n = rnd.randint(0, 300)
        if mode == "fa":
            s = ">h\n" + "".join(rnd.choice("ACGT" * 6 + "acgtuNn>@+ \t\r\n\n-") for _ in range(n))
        elif mode == "fq":
            recs = []
            for r in range(rnd.randint(1, 5)):
                L = rnd.randint(0, 70)
                seq = "".join(rnd.choice("ACGT" * 8 + "Nn") for _ in range(L))
                q = "".join(rnd.choice("I@+>AC#") for _ in range(L))
                recs.append("@r%d\n%s\n+\n%s\n" % (r, seq, q))
                if rnd.random() < 0.2:
                    recs.append(rnd.choice(["\n", "\n\n", "XX\n", "@\n", ">r\nACGTACGTAGCTAGCTAGCATCGATCGA\n"]))
            s = "".join(recs)
            if rnd.random() < 0.3:
                s = s.rstrip("\n")
        else:
            s = "".join(rnd.choice("ACGT" * 4 + "acgtu>@+ \t\r\n\n-NI") for _ in range(n))
        fuzz.append((s.encode(), k))
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for data, k in crafted + fuzz:
            lst = glistmaker_list(data, k, tmp)
            cases.append({"k": k, "input_b64": b64(data), "list_b64": None if lst is None else b64(lst)})
    with open(os.path.join(GOLD, "tokenizer_cases.json"), "w") as f:
        json.dump({"source": "glistmaker 4.2.3 (reference bin/)", "cases": cases}, f)
    print("tokenizer cases:", len(cases))


# ------------------------------------------------------------------------------------------
def run_reference_dataset(tag, gs, na, flags, k=13, nt=4, fastq_for=()):
    """Write the dataset, run the reference pipeline on it, capture the artefacts."""
    out = os.path.join(GOLD, tag)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    tmp = tempfile.mkdtemp(prefix="psk_gold_")
    rows = ["ID\tAddresses\tPheno"]
    names = []
    for i in range(gs.n):
        name, fa = gs.sample(i)
        names.append(name)
        if i in fastq_for:  # this sample is handed over as FASTQ reads instead of an assembly
            fa = fastq_reads(gs.codes(i), n_reads=600, read_len=100, seed=[gs.seed, 77, i])
            fn = name + ".fastq"
        else:
            fn = name + ".fasta"
        with open(os.path.join(tmp, fn), "wb") as f:
            f.write(fa)
        with gzip.GzipFile(os.path.join(out, fn + ".gz"), "wb", mtime=0) as f:
            f.write(fa)
        ph = "NA" if i in na else gs.phenotype(i)
        rows.append("%s\t%s\t%s" % (name, fn, ph))
    pheno_txt = "\n".join(rows) + "\n"
    with open(os.path.join(tmp, "data.pheno"), "w") as f:
        f.write(pheno_txt)
    with open(os.path.join(out, "data.pheno"), "w") as f:
        f.write(pheno_txt)

    # k-mer plane artefacts straight from the binaries (same commands as modeling.py:308-311,
    # :376-379, :324-329)
    meta = {"k": k, "flags": flags, "nt": nt, "lists": {}}
    os.makedirs(os.path.join(tmp, "L"))
    for line in rows[1:]:
        name, fn, _ = line.split("\t")
        sh("glistmaker %s -o L/%s_0 -w %d -c 1" % (fn, name, k), tmp)
        with open(os.path.join(tmp, "L", "%s_0_%d.list" % (name, k)), "rb") as f:
            data = f.read()
        h64 = np.frombuffer(data, dtype="<u8", count=3, offset=16)
        meta["lists"][name] = {"sha256": hashlib.sha256(data).hexdigest(), "n_unique": int(h64[0]),
                               "n_total": int(h64[1])}
        if name == names[0]:
            with open(os.path.join(out, "%s_0_%d.list" % (name, k)), "wb") as f:
                f.write(data)
    all_lists = " ".join("L/%s_0_%d.list" % (n, k) for n in names)
    # glistcompare takes 2-3 lists at a time in the reference's tree; the union of all is the
    # same set, so build it pairwise here
    cur = "L/%s_0_%d.list" % (names[0], k)
    for j, n in enumerate(names[1:]):
        sh("glistcompare -u -o L/u%d %s L/%s_0_%d.list" % (j, cur, n, k), tmp)
        cur = "L/u%d_%d_union.list" % (j, k)
    with open(os.path.join(tmp, cur), "rb") as f:
        udata = f.read()
    urec = np.frombuffer(udata, dtype=np.dtype([("word", "<u8"), ("freq", "<u4")]),
                         count=int(np.frombuffer(udata, dtype="<u8", count=1, offset=16)[0]), offset=40)
    np.save(os.path.join(out, "union_words.npy"), urec["word"])
    np.save(os.path.join(out, "union_freqs.npy"), urec["freq"])
    meta["union_sha256"] = hashlib.sha256(udata).hexdigest()
    meta["n_union"] = int(len(urec))
    with open(os.path.join(tmp, "map0.txt"), "wb") as f:
        sh("glistquery L/%s_0_%d.list -l %s" % (names[1], k, cur), tmp, stdout=f)
    with open(os.path.join(tmp, "map0.txt"), "rb") as f:
        meta["mapped_sample"] = names[1]
        meta["mapped_sha256"] = hashlib.sha256(f.read()).hexdigest()
    del all_lists

    # the reference pipeline itself
    with open(os.path.join(tmp, "stderr.txt"), "w") as err:
        subprocess.run([sys.executable, os.path.join(HERE, "ref_shim.py"), "modeling", "data.pheno", "-nt", str(nt),
                        "-l", str(k)] + flags, cwd=tmp, env=ENV, stderr=err, stdout=err)
    for fn in sorted(os.listdir(tmp)):
        if fn.startswith("chi2_results_") or fn.endswith("_MLdf.csv"):
            shutil.copy(os.path.join(tmp, fn), os.path.join(out, fn))
    meta["reference_outputs"] = sorted(fn for fn in os.listdir(out) if fn.startswith("chi2_") or fn.endswith(".csv"))
    with open(os.path.join(out, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)
    print(tag, "M =", meta["n_union"], "outputs:", meta["reference_outputs"])



def run_reference_dataset_regenerated(tag, synth_kw, na, flag_sets, k=13, nt=8):
    """A larger set whose inputs are NOT stored: the genomes come from phenotypeseeker_amd.synth with the parameters
    kept in meta.json (the tests regenerate them and check their sha256), only what the reference made of them is
    committed -- list hashes from glistmaker, the union from glistcompare, one glistquery mapping, and per flag set the
    result tables of the unmodified modeling.py.  Used for the AT-rich, multi-contig set (29 % GC: the reference's
    example organism, C. difficile)."""
    out = os.path.join(GOLD, tag)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    tmp = tempfile.mkdtemp(prefix="psk_gold_")
    gs = GenomeSet(**synth_kw)
    rows = ["ID\tAddresses\tPheno"]
    names, inputs = [], {}
    for i in range(gs.n):
        name, fa = gs.sample(i)
        names.append(name)
        fn = name + ".fasta"
        with open(os.path.join(tmp, fn), "wb") as f:
            f.write(fa)
        inputs[name] = hashlib.sha256(fa).hexdigest()
        rows.append("%s\t%s\t%s" % (name, fn, "NA" if i in na else gs.phenotype(i)))
    pheno_txt = "\n".join(rows) + "\n"
    for d in (tmp, out):
        with open(os.path.join(d, "data.pheno"), "w") as f:
            f.write(pheno_txt)
    meta = {"k": k, "nt": nt, "synth": synth_kw, "inputs_sha256": inputs, "lists": {}, "runs": {}}
    os.makedirs(os.path.join(tmp, "L"))
    for name in names:
        sh("glistmaker %s.fasta -o L/%s_0 -w %d -c 1" % (name, name, k), tmp)
        with open(os.path.join(tmp, "L", "%s_0_%d.list" % (name, k)), "rb") as f:
            data = f.read()
        h64 = np.frombuffer(data, dtype="<u8", count=3, offset=16)
        meta["lists"][name] = {"sha256": hashlib.sha256(data).hexdigest(), "n_unique": int(h64[0]), "n_total": int(h64[1])}
    cur = "L/%s_0_%d.list" % (names[0], k)
    for j, n in enumerate(names[1:]):
        sh("glistcompare -u -o L/u%d %s L/%s_0_%d.list" % (j, cur, n, k), tmp)
        cur = "L/u%d_%d_union.list" % (j, k)
    with open(os.path.join(tmp, cur), "rb") as f:
        udata = f.read()
    urec = np.frombuffer(udata, dtype=np.dtype([("word", "<u8"), ("freq", "<u4")]),
                         count=int(np.frombuffer(udata, dtype="<u8", count=1, offset=16)[0]), offset=40)
    meta["n_union"] = int(len(urec))
    meta["union_words_sha256"] = hashlib.sha256(np.ascontiguousarray(urec["word"]).tobytes()).hexdigest()
    meta["union_freqs_sha256"] = hashlib.sha256(np.ascontiguousarray(urec["freq"]).tobytes()).hexdigest()
    with open(os.path.join(tmp, "map0.txt"), "wb") as f:
        sh("glistquery L/%s_0_%d.list -l %s" % (names[1], k, cur), tmp, stdout=f)
    with open(os.path.join(tmp, "map0.txt"), "rb") as f:
        meta["mapped_sample"] = names[1]
        meta["mapped_sha256"] = hashlib.sha256(f.read()).hexdigest()
    for run, flags in flag_sets.items():
        rd = os.path.join(out, run)
        os.makedirs(rd)
        for fn in os.listdir(tmp):
            if fn.startswith(("chi2_results_", "log.txt")) or fn.endswith(("_MLdf.csv", ".pkl")):
                os.remove(os.path.join(tmp, fn))
        t0 = time.time()
        with open(os.path.join(tmp, "stderr_%s.txt" % run), "w") as err:
            subprocess.run([sys.executable, os.path.join(HERE, "ref_shim.py"), "modeling", "data.pheno", "-nt", str(nt),
                            "-l", str(k)] + flags, cwd=tmp, env=ENV, stderr=err, stdout=err)
        kept = []
        for fn in sorted(os.listdir(tmp)):
            if fn.startswith("chi2_results_") or fn.endswith("_MLdf.csv"):
                with open(os.path.join(tmp, fn), "rb") as f, gzip.GzipFile(os.path.join(rd, fn + ".gz"), "wb", mtime=0) as g:
                    g.write(f.read())
                kept.append(fn)
        meta["runs"][run] = {"flags": flags, "outputs": kept, "reference_wall_s": round(time.time() - t0, 1)}
    with open(os.path.join(out, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)
    print(tag, "M =", meta["n_union"], meta["runs"])


# ------------------------------------------------------------------------------------------
def gen_chi2_kat():
    import ref_shim
    M = ref_shim.load_modeling()
    rng = np.random.default_rng(4242)
    cases = []

    def run(pheno, pres, weights, mn, mx, cutoff, omit_B, n_kmers):
        M.Samples.no_samples = 0
        samples = [M.Samples("s%02d" % i, "x", {"P": pheno[i]}, weights[i]) for i in range(len(pheno))]
        M.Samples.min_samples, M.Samples.max_samples = mn, mx
        M.phenotypes.pvalue_cutoff = cutoff
        M.phenotypes.omit_B = omit_B
        M.phenotypes.no_kmers_to_analyse = n_kmers
        ph = M.phenotypes("P")
        try:
            r = ph.conduct_chi_squared_test("ACGTACGTACGTA", list(pres), samples)
        except ValueError as e:  # scipy>=1.14 raises where the pinned scipy 1.4.1 returned NaN
            r = "ValueError"
        if r is not None and r != "ValueError":
            r = [r[0], float(r[1]), r[2], int(r[3]), r[4]] + [int(x) for x in r[5:]]
        cases.append({"pheno": list(pheno), "presence": [int(x) for x in pres],
                      "weights": [float(w) if not isinstance(w, int) else w for w in weights], "min": mn, "max": mx,
                      "pvalue_cutoff": cutoff, "omit_B": omit_B, "n_kmers": n_kmers, "result": r})

    # the survey's hand KATs
    run([1, 1, 1, 0, 0, 0, 1, 0], [1, 1, 1, 0, 0, 0, 0, 1], [1] * 8, 2, 6, 0.5, True, 10)
    run([1] * 6 + [0] * 6, [1] * 6 + [0] * 6, [1] * 12, 2, 10, 0.05, True, 10)
    run([1, 1, 1, "NA", 0, 0, 0, "NA", 1, 0], [1, 1, 0, 1, 0, 0, 1, 1, 1, 0], [1] * 10, 2, 8, 0.5, True, 10)
    run([1, 1, 1, 0, 0, 0, 1, 0], [1, 1, 1, 0, 0, 0, 0, 1], [.5, 1.5, 1, 2, .25, 1, .75, 1], 2, 6, 0.5, True, 10)
    run([1] * 6 + [0] * 6, [1] * 6 + [0] * 6, [1] * 12, 2, 10, 0.05, False, 10)
    run([1] * 6 + [0] * 6, [1] * 6 + [0] * 6, [1] * 12, 2, 10, 0.05, False, 100)
    run([0] * 8, [1, 1, 1, 0, 0, 0, 0, 1], [1] * 8, 2, 6, 0.5, True, 10)
    run([1, 1, 1, 0, 0, 0, 1, 0], [1, 0, 0, 0, 0, 0, 0, 0], [1] * 8, 2, 6, 0.5, True, 10)   # n_w = 1
    run([1, 1, 1, 0, 0, 0, 1, 0], [1, 1, 1, 1, 1, 1, 1, 0], [1] * 8, 2, 6, 0.5, True, 10)   # n_wo = 1
    run([1, 1, 1, 0, 0, 0, 1, 0], [1, 1, 1, 1, 1, 1, 0, 0], [1] * 8, 2, 5, 0.5, True, 10)   # n_w > max
    run([1, 1, 1, 0, 0, 0, 1, 0], [2, 5, 1, 0, 0, 0, 0, 3], [1] * 8, 2, 6, 0.5, True, 10)   # real counts
    for _ in range(300):
        n = int(rng.integers(6, 70))
        pheno = [("NA" if rng.random() < 0.1 else int(rng.random() < 0.5)) for _ in range(n)]
        assoc = rng.random()
        pres = [int(rng.random() < (0.15 + 0.7 * assoc * (p == 1))) if p != "NA" else int(rng.random() < 0.5)
                for p in pheno]
        if rng.random() < 0.5:
            weights = [1] * n
        else:
            weights = [float(np.round(rng.uniform(0.05, 3.0), 6)) for _ in range(n)]
        mn = int(rng.integers(1, 4))
        mx = int(n - rng.integers(0, 4))
        cutoff = float(rng.choice([0.05, 0.5, 0.9, 1e-3]))
        omit_B = bool(rng.random() < 0.5)
        n_kmers = int(rng.choice([1, 10, 1000, 100000]))
        run(pheno, pres, weights, mn, mx, cutoff, omit_B, n_kmers)
    with open(os.path.join(GOLD, "chi2_kat.json"), "w") as f:
        json.dump({"source": "PhenotypeSeeker.modeling.phenotypes.conduct_chi_squared_test (modeling.py:759-798)",
                   "cases": cases}, f)
    print("chi2 KATs:", len(cases), "kept:", sum(1 for c in cases if c["result"] not in (None, "ValueError")),
          "ValueError:", sum(1 for c in cases if c["result"] == "ValueError"))


def gen_welch_kat():
    from scipy import stats
    rng = np.random.default_rng(777)
    cases = []
    for it in range(200):
        n = int(rng.integers(6, 80))
        pres = (rng.random(n) < rng.uniform(0.2, 0.8)).astype(int)
        if pres.sum() < 2 or (1 - pres).sum() < 2:
            continue
        vals = np.round(rng.normal(0, 1, n) * rng.uniform(0.1, 5) + pres * rng.uniform(-2, 2) + rng.uniform(-10, 10), 4)
        integer_w = it % 2 == 1
        w = rng.integers(1, 5, n) if integer_w else np.ones(n, dtype=int)
        x = np.repeat(vals[pres == 1], w[pres == 1])
        y = np.repeat(vals[pres == 0], w[pres == 0])
        r = stats.ttest_ind(x, y, equal_var=False)
        cases.append({"values": vals.tolist(), "presence": pres.tolist(), "weights": w.tolist(),
                      "t": float(r.statistic), "p": float(r.pvalue), "df": float(r.df),
                      "mean_x": float(np.average(vals[pres == 1], weights=w[pres == 1])),
                      "mean_y": float(np.average(vals[pres == 0], weights=w[pres == 0]))})
    # Student-t survival function table
    tsf = [{"t": float(t), "df": float(df), "p": float(2 * stats.t.sf(abs(t), df))}
           for t in (0.0, 0.1, 0.5, 1.0, 2.0, 3.5, 6.0, 12.0, 40.0) for df in (1.0, 1.7, 2.0, 5.5, 10.0, 30.0, 250.0, 2046.0)]
    with open(os.path.join(GOLD, "welch_kat.json"), "w") as f:
        json.dump({"source": "scipy.stats.ttest_ind(equal_var=False); integer weights = replicated observations "
                             "(frequency-weight identity of statsmodels DescrStatsW, modeling.py:734)",
                   "cases": cases, "t_sf": tsf}, f)
    print("welch KATs:", len(cases))


def gen_model_kat():
    """Converged optima of the two estimators the reference fits (modeling.py:999-1014,
    :1075-1085, :1208-1216) + the CV splitters GridSearchCV uses for an integer cv."""
    import pandas as pd
    from sklearn.linear_model import Lasso, LogisticRegression
    from sklearn.model_selection import GridSearchCV, KFold, StratifiedKFold
    df = pd.read_csv(os.path.join(GOLD, "ds_omitB", "Pheno_MLdf.csv"), index_col=0)
    X = df.iloc[:, 0:-2].values.astype(np.float64)
    y = df.iloc[:, -1].values.astype(int)
    rng = np.random.default_rng(99)
    # a second, less degenerate design: random sparse binary columns with a planted signal
    n2, p2 = 60, 40
    X2 = (rng.random((n2, p2)) < 0.3).astype(np.float64)
    logit = 2.5 * X2[:, 0] - 2.0 * X2[:, 1] + 1.5 * X2[:, 2] - 0.5
    y2 = (rng.random(n2) < 1 / (1 + np.exp(-logit))).astype(int)
    yc2 = 1.5 * X2[:, 0] - 2.0 * X2[:, 3] + 0.7 * X2[:, 5] + rng.normal(0, 0.3, n2) + 4.0
    Cs = [1 / a for a in np.logspace(-3, 3, 13)]
    out = {"X1": X, "y1": y, "X2": X2, "y2": y2, "yc2": yc2, "Cs": np.array(Cs)}
    for tag, XX, yy in (("1", X, y), ("2", X2, y2)):
        coefs, icpts, objs = [], [], []
        for C in Cs:
            m = LogisticRegression(penalty="l1", solver="liblinear", C=C, tol=1e-10, max_iter=20000,
                                   random_state=0).fit(XX, yy)
            w, b = m.coef_[0], m.intercept_[0]
            z = XX @ w + b
            ypm = 2 * yy - 1
            obj = np.abs(w).sum() + abs(b) + C * np.logaddexp(0, -ypm * z).sum()
            coefs.append(w); icpts.append(b); objs.append(obj)
        out["logreg_coef" + tag] = np.array(coefs)
        out["logreg_icpt" + tag] = np.array(icpts)
        out["logreg_obj" + tag] = np.array(objs)
        cv = int(min(np.bincount(yy).min(), 10))
        folds = np.full(len(yy), -1)
        for f, (_, te) in enumerate(StratifiedKFold(n_splits=cv).split(XX, yy)):
            folds[te] = f
        out["skf_folds" + tag] = folds
        gs = GridSearchCV(LogisticRegression(penalty="l1", solver="liblinear", tol=1e-10, max_iter=20000,
                                             random_state=0), {"C": Cs}, cv=cv).fit(XX, yy)
        out["gs_mean_score" + tag] = gs.cv_results_["mean_test_score"]
        out["gs_std_score" + tag] = gs.cv_results_["std_test_score"]
        out["gs_best_C" + tag] = np.array(gs.best_params_["C"])
    alphas = np.logspace(-3, 3, 13)
    lc, li = [], []
    for a in alphas:
        m = Lasso(alpha=a, tol=1e-13, max_iter=1000000).fit(X2, yc2)
        lc.append(m.coef_); li.append(m.intercept_)
    out["alphas"] = alphas
    out["lasso_coef2"] = np.array(lc)
    out["lasso_icpt2"] = np.array(li)
    kfolds = np.full(n2, -1)
    for f, (_, te) in enumerate(KFold(n_splits=10).split(X2)):
        kfolds[te] = f
    out["kf_folds2"] = kfolds
    gs = GridSearchCV(Lasso(tol=1e-13, max_iter=1000000), {"alpha": alphas}, cv=10).fit(X2, yc2)
    out["lasso_gs_mean_score2"] = gs.cv_results_["mean_test_score"]
    out["lasso_gs_best_alpha2"] = np.array(gs.best_params_["alpha"])
    np.savez_compressed(os.path.join(GOLD, "model_kat.npz"), **out)
    print("model KATs: logreg best C", out["gs_best_C1"], out["gs_best_C2"], "lasso best alpha",
          out["lasso_gs_best_alpha2"])


def gen_model_l2_kat():
    """Converged optima of the `--penalty L2` estimators (set_model, modeling.py:1001-1002, :1015-1019) on
    the designs of model_kat.npz: Ridge (direct solve) and LogisticRegression(penalty='l2') with a free
    intercept (newton-cg stands for lbfgs/newton-cg/sag/saga: one objective) and with liblinear's
    penalised one; plus the GridSearchCV scores."""
    from sklearn.linear_model import LogisticRegression, Ridge
    from sklearn.model_selection import GridSearchCV
    z = np.load(os.path.join(GOLD, "model_kat.npz"))
    X2, y2, yc2, Cs, alphas = z["X2"], z["y2"], z["yc2"], z["Cs"], z["alphas"]
    X1, y1 = z["X1"], z["y1"]
    rng = np.random.default_rng(7)
    yc1 = 3.0 * X1[:, 0] + rng.normal(0, 0.5, X1.shape[0]) + 1.0  # continuous target on the degenerate design
    out = {"yc1": yc1}
    for tag, XX, yy in (("1", X1, yc1), ("2", X2, yc2)):
        rc, ri = [], []
        for a in alphas:
            m = Ridge(alpha=a).fit(XX, yy)
            rc.append(m.coef_); ri.append(m.intercept_)
        out["ridge_coef" + tag], out["ridge_icpt" + tag] = np.array(rc), np.array(ri)
    gs = GridSearchCV(Ridge(), {"alpha": list(alphas)}, cv=10).fit(X2, yc2)
    out["ridge_gs_mean_score2"] = gs.cv_results_["mean_test_score"]
    out["ridge_gs_best_alpha2"] = np.array(gs.best_params_["alpha"])
    for tag, XX, yy in (("1", X1, y1), ("2", X2, y2)):
        ypm = 2.0 * yy - 1.0
        for name, kw in (("free", dict(solver="newton-cg", tol=1e-13, max_iter=100000)),
                         ("liblinear", dict(solver="liblinear", tol=1e-13, max_iter=100000))):
            cc, ii, gg = [], [], []
            for C in Cs:
                m = LogisticRegression(penalty="l2", C=C, **kw).fit(XX, yy)
                w, b = m.coef_[0], m.intercept_[0]
                s = 1.0 / (1.0 + np.exp(ypm * (XX @ w + b)))
                gw = w - C * (XX.T @ (ypm * s))
                gb = (b if name == "liblinear" else 0.0) - C * (ypm * s).sum()
                cc.append(w); ii.append(b); gg.append(max(np.abs(gw).max(), abs(gb)) / max(C, 1.0))
            out["l2_%s_coef%s" % (name, tag)] = np.array(cc)
            out["l2_%s_icpt%s" % (name, tag)] = np.array(ii)
            out["l2_%s_gradmax%s" % (name, tag)] = np.array(gg)
        cv = int(min(np.bincount(yy).min(), 10))
        gs = GridSearchCV(LogisticRegression(penalty="l2", solver="newton-cg", tol=1e-13, max_iter=100000),
                          {"C": list(Cs)}, cv=cv).fit(XX, yy)
        out["l2_gs_mean_score" + tag] = gs.cv_results_["mean_test_score"]
        out["l2_gs_best_C" + tag] = np.array(gs.best_params_["C"])
    np.savez_compressed(os.path.join(GOLD, "model_l2_kat.npz"), **out)
    print("L2 model KATs: worst scaled gradient free %.2e liblinear %.2e; best alpha %g best C %g %g" % (
        max(out["l2_free_gradmax1"].max(), out["l2_free_gradmax2"].max()),
        max(out["l2_liblinear_gradmax1"].max(), out["l2_liblinear_gradmax2"].max()),
        out["ridge_gs_best_alpha2"], out["l2_gs_best_C1"], out["l2_gs_best_C2"]))


def gen_model_mid_kat():
    """Mid-size L1 logistic problems (the covariance-form QP with 2 and 3 feature slots per lane, random sweeps,
    the CG accelerator): tightly converged liblinear objectives on a random sparse design and on a
    near-duplicate one (columns = one pattern with a few flipped samples)."""
    from sklearn.linear_model import LogisticRegression
    rng = np.random.default_rng(2024)
    n = 256
    Xa = (rng.random((n, 100)) < 0.3)
    logit = 2.5 * Xa[:, 0] - 2.0 * Xa[:, 1] + 1.5 * Xa[:, 2] + 1.0 * Xa[:, 3] - 1.0
    ya = (rng.random(n) < 1 / (1 + np.exp(-logit))).astype(int)
    base = rng.random(n) < 0.5
    Xb = np.repeat(base[:, None], 150, axis=1)
    for j in range(150):
        flip = rng.choice(n, size=int(rng.integers(1, 6)), replace=False)
        Xb[flip, j] = ~Xb[flip, j]
    yb = (base ^ (rng.random(n) < 0.1)).astype(int)
    Cs = [100.0, 3.1622776601683795, 0.31622776601683794]
    out = {"Xa": np.packbits(Xa, axis=0), "ya": ya, "Xb": np.packbits(Xb, axis=0), "yb": yb, "n": np.array(n), "Cs": np.array(Cs)}
    for tag, XX, yy in (("a", Xa.astype(np.float64), ya), ("b", Xb.astype(np.float64), yb)):
        objs = []
        ypm = 2.0 * yy - 1.0
        for C in Cs:
            m = LogisticRegression(penalty="l1", solver="liblinear", C=C, tol=1e-9, max_iter=100000).fit(XX, yy)
            w, b = m.coef_[0], m.intercept_[0]
            objs.append(np.abs(w).sum() + abs(b) + C * np.logaddexp(0, -ypm * (XX @ w + b)).sum())
        out["obj_" + tag] = np.array(objs)
    np.savez_compressed(os.path.join(GOLD, "model_mid_kat.npz"), **out)
    print("mid-size model KATs:", out["obj_a"], out["obj_b"])


def large_designs():
    """The three designs of model_large_kat.npz: (tag, X bool [n][p], y int [n]).  'g' is the (2,048 x 907) top-1000 design of a
    2,048-genome run (tests/golden/fit2048_907.npz: the columns our own pipeline selected -- an INPUT; everything stored about
    it below is scikit-learn's output), 'h' and 'i' are near-duplicate designs (columns = one of nb factors with a few
    samples flipped) on either side of the 1,024-sample border: 1,500 x 300 and 300 x 400."""
    d = np.load(os.path.join(GOLD, "fit2048_907.npz"))
    out = [("g", np.unpackbits(d["Xbits"], axis=1)[:, : int(d["p"])].astype(bool), d["y"].astype(int))]
    rng = np.random.default_rng(4242)
    for tag, n, p, nb, flip, noise in (("h", 1500, 300, 20, 0.03, 0.05), ("i", 300, 400, 30, 0.04, 0.08)):
        base = rng.random((n, nb)) < 0.35
        X = base[:, rng.integers(0, nb, p)] ^ (rng.random((n, p)) < flip)
        y = ((base[:, 0] & base[:, 3]) ^ (rng.random(n) < noise)).astype(int)
        assert len({X[:, j].tobytes() for j in range(p)}) == p      # distinct columns: coefficients are comparable one by one
        out.append((tag, X, y))
    return out


def gen_model_large_kat(only=None):
    """reference outputs for the solver forms that carry every large fit (Gram matrix in global memory: more
    than 192 distinct columns at any n, more than 64 from 1,024 samples on; and the four-wave array form behind it).
    modeling.py:1011-1014 (LogisticRegression(penalty='l1', solver='liblinear')), :1078-1085 + :1208-1216 (GridSearchCV over
    C = 1 / logspace(-3, 3, 13), cv = min(min class, 10) from :1512-1524, accuracy, refit).  Per design:
      * liblinear run as tightly as it converges in minutes (tol 1e-8 up to C = 10, 1e-6 at C = 100) at C in {0.01 .. 100} on
        all samples and on two training folds of the StratifiedKFold split: coefficients, intercept, objective, n_iter;
      * the exact optimum next to it: oracle_model.logreg_l1_arbiter seeded with liblinear's point (active-set Newton until
        the KKT residual of liblinear's own objective is < 1e-10 max(1, C) -- a certificate anyone can re-check from the stored
        point with one gradient evaluation, tests/test_oracle_golden.py does);
      * GridSearchCV exactly as the reference configures it (tol = 1e-4, max_iter = 1000; the reference leaves
        random_state = None, i.e. liblinear's coordinate order changes from run to run, so three seeds are recorded: their
        spread is what 'the same scores as scikit-learn' can mean)."""
    import oracle_model as OM
    from sklearn.linear_model import LogisticRegression
    from sklearn.model_selection import GridSearchCV, StratifiedKFold
    Cs_fit = [0.01, 0.1, 1.0, 10.0, 100.0]
    grid = [1 / a for a in np.logspace(-3, 3, 13)]
    path = os.path.join(GOLD, "model_large_kat.npz")
    out = dict(np.load(path)) if (only and os.path.exists(path)) else {}
    out["Cs_fit"], out["grid"] = np.array(Cs_fit), np.array(grid)
    for tag, Xb, y in large_designs():
        if only and tag not in only:
            continue
        n, p = Xb.shape
        X = Xb.astype(np.float64)
        cv = int(min(np.bincount(y).min(), 10))
        fold = np.full(n, -1)
        for f, (_, te) in enumerate(StratifiedKFold(n_splits=cv).split(X, y)):
            fold[te] = f
        held = [-1, 0, cv - 3]
        if tag != "g":
            out["X_" + tag] = np.packbits(Xb, axis=0)
            out["y_" + tag] = y
        out["n_" + tag], out["fold_" + tag], out["held_" + tag] = np.array(n), fold, np.array(held)
        rec = {k: [] for k in ("lib_coef", "lib_icpt", "lib_obj", "lib_tol", "lib_niter", "lib_secs", "arb_coef", "arb_group", "arb_icpt",
                               "arb_obj", "arb_kkt", "fit_C", "fit_held")}
        for hf in held:
            tr = fold != hf
            Xt, yt = X[tr], y[tr]
            ypm = 2.0 * yt - 1.0
            for C in Cs_fit:
                tol = 1e-8 if C <= 10 else 1e-6
                t0 = time.time()
                m = LogisticRegression(penalty="l1", solver="liblinear", C=C, tol=tol, max_iter=1000000, random_state=0).fit(Xt, yt)
                secs = time.time() - t0
                w, b = m.coef_[0].copy(), float(m.intercept_[0])
                obj = np.abs(w).sum() + abs(b) + C * np.logaddexp(0, -ypm * (Xt @ w + b)).sum()
                a = OM.logreg_l1_arbiter(Xt, yt, C, w, b, max_rounds=400)
                # (the residual's floor is the rounding of the gradient's sums, ~ C n 2^-53 per coordinate: 1e-9 at C = 100)
                assert a["kkt"] < 1e-10 * max(1.0, C) and not a["rank_deficient"], (tag, hf, C, a["kkt"], a["rank_deficient"])
                # on a training fold two columns may differ in held-out rows only: the arbiter collapses them and the SUM of
                # their coefficients is what is unique -- stored on the first column of each group, the groups beside it
                grp = a["group"]
                first = np.full(len(a["w_groups"]), -1)
                for j in range(p - 1, -1, -1):
                    first[grp[j]] = j
                aw = np.zeros(p)
                aw[first] = a["w_groups"]
                assert a["objective"] <= obj * (1 + 1e-14), (tag, hf, C, a["objective"], obj)
                for k, v in (("lib_coef", w), ("lib_icpt", b), ("lib_obj", obj), ("lib_tol", tol), ("lib_niter", int(m.n_iter_[0])),
                             ("lib_secs", secs), ("arb_coef", aw), ("arb_group", grp), ("arb_icpt", a["b"]), ("arb_obj", a["objective"]),
                             ("arb_kkt", a["kkt"]), ("fit_C", C), ("fit_held", hf)):
                    rec[k].append(v)
                print("large KAT %s held %2d C %-6g liblinear %.1fs n_iter %d obj %.12g; arbiter obj %.12g (rel %.1e) kkt %.1e nnz %d"
                      % (tag, hf, C, secs, m.n_iter_[0], obj, a["objective"], obj / a["objective"] - 1, a["kkt"],
                         int((a["w_groups"] != 0).sum())), flush=True)
        for k, v in rec.items():
            out[k + "_" + tag] = np.array(v)
        sc, best = [], []
        for seed in (0, 1, 2):
            t0 = time.time()
            gs = GridSearchCV(LogisticRegression(penalty="l1", solver="liblinear", tol=1e-4, max_iter=1000, random_state=seed),
                              {"C": grid}, cv=cv, n_jobs=6).fit(X, y)
            sc.append(np.array([gs.cv_results_["split%d_test_score" % f] for f in range(cv)]).T)   # [candidate][fold]
            best.append(gs.best_params_["C"])
            print("large KAT %s GridSearchCV seed %d: %.0fs best C %g mean scores %s" % (
                tag, seed, time.time() - t0, best[-1], np.round(gs.cv_results_["mean_test_score"], 4).tolist()), flush=True)
        out["gs_split_scores_" + tag] = np.array(sc)          # [seed][candidate][fold]
        out["gs_best_C_" + tag] = np.array(best)
        np.savez_compressed(path, **out)
    np.savez_compressed(path, **out)


def gen_lasso_large_kat():
    """scikit-learn's Lasso exactly as the reference configures it (modeling.py:999-1000: Lasso(max_iter=1000,
    tol=1e-4); :1041 alpha = logspace(-3, 3, 13); GridSearchCV with cv = 10, R^2) on a 1,024 x 907 design -- the first 1,024
    genomes of fit2048_907.npz with a continuous phenotype carried by two k-mers --: per alpha, on all samples and on two
    training folds of KFold(10): coef_, intercept_, n_iter_ (several fits end at the 1,000-sweep limit: the fixture pins the
    point scikit-learn's cyclic descent has reached THEN, not an optimum) and dual_gap_; plus the grid search's scores."""
    import warnings
    from sklearn.linear_model import Lasso
    from sklearn.model_selection import GridSearchCV, KFold
    d = np.load(os.path.join(GOLD, "fit2048_907.npz"))
    X = np.unpackbits(d["Xbits"], axis=1)[:1024, : int(d["p"])].astype(np.float64)
    rng = np.random.default_rng(5)
    y = 2.0 * X[:, 3] - 1.5 * X[:, 40] + rng.normal(0, 0.5, 1024)
    alphas = np.logspace(-3, 3, 13)
    fold = np.full(1024, -1)
    for f, (_, te) in enumerate(KFold(n_splits=10).split(X)):
        fold[te] = f
    out = {"y": y, "alphas": alphas, "fold": fold, "held": np.array([-1, 0, 6])}
    rec = {k: [] for k in ("coef", "icpt", "n_iter", "gap", "fit_alpha", "fit_held")}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for hf in out["held"]:
            tr = fold != hf
            for a in alphas:
                m = Lasso(alpha=a, tol=1e-4, max_iter=1000).fit(X[tr], y[tr])
                for k, v in (("coef", m.coef_), ("icpt", m.intercept_), ("n_iter", m.n_iter_), ("gap", m.dual_gap_), ("fit_alpha", a),
                             ("fit_held", hf)):
                    rec[k].append(v)
                print("lasso KAT held %2d alpha %-8g n_iter %4d gap %.3e nnz %d" % (hf, a, m.n_iter_, m.dual_gap_, (m.coef_ != 0).sum()), flush=True)
        gs = GridSearchCV(Lasso(tol=1e-4, max_iter=1000), {"alpha": alphas}, cv=10).fit(X, y)
    for k, v in rec.items():
        out[k] = np.array(v)
    out["gs_split_scores"] = np.array([gs.cv_results_["split%d_test_score" % f] for f in range(10)]).T
    out["gs_best_alpha"] = np.array(gs.best_params_["alpha"])
    np.savez_compressed(os.path.join(GOLD, "lasso_large_kat.npz"), **out)
    print("lasso large KAT: best alpha", out["gs_best_alpha"], "mean scores", np.round(gs.cv_results_["mean_test_score"], 5).tolist())


def gen_gmer_counter():
    """prediction.py:72-80,145-148: db line 'KMER\\t1\\tKMER', output parsed at :82-100."""
    gs = GenomeSet(4, 6000, seed=31, gene_len=200)
    k = 13
    cases = []
    rng = np.random.default_rng(5)
    with tempfile.TemporaryDirectory() as tmp:
        name, fa = gs.sample(0)
        codes = gs.codes(0)
        # dictionary: k-mers taken from the genome (either strand as written), some absent ones
        kmers = []
        for s in rng.integers(0, len(codes) - k, 25):
            kmers.append("".join("ACGT"[c] for c in codes[s:s + k]))
        kmers += ["ACGTACGTACGTA", "TTTTTTTTTTTTT", "GGGGGGGGGGGGC"]
        kmers = list(dict.fromkeys(kmers))
        with open(os.path.join(tmp, "db.txt"), "w") as f:
            for km in kmers:
                f.write("%s\t1\t%s\n" % (km, km))
        for i in range(3):
            name, fa = gs.sample(i)
            with open(os.path.join(tmp, name + ".fasta"), "wb") as f:
                f.write(fa)
            with open(os.path.join(tmp, "o.txt"), "wb") as f:
                sh("gmer_counter -db db.txt %s.fasta" % name, tmp, stdout=f)
            with open(os.path.join(tmp, "o.txt")) as f:
                txt = f.read()
            cases.append({"fasta_gz_b64": b64(gzip.compress(fa, mtime=0)), "output": txt})
    with open(os.path.join(GOLD, "gmer_counter.json"), "w") as f:
        json.dump({"source": "gmer_counter 4.2.7 (reference bin/)", "k": k, "kmers": kmers, "cases": cases}, f)
    print("gmer_counter cases:", len(cases))


def gen_mash():
    """modeling.py:386-412: `mash sketch -r` per sample, `mash paste`, `mash dist` (bin/mash 2.2)."""
    env_run = lambda cmd, cwd: subprocess.run(cmd, shell=True, cwd=cwd, env=ENV, capture_output=True, text=True)
    gs = GenomeSet(6, 12000, seed=77, gene_len=400)
    out = {"source": "mash 2.2 (reference bin/): `mash sketch -r`, `mash info -d`, `mash paste`, `mash dist`",
           "k": 21, "sketch_size": 1000, "seed": 42, "samples": []}
    with tempfile.TemporaryDirectory() as tmp:
        names = []
        for i in range(gs.n):
            name, fa = gs.sample(i)
            if i == 2:   # lower case + N runs
                fa = fa.replace(b"ACGT", b"acgt", 50).replace(b"GATT", b"GNNT", 3)
            if i == 5:   # fewer than sketch_size distinct k-mers
                fa = b">short\n" + fa.split(b"\n", 1)[1][:700] + b"\n"
            with open(os.path.join(tmp, name + ".fasta"), "wb") as f:
                f.write(fa)
            env_run("mash sketch -r %s.fasta -o %s" % (name, name), tmp)
            info = json.loads(env_run("mash info -d %s.msh" % name, tmp).stdout)
            out["samples"].append({"name": name, "fasta_gz_b64": b64(gzip.compress(fa, mtime=0)),
                                   "hashes": info["sketches"][0]["hashes"]})
            names.append(name)
        env_run("mash paste reference " + " ".join(n + ".msh" for n in names), tmp)
        out["dist_table"] = env_run("mash dist reference.msh reference.msh", tmp).stdout
        env_run("mash sketch -k 17 -s 50 %s.fasta -o small" % names[0], tmp)
        out["k17_s50_hashes"] = json.loads(env_run("mash info -d small.msh", tmp).stdout)["sketches"][0]["hashes"]
        env_run("mash sketch -k 15 -s 40 %s.fasta -o k15" % names[0], tmp)
        info = json.loads(env_run("mash info -d k15.msh", tmp).stdout)
        out["k15_s40_hashes"], out["k15_bits"] = info["sketches"][0]["hashes"], info["hashBits"]
    with open(os.path.join(GOLD, "mash.json"), "w") as f:
        json.dump(out, f)
    print("mash sketches:", [len(s["hashes"]) for s in out["samples"]])


def gen_gsc_kat():
    """the parts of the -w path that ARE reference code, run as the reference runs them (through
    ref_shim.load_modeling(), nothing altered):
      plumbing  Samples.get_mash_sketches (:386-390), get_mash_distances (:402-412; bin/mash 2.2 from /root/reference/bin),
                _mash_output_to_distance_matrix (:415-428) and _distance_matrix_modifier (:431-444) on small genome sets:
                one whose data.pheno is name-sorted, one that is not (the rows of distances.mat are then labelled with
                the wrong samples: mash paste takes K-mer_lists/*.msh in glob order, the labels come in pheno order) and
                whose names sort differently with and without the '.msh' suffix;
      gsc       Samples.GSC_weights_from_newick(path, normalize="mean1") (:461-476 -> clip_branch_lengths /
                set_branch_sum / set_node_weight :478-503) on newick texts: hand-written ones (two leaves, a three-child
                root, zero / negative / huge branch lengths, a caterpillar) and neighbour-joining trees of 3 ... 200
                leaves.  ete3 is absent: the Tree the reference instantiates at :465 is oracle_weights.Tree -- OUR
                newick parser and node class; every number after the parse is computed by the reference's functions;
      chains    plumbing -> (neighbour joining + "%1.5f" newick: oracle_weights.nj_newick, OURS, Biopython is absent --
                parity unpinned for that link) -> gsc, per genome set: the weights a `-w` run must end up with.
    """
    import ref_shim
    import oracle_weights as OW
    M = ref_shim.load_modeling()
    M.Tree = OW.Tree
    env_run = lambda cmd, cwd: subprocess.run(cmd, shell=True, cwd=cwd, env=ENV, capture_output=True, text=True)
    os.environ["PATH"] = ENV["PATH"]           # the reference shells out to `mash` by name
    out = {"source": "PhenotypeSeeker.modeling.Samples.{get_mash_sketches, get_mash_distances, _mash_output_to_distance_matrix, "
                     "_distance_matrix_modifier, GSC_weights_from_newick} (modeling.py:386-503) through oracle/ref_shim.py; mash 2.2 "
                     "(reference bin/).  NOT the reference's: the newick parser / Tree class (oracle_weights.Tree in ete3.Tree's "
                     "place) and, in 'chains', the neighbour joining + newick text (oracle_weights.nj_newick in Biopython's place).",
           "plumbing": [], "gsc": [], "chains": {}}

    def plumbing(tag, names, fastas):
        tmp = tempfile.mkdtemp(prefix="psk_gsc_")
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            os.makedirs("K-mer_lists")
            M.Input.samples = M.OrderedDict()
            M.Samples.no_samples = 0
            for nm, fa in zip(names, fastas):
                addr = nm + (".fastq" if fa[:1] == b"@" else ".fasta")
                with open(addr, "wb") as f:
                    f.write(fa)
                M.Input.samples[nm] = M.Samples(nm, addr, {"P": 1}, 1)
            for smp in M.Input.samples.values():
                smp.get_mash_sketches()
            hashes = {nm: json.loads(env_run("mash info -d K-mer_lists/%s.msh" % nm, tmp).stdout)["sketches"][0]["hashes"]
                      for nm in names}
            M.Samples.get_mash_distances()
            M.Samples._mash_output_to_distance_matrix(list(M.Input.samples.keys()), "mash_distances.mat")
            lower = M.Samples._distance_matrix_modifier("distances.mat")
            rec = {"tag": tag, "names": list(names), "hashes": hashes,
                   "fasta_gz_b64": {nm: b64(gzip.compress(fa, mtime=0)) for nm, fa in zip(names, fastas)},
                   "mash_distances_mat": open("mash_distances.mat").read(), "distances_mat": open("distances.mat").read(),
                   "lower_triangle": lower}
        finally:
            os.chdir(cwd)
            shutil.rmtree(tmp, ignore_errors=True)
        return rec

    def gsc(newick, note):
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "tree_newick.txt")
            with open(path, "w") as f:
                f.write(newick)
            w = M.Samples.GSC_weights_from_newick(path, normalize="mean1")
        return {"note": note, "newick": newick, "weights": {k: float(v) for k, v in w.items()}}

    def chain(rec):
        names = rec["names"]
        full = [[0.0] * len(names) for _ in names]
        for i, row in enumerate(rec["lower_triangle"]):
            for j, v in enumerate(row):
                full[i][j] = full[j][i] = v
        nw = OW.nj_newick(names, full)
        g = gsc(nw, "chain " + rec["tag"])
        return {"names": names, "newick": nw, "weights": g["weights"]}

    gs = GenomeSet(6, 9000, seed=301, gene_len=300)
    out["plumbing"].append(plumbing("sorted6", [gs.name(i) for i in range(6)], [gs.sample(i)[1] for i in range(6)]))
    gs = GenomeSet(9, 8000, seed=302, gene_len=300, sub_rate=0.01)
    odd = ["zeta", "alpha", "S1", "S1-2", "S1+x", "mid", "Beta", "beta", "a_b"]
    fas = []
    for i, nm in enumerate(odd):
        fa = gs.sample(i)[1]
        if nm == "mid":       # a distant sample: no shared hashes with some of the others
            fa = GenomeSet(1, 8000, seed=999).sample(0)[1]
        fas.append(fa)
    out["plumbing"].append(plumbing("shuffled9", odd, fas))
    # the e2e set of the GPU tests (20 samples, one of them FASTQ reads), as the reference's `-w` sees it
    from helpers_golden import load_dataset_files
    names, files = load_dataset_files(os.path.join(GOLD, "ds_omitB"))
    rec = plumbing("ds_omitB", names, [files[nm] for nm in names])
    rec.pop("fasta_gz_b64")                    # the inputs are tests/golden/ds_omitB/*.gz
    out["plumbing"].append(rec)
    for rec in out["plumbing"]:
        out["chains"][rec["tag"]] = chain(rec)

    hand = [
        ("(A:1,B:2);", "two leaves"),
        ("(A:0.5,B:0.5)Inner:0.00000;", "two leaves as Biopython roots them"),
        ("(A:1,B:2,C:3)Inner1:0.00000;", "three-child root"),
        ("((A:1,B:2)Inner1:1.5,(C:1,D:3)Inner2:0.5);", "two cherries"),
        ("((A:0.00000,B:0.00000)Inner1:0.00000,C:0.01234,D:0.00000)Inner2:0.00000;", "zero lengths: all clipped to 1e-9"),
        ("((A:-0.00012,B:0.00410)Inner1:0.00100,C:-0.00001,D:0.02000)Inner2:0.00000;", "negative lengths (NJ produces them): clipped"),
        ("((A:2e9,B:1)Inner1:3e10,C:5)Inner2:0;", "lengths above 1e9: clipped"),
        ("((((A:1,B:1)I1:1,C:1)I2:1,D:1)I3:1,E:1)I4:0;", "caterpillar"),
        ("((A:0.1,B:0.2,C:0.3,D:0.4)I1:0.5,E:0.6)I2:0;", "four-child inner node"),
        ("(A:1e-9,B:1e-9,C:1e-9)R:0;", "lengths at the lower clip"),
        ("((A:0.00001,B:0.00001)Inner1:0.00001,(C:0.12345,D:0.54321)Inner2:0.33333,E:0.99999)Inner3:0.00000;", "five digits"),
        ("((A,B)I1,C)I2;", "no lengths at all: the parser's defaults (1.0; root 0.0)"),
    ]
    for nw, note in hand:
        out["gsc"].append(gsc(nw, note))
    rng = np.random.default_rng(606)
    for n in (3, 4, 5, 6, 7, 8, 10, 13, 17, 24, 33, 50, 77, 100, 150, 200):
        for kind in ("mash-like", "ties", "clonal"):
            if kind == "mash-like":
                half = np.array([[float("%g" % v) for v in row] for row in rng.random((n, n)) * 0.1])
            elif kind == "ties":
                half = rng.choice([0.0, 0.001, 0.002, 0.0153, 1.0], (n, n))
            else:                 # a few clones + noise of the size of mash's 6th digit: near-zero and negative branches
                grp = rng.integers(0, max(2, n // 4), n)
                half = np.where(grp[:, None] == grp[None, :], 0.0, 0.02) + np.round(rng.random((n, n)) * 1e-4, 6)
            m = np.tril(half, -1)
            m = m + m.T
            nw = OW.nj_newick(["s%d" % i for i in range(n)], m)
            out["gsc"].append(gsc(nw, "neighbour joining (oracle_weights.nj_newick) of a %s %d x %d matrix" % (kind, n, n)))
    with open(os.path.join(GOLD, "gsc_kat.json"), "w") as f:
        json.dump(out, f)
    print("gsc KATs: plumbing", [r["tag"] for r in out["plumbing"]], "gsc cases", len(out["gsc"]), "chains", list(out["chains"]))


def gen_model_files():
    """the a11 artefacts as the REFERENCE writes them (modeling.py:975-988 model package, :1219-1412 summary,
    :1414-1455 coefficient table), kept under tests/golden/<set>/model/:
      ds_omitB, ds_bonf   the whole unmodified pipeline again (its chi2 tables and _MLdf.csv must equal the committed ones),
                          keeping summary_of_log_reg_analysis_Pheno.txt, k-mers_and_coefficients_in_log_reg_model_Pheno.txt
                          and log_reg_model_Pheno.pkl (liblinear is unseeded there: grid scores / C / coefficients are ONE
                          draw; the .pkl is that draw's fitted GridSearchCV, so the text files can be re-derived from it);
      ds_cont             `modeling data.pheno -jt modelling` on a committed MIC_MLdf.csv (ds_omitB's reference-written
                          k-mer columns, unit weights, a continuous phenotype 2^(gene) x lognormal noise): the regressor
                          branch -- Lasso grid, MSE / R^2 / Spearman / Pearson / within-one-dilution lines -- which needs no
                          statsmodels; plus the hold-out and outer-CV layouts (-ts 0.25, -cv1 3) of the same set."""
    keep = lambda fn: fn.startswith(("summary_of_", "k-mers_and_coefficients_")) or fn.endswith(".pkl")

    def run_ref(tmp, argv):
        with open(os.path.join(tmp, "stderr.txt"), "a") as err:
            r = subprocess.run([sys.executable, os.path.join(HERE, "ref_shim.py"), "modeling", "data.pheno"] + argv,
                               cwd=tmp, env=ENV, stderr=err, stdout=err)
        assert r.returncode == 0, open(os.path.join(tmp, "stderr.txt")).read()[-3000:]

    from helpers_golden import load_dataset_files
    for tag in ("ds_omitB", "ds_bonf"):
        src = os.path.join(GOLD, tag)
        with open(os.path.join(src, "meta.json")) as f:
            meta = json.load(f)
        tmp = tempfile.mkdtemp(prefix="psk_model_")
        names, files = load_dataset_files(src)
        shutil.copy(os.path.join(src, "data.pheno"), tmp)
        for line in open(os.path.join(src, "data.pheno")).read().splitlines()[1:]:
            nm, fn = line.split()[:2]
            with open(os.path.join(tmp, fn), "wb") as f:
                f.write(files[nm])
        run_ref(tmp, ["-nt", str(meta["nt"]), "-l", str(meta["k"])] + meta["flags"])
        for fn in meta["reference_outputs"]:
            assert open(os.path.join(tmp, fn), "rb").read() == open(os.path.join(src, fn), "rb").read(), (tag, fn)
        out = os.path.join(src, "model")
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
        kept = sorted(fn for fn in os.listdir(tmp) if keep(fn))
        for fn in kept:
            shutil.copy(os.path.join(tmp, fn), os.path.join(out, fn))
        print(tag, "model files:", kept)
        shutil.rmtree(tmp, ignore_errors=True)

    # the continuous set
    import pandas as pd
    out = os.path.join(GOLD, "ds_cont")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    gs = GenomeSet(20, 10000, seed=11, gene_len=300)          # ds_omitB's genomes (same NA rows: 3 and 14)
    rows = ["ID\tAddresses\tMIC"]
    for i in range(gs.n):
        rows.append("%s\t%s.fasta\t%s" % (gs.name(i), gs.name(i), "NA" if i in (3, 14) else repr(round(gs.continuous_phenotype(i), 4))))
    df = pd.read_csv(os.path.join(GOLD, "ds_omitB", "Pheno_MLdf.csv"), index_col=0)
    df["weights"] = 1
    df["phenotype"] = [round(gs.continuous_phenotype(int(nm[1:])), 4) for nm in df.index]
    for sub, argv in (("whole", []), ("holdout", ["-ts", "0.25"]), ("outer_cv", ["-cv1", "3"])):
        tmp = tempfile.mkdtemp(prefix="psk_model_")
        with open(os.path.join(tmp, "data.pheno"), "w") as f:
            f.write("\n".join(rows) + "\n")
        df.to_csv(os.path.join(tmp, "MIC_MLdf.csv"))
        run_ref(tmp, ["-jt", "modelling"] + argv)
        os.makedirs(os.path.join(out, sub))
        kept = sorted(fn for fn in os.listdir(tmp) if keep(fn))
        for fn in kept:
            shutil.copy(os.path.join(tmp, fn), os.path.join(out, sub, fn))
        if sub == "whole":
            shutil.copy(os.path.join(tmp, "data.pheno"), out)
            shutil.copy(os.path.join(tmp, "MIC_MLdf.csv"), out)
        print("ds_cont", sub, kept)
        shutil.rmtree(tmp, ignore_errors=True)


def gen_prediction_of_product_pkl():
    """VERDICT r04 #2, the reverse direction of the .pkl contract: model files THIS package wrote on the GPU box
    (tools/capture_product_run.py -> gpurun_out/product_run/<set>/) are handed to the REFERENCE's prediction.py
    (prediction.py:185-204, with bin/gmer_counter doing the counting) on the same samples; the .pkl and the reference's
    predictions_<pheno>.txt are committed under tests/golden/product_pkl/<set>/.  The -m gpu suite then runs this
    package's `prediction` on the committed .pkl and compares byte for byte."""
    from helpers_golden import load_dataset_files
    src_root = os.path.join(ROOT, "gpurun_out", "product_run")
    for tag, pheno, short in (("ds_omitB", "Pheno", "log_reg"), ("ds_bonf", "Pheno", "log_reg"), ("ds_cont", "MIC", "linreg")):
        src = os.path.join(src_root, tag)
        pkl = "%s_model_%s.pkl" % (short, pheno)
        names, files = load_dataset_files(os.path.join(GOLD, "ds_omitB" if tag == "ds_cont" else tag))
        tmp = tempfile.mkdtemp(prefix="psk_pred_")
        for fn in ("samples.txt", "phenos.txt", pkl):
            shutil.copy(os.path.join(src, fn), tmp)
        for line in open(os.path.join(tmp, "samples.txt")).read().splitlines():
            nm, fn = line.split()[:2]
            with open(os.path.join(tmp, fn), "wb") as f:
                f.write(files[nm])
        with open(os.path.join(tmp, "stderr.txt"), "w") as err:
            r = subprocess.run([sys.executable, os.path.join(HERE, "ref_shim.py"), "prediction", "samples.txt", "phenos.txt"],
                               cwd=tmp, env=ENV, stderr=err, stdout=err)
        assert r.returncode == 0, open(os.path.join(tmp, "stderr.txt")).read()[-3000:]
        out = os.path.join(GOLD, "product_pkl", tag)
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
        for fn in ("samples.txt", "phenos.txt", pkl, "predictions_%s.txt" % pheno):
            shutil.copy(os.path.join(tmp, fn), os.path.join(out, fn))
        same = open(os.path.join(tmp, "predictions_%s.txt" % pheno)).read() == open(os.path.join(src, "predictions_%s.txt" % pheno)).read()
        print(tag, "reference prediction.py on the product's .pkl: written; equal to the product's own predictions:", same)
        shutil.rmtree(tmp, ignore_errors=True)


def gen_split():
    """modeling.py:924-934: train_test_split(ML_df, test_size, random_state=55, stratify=...)."""
    from sklearn.model_selection import train_test_split
    rng = np.random.default_rng(8)
    cases = []
    for _ in range(40):
        n = int(rng.integers(8, 120))
        ts = float(rng.choice([0.1, 0.2, 0.25, 0.3, 0.5]))
        y = (rng.random(n) < rng.uniform(0.25, 0.75)).astype(int)
        if min(np.bincount(y, minlength=2)) < 2:
            continue
        for strat in (True, False):
            try:
                tr, te = train_test_split(np.arange(n), test_size=ts, random_state=55, stratify=y if strat else None)
            except ValueError:
                continue
            cases.append({"n": n, "test_size": ts, "y": y.tolist() if strat else None, "train": tr.tolist(),
                          "test": te.tolist()})
    with open(os.path.join(GOLD, "split_kat.json"), "w") as f:
        json.dump({"source": "sklearn.model_selection.train_test_split(random_state=55)", "cases": cases}, f)
    print("split KATs:", len(cases))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    what = sys.argv[1:] or ["tok", "ds", "chi2", "welch", "model", "model_l2", "model_mid", "gmer", "mash", "split", "gsc"]
    if "tok" in what:
        gen_tokenizer_cases()
    if "ds" in what:
        run_reference_dataset("ds_omitB", GenomeSet(20, 10000, seed=11, gene_len=300), na={3, 14},
                              flags=["--omit_B_correction", "--n_kmers", "100"], fastq_for={5})
        run_reference_dataset("ds_bonf", GenomeSet(44, 6000, seed=23, gene_len=150), na={9}, flags=[])
    if "k21" in what:      # r05: the 64-bit word routes (k >= 17) against the reference itself, whole pipeline: `-l 21`
        run_reference_dataset("ds_k21", GenomeSet(44, 6000, seed=23, gene_len=150), na={9}, flags=[], k=21)
    if "atrich" in what:
        run_reference_dataset_regenerated("ds_atrich", dict(n_samples=60, length=1_000_000, seed=29, gene_len=2000, gc=0.29, contigs=6),
                                          na={4}, flag_sets={"bonf": [], "omitB": ["--omit_B_correction"]})
    if "chi2" in what:
        gen_chi2_kat()
    if "welch" in what:
        gen_welch_kat()
    if "model" in what:
        gen_model_kat()
    if "model_l2" in what:
        gen_model_l2_kat()
    if "model_mid" in what:
        gen_model_mid_kat()
    if "lasso_large" in what:
        gen_lasso_large_kat()
    if "model_large" in what:      # ~25 min (liblinear at C = 10 / 100 on the 2,048 x 907 design: 2 min per fit); not in the default list
        gen_model_large_kat()
    for w_ in what:
        if w_.startswith("model_large:"):
            gen_model_large_kat(only=w_.split(":")[1].split(","))
    if "gmer" in what:
        gen_gmer_counter()
    if "mash" in what:
        gen_mash()
    if "split" in what:
        gen_split()
    if "gsc" in what:
        gen_gsc_kat()
    if "model_files" in what:
        gen_model_files()
    if "product_pkl" in what:      # needs gpurun_out/product_run/ (tools/capture_product_run.py on the GPU box)
        gen_prediction_of_product_pkl()
