#!/usr/bin/env python3
"""cProfile of one `phenotypeseeker modeling` run on a synthetic dataset (second run of the process, so
imports and first-launch costs are out).  usage: tools/e2e_profile.py N LENGTH [--continuous] [extra CLI flags]"""
import cProfile
import io
import os
import pstats
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.cli import build_parser  # noqa: E402
from phenotypeseeker_amd.synth import GenomeSet  # noqa: E402

n, length = int(sys.argv[1]), int(sys.argv[2])
extra = sys.argv[3:]
continuous = "--continuous" in extra
extra = [e for e in extra if e != "--continuous"]
tmp = tempfile.mkdtemp(prefix="psk_prof_")
gs = GenomeSet(n, length, seed=12345)
rows = ["ID\tAddresses\tPheno"]
for i in range(n):
    name, fa = gs.sample(i)
    with open(os.path.join(tmp, name + ".fasta"), "wb") as f:
        f.write(fa)
    rows.append("%s\t%s.fasta\t%s" % (name, name, repr(round(gs.continuous_phenotype(i), 4)) if continuous else str(gs.phenotype(i))))
with open(os.path.join(tmp, "data.pheno"), "w") as f:
    f.write("\n".join(rows) + "\n")
os.chdir(tmp)
err = sys.stderr
sys.stderr = open(os.devnull, "w")
args = build_parser().parse_args(["modeling", "data.pheno"] + extra)
args.func(args)
pr = cProfile.Profile()
args = build_parser().parse_args(["modeling", "data.pheno"] + extra)
pr.enable()
args.func(args)
pr.disable()
sys.stderr = err
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(45)
print(out.getvalue())
shutil.rmtree(tmp, ignore_errors=True)
