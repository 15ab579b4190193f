#!/usr/bin/env python3
"""BASELINE config 4 at full size: 1,024 synthetic 5-Mbp genomes, continuous phenotype, Mash-weighted (-w): wall-clock of
`phenotypeseeker modeling` from FASTA files on disk to the model, with the stage times of log.txt.
usage (GPU box): tools/cfg4_run.py [n_genomes]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from phenotypeseeker_amd.synth import GenomeSet  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
gs = GenomeSet(n, 5_000_000, seed=4242)
rng = np.random.default_rng(7)
tmp = tempfile.mkdtemp(prefix="psk_cfg4_")
rows = ["ID\tAddresses\tMIC"]
t0 = time.time()
for i in range(n):
    name, fa = gs.sample(i)
    with open(os.path.join(tmp, name + ".fasta"), "wb") as f:
        f.write(fa)
    rows.append("%s\t%s.fasta\t%.4f" % (name, name, 2.0 * gs.phenotype(i) + rng.normal(0, 0.5)))
with open(os.path.join(tmp, "data.pheno"), "w") as f:
    f.write("\n".join(rows) + "\n")
print("dataset written in %.1f s" % (time.time() - t0), flush=True)
env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
t0 = time.time()
cmd = [sys.executable, os.path.join(ROOT, "scripts", "phenotypeseeker"), "modeling", "data.pheno", "-w"]
if os.environ.get("PSK_PROFILE"):   # where the host time goes: the CLI under cProfile, the 30 heaviest calls by cumulative time
    cmd = [sys.executable, "-m", "cProfile", "-o", os.path.join(tmp, "prof.out")] + cmd[1:]
r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
print("modeling -w: %.2f s, rc %d" % (time.time() - t0, r.returncode))
if r.returncode:
    print(r.stderr.decode(errors="replace")[-1500:])
print(open(os.path.join(tmp, "log.txt")).read())
if os.environ.get("PSK_PROFILE"):
    import pstats
    pstats.Stats(os.path.join(tmp, "prof.out")).sort_stats("cumulative").print_stats(45)
print(sorted(f for f in os.listdir(tmp) if not f.endswith(".fasta")))
subprocess.run(["rm", "-rf", tmp])
