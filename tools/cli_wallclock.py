#!/usr/bin/env python3
"""Wall-clock of `phenotypeseeker modeling` as a fresh PROCESS (start to exit, what a user's shell sees) on a synthetic data set
written to disk once: the BASELINE's second figure.  Each run is its own subprocess; the phase table of log.txt follows.
usage (GPU box): tools/cli_wallclock.py N_GENOMES [--continuous] [--runs R] [--pause S] [--env NAME=VAL ...] [-- extra CLI flags]
--pause S: sleep S seconds between the runs (r06: a process that starts right after one that held ~170 GB of device memory has
exited pays for the driver's clearing of that memory -- config 3's runs 1, 2 took 6.6 s where run 0 took 1.1 s)"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from phenotypeseeker_amd.synth import GenomeSet  # noqa: E402

argv = sys.argv[1:]
extra = []
if "--" in argv:
    extra = argv[argv.index("--") + 1:]
    argv = argv[:argv.index("--")]
n = int(argv[0])
continuous = "--continuous" in argv
runs = int(argv[argv.index("--runs") + 1]) if "--runs" in argv else 2
pause = float(argv[argv.index("--pause") + 1]) if "--pause" in argv else 0.0
envs = [a for i, a in enumerate(argv) if i > 0 and argv[i - 1] == "--env"]
gs = GenomeSet(n, 5_000_000, seed=4242)
rng = np.random.default_rng(7)
tmp = tempfile.mkdtemp(prefix="psk_cli_")
rows = ["ID\tAddresses\tPheno"]
t0 = time.time()
for i in range(n):
    name, fa = gs.sample(i)
    with open(os.path.join(tmp, name + ".fasta"), "wb") as f:
        f.write(fa)
    rows.append("%s\t%s.fasta\t%s" % (name, name, "%.4f" % (2.0 * gs.phenotype(i) + rng.normal(0, 0.5)) if continuous else str(gs.phenotype(i))))
with open(os.path.join(tmp, "data.pheno"), "w") as f:
    f.write("\n".join(rows) + "\n")
print("dataset of %d genomes written in %.1f s" % (n, time.time() - t0), flush=True)
env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
for e in envs:
    k, v = e.split("=", 1)
    env[k] = v
cmd = [sys.executable, os.path.join(ROOT, "scripts", "phenotypeseeker"), "modeling", "data.pheno"] + extra
for r in range(runs):
    if os.path.exists(os.path.join(tmp, "log.txt")):
        print("".join(l for l in open(os.path.join(tmp, "log.txt")) if l.startswith("Phases of rank 0")), end="", flush=True)
        os.remove(os.path.join(tmp, "log.txt"))
    if r and pause:
        time.sleep(pause)
    t0 = time.time()
    p = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    print("run %d: modeling %s%s: %.3f s, rc %d" % (r, " ".join(extra), (" [" + " ".join(envs) + "]") if envs else "", time.time() - t0, p.returncode), flush=True)
    if p.returncode:
        print(p.stderr.decode(errors="replace")[-1500:])
    elif any(e.startswith("PSK_TRACE=") for e in envs):
        print("\n".join(l for l in p.stderr.decode(errors="replace").splitlines() if l.startswith(("[psk]", "count batch", "psk_"))))
print(open(os.path.join(tmp, "log.txt")).read())
subprocess.run(["rm", "-rf", tmp])
