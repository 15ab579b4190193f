#!/usr/bin/env python3
"""End-to-end `phenotypeseeker modeling` wall-clock (FASTA files on disk -> .pkl written), the second
half of the north-star metric.  Writes a synthetic genome set to a scratch directory, runs the CLI
entry point in-process and prints one JSON line with the stage timings.
usage: tools/e2e_wallclock.py N LENGTH [--continuous] [extra CLI flags]"""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.cli import build_parser  # noqa: E402
from phenotypeseeker_amd.synth import GenomeSet  # noqa: E402

n, length = int(sys.argv[1]), int(sys.argv[2])
extra = sys.argv[3:]
continuous = "--continuous" in extra  # tool flag: continuous phenotype column (Welch t-test + Lasso path)
extra = [e for e in extra if e != "--continuous"]
tmp = tempfile.mkdtemp(prefix="psk_e2e_")
gs = GenomeSet(n, length, seed=12345)
t0 = time.time()
rows = ["ID\tAddresses\tPheno"]
for i in range(n):
    name, fa = gs.sample(i)
    with open(os.path.join(tmp, name + ".fasta"), "wb") as f:
        f.write(fa)
    rows.append("%s\t%s.fasta\t%s" % (name, name, repr(round(gs.continuous_phenotype(i), 4)) if continuous else str(gs.phenotype(i))))
with open(os.path.join(tmp, "data.pheno"), "w") as f:
    f.write("\n".join(rows) + "\n")
t_write = time.time() - t0
os.chdir(tmp)
args = build_parser().parse_args(["modeling", "data.pheno"] + extra)
# stage timers (monkey-patched wrappers around the host pipeline's stages)
from phenotypeseeker_amd import modeling as _M  # noqa: E402
stage = {}


def _timed(cls, name):
    orig = getattr(cls, name)
    fn = orig.__func__ if hasattr(orig, "__func__") else orig

    def wrapper(*a, **k):
        t = time.time()
        r = fn(*a, **k)
        stage[name] = round(stage.get(name, 0.0) + time.time() - t, 3)
        return r
    setattr(cls, name, classmethod(wrapper) if isinstance(cls.__dict__[name], classmethod) else wrapper)


_timed(_M.Samples, "get_kmer_lists_batched")
_timed(_M.Samples, "get_feature_vector")
_timed(_M.phenotypes, "test_kmers_association_with_phenotype")
_timed(_M.phenotypes, "get_ML_df")
_timed(_M.phenotypes, "machine_learning_modelling")
from phenotypeseeker_amd import engine as _E  # noqa: E402
_timed(_E.PskContext, "count_kmers_batch")
_timed(_E.PskContext, "_fit")
err = sys.stderr
sys.stderr = open(os.devnull, "w")
t0 = time.time()
try:
    if os.environ.get("PSK_E2E_CPROFILE") == "1":   # host hot spots: top of the cumulative profile to stderr
        import cProfile
        import pstats
        prof = cProfile.Profile()
        prof.runcall(args.func, args)
        wall = time.time() - t0
        pstats.Stats(prof, stream=err).sort_stats("tottime").print_stats(22)
        pstats.Stats(prof, stream=err).sort_stats("cumulative").print_stats("phenotypeseeker_amd|joblib|sklearn", 45)
    else:
        args.func(args)
        wall = time.time() - t0
except BaseException as exc:   # the pipeline's own messages went to the muted stderr: say why it stopped
    sys.stderr = err
    print("modeling stopped after %.1f s: %r" % (time.time() - t0, exc), file=err)
    shutil.rmtree(tmp, ignore_errors=True)
    raise
sys.stderr = err
log = open("log.txt").read().strip().splitlines()
out = {"samples": n, "length": length, "flags": extra, "write_dataset_s": round(t_write, 2), "modeling_wall_s": round(wall, 3), "stages_s": stage,
       "log": log, "outputs": sorted(f for f in os.listdir(".") if not f.endswith(".fasta"))}
print(json.dumps(out))
shutil.rmtree(tmp, ignore_errors=True)
