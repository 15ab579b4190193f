#!/usr/bin/env python3
"""`phenotypeseeker modeling` under torch.distributed.run with R ranks on ONE visible GPU (gloo collectives,
PSK_SHARE_GPU): the ingest with the list exchange against every-rank-counts-everything.  Both share one GPU and
stage the collectives through the host, so this shows the host-side cost and the logic, not xGMI.
usage: tools/multirank_wallclock.py N LENGTH [RANKS]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.synth import GenomeSet  # noqa: E402

n, length = int(sys.argv[1]), int(sys.argv[2])
ranks = int(sys.argv[3]) if len(sys.argv) > 3 else 2
tmp = tempfile.mkdtemp(prefix="psk_mr_")
gs = GenomeSet(n, length, seed=12345)
rows = ["ID\tAddresses\tPheno"]
for i in range(n):
    name, fa = gs.sample(i)
    with open(os.path.join(tmp, name + ".fasta"), "wb") as f:
        f.write(fa)
    rows.append("%s\t%s.fasta\t%s" % (name, name, gs.phenotype(i)))
with open(os.path.join(tmp, "data.pheno"), "w") as f:
    f.write("\n".join(rows) + "\n")
for mode in ("0", "1"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PSK_SHARE_GPU="1", PSK_DIST_BACKEND="gloo", OMP_NUM_THREADS="1",
               PSK_REDUNDANT_INGEST=mode)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
           "127.0.0.1", "--master-port", "29655", os.path.join(ROOT, "scripts", "phenotypeseeker"), "modeling", "data.pheno"]
    if os.path.exists(os.path.join(tmp, "log.txt")):
        os.remove(os.path.join(tmp, "log.txt"))
    t = time.time()
    r = subprocess.run(cmd, env=env, cwd=tmp, capture_output=True, text=True)
    wall = time.time() - t
    log = open(os.path.join(tmp, "log.txt")).read().strip().splitlines() if r.returncode == 0 else [r.stderr[-1500:]]
    print("PSK_REDUNDANT_INGEST=%s  %d ranks  process wall %.1f s (torch + rendezvous included)  %s" % (mode, ranks, wall, log))
shutil.rmtree(tmp, ignore_errors=True)
