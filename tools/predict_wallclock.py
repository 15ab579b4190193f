#!/usr/bin/env python3
"""`phenotypeseeker modeling` then `phenotypeseeker prediction` on the same synthetic samples: wall-clock of the
prediction command (dictionary counting of every sample + the stored model).  usage: tools/predict_wallclock.py N LENGTH"""
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
tmp = tempfile.mkdtemp(prefix="psk_pred_")
gs = GenomeSet(n, length, seed=12345)
rows, srows = ["ID\tAddresses\tPheno"], []
for i in range(n):
    name, fa = gs.sample(i)
    with open(os.path.join(tmp, name + ".fasta"), "wb") as f:
        f.write(fa)
    rows.append("%s\t%s.fasta\t%d" % (name, name, gs.phenotype(i)))
    srows.append("%s\t%s.fasta" % (name, name))
open(os.path.join(tmp, "data.pheno"), "w").write("\n".join(rows) + "\n")
open(os.path.join(tmp, "samples.txt"), "w").write("\n".join(srows) + "\n")
os.chdir(tmp)
err = sys.stderr
sys.stderr = open(os.devnull, "w")
a = build_parser().parse_args(["modeling", "data.pheno"])
t = time.time(); a.func(a); t_model = time.time() - t
open("models.txt", "w").write("Pheno\tlog_reg_model_Pheno.pkl\n")
a = build_parser().parse_args(["prediction", "samples.txt", "models.txt"])
t = time.time(); a.func(a); t_pred = time.time() - t
sys.stderr = err
out = open("predictions_Pheno.txt").read().splitlines()
print(json.dumps({"samples": n, "length": length, "modeling_s": round(t_model, 3), "prediction_s": round(t_pred, 3),
                  "prediction_lines": len(out), "first": out[:2]}))
shutil.rmtree(tmp, ignore_errors=True)
