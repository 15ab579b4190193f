#!/usr/bin/env python3
"""the per-rank phase table of the sharded end-to-end leg (`bench.py --gpus N --share-gpu`: N ranks of
`phenotypeseeker modeling` as child processes, here all on the ONE GPU of a test box, collectives through host files) at
N = 1, 2, 4, 8 -- so that start-up, rendezvous and ingest costs are known before the first run on eight GPUs.
Writes gpurun_out/r04_phases/phases.json and phases.md (copied to profiles/ by hand).
usage (GPU box, repo root): tools/phase_table.py [samples] [length]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
samples = sys.argv[1] if len(sys.argv) > 1 else "128"
length = sys.argv[2] if len(sys.argv) > 2 else "300000"
out_dir = os.path.join(ROOT, "gpurun_out", "r04_phases")
os.makedirs(out_dir, exist_ok=True)
size = ["--samples", samples, "--length", length, "--kmer", "16", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
runs = []
for n, ingest in ((1, None), (2, "filter"), (2, "exchange"), (4, "filter"), (4, "exchange"), (8, "filter"), (8, "exchange")):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + size
    if n > 1:
        cmd += ["--share-gpu", "--ingest", ingest]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode or not line:
        runs.append({"ranks": n, "ingest": ingest, "error": r.stderr[-500:]})
        continue
    d = json.loads(line[-1])
    runs.append({"ranks": n, "ingest": ingest, "e2e": d.get("e2e"), "scaling": d.get("scaling")})
    print("ranks %d %s: e2e %.2f s" % (n, ingest, d["e2e"].get("modeling_wall_s", -1)), flush=True)
with open(os.path.join(out_dir, "phases.json"), "w") as f:
    json.dump({"samples": int(samples), "length": int(length), "k": 16, "runs": runs}, f, indent=1)
# the table: one row per phase, one column per run (the slowest rank's table)
cols, names = [], []
for run in runs:
    e = run.get("e2e") or {}
    ph = {k: v for k, v in (e.get("phases") or {}).items() if v}
    if not ph:
        continue
    slow = max(ph.values(), key=lambda v: v["total_s"])
    cols.append(("%d rank%s%s" % (run["ranks"], "s" if run["ranks"] > 1 else "", (", " + run["ingest"]) if run["ingest"] else ""), slow, e["modeling_wall_s"]))
    for k in slow["phases_s"]:
        base = k.split(" (")[0] if k.startswith("ingest") else k
        if base not in names:
            names.append(base)
with open(os.path.join(out_dir, "phases.md"), "w") as f:
    f.write("| phase (s, slowest rank) | " + " | ".join(c[0] for c in cols) + " |\n|---|" + "---|" * len(cols) + "\n")
    for nm in names:
        row = []
        for _, slow, _ in cols:
            v = [val for k, val in slow["phases_s"].items() if (k.split(" (")[0] if k.startswith("ingest") else k) == nm]
            row.append("%.3f" % v[0] if v else "")
        f.write("| %s | %s |\n" % (nm, " | ".join(row)))
    f.write("| **total of the table** | " + " | ".join("%.3f" % c[1]["total_s"] for c in cols) + " |\n")
    f.write("| **wall-clock of the leg (process start to exit, slowest rank)** | " + " | ".join("%.3f" % c[2] for c in cols) + " |\n")
print(open(os.path.join(out_dir, "phases.md")).read())
