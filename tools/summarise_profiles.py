#!/usr/bin/env python3
"""Turns gpurun_out/profiles_<tag>/ (tools/make_profiles.sh) into the committed summaries under profiles/:
kernel stats CSV, per-kernel FETCH_SIZE / WRITE_SIZE means, the bench lines, and the scan kernel's HBM
traffic per launch (gfx950 correction: read bytes = 2 x FETCH_SIZE, MI355X_MICROARCH.md).
usage: tools/summarise_profiles.py r01"""
import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "profiles_" + tag)
dst = os.path.join(root, "profiles")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_bench_cfg2_kernel_stats.csv"))
for n in ("unprofiled", "under_rocprof"):
    shutil.copy(os.path.join(src, "bench_%s.json" % n), os.path.join(dst, "%s_bench_cfg2_%s.json" % (tag, n)))
means = {}
for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    acc = defaultdict(list)
    with open(os.path.join(src, sub, "bench_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    rows = sorted(((k, len(v), sum(v) / len(v), sum(v)) for k, v in acc.items()), key=lambda r: -r[3])
    with open(os.path.join(dst, "%s_bench_cfg2_pmc_%s.csv" % (tag, counter)), "w") as f:
        f.write("kernel,calls,mean_%s_KB,sum_%s_KB\n" % (counter, counter))
        for k, c, m, s in rows:
            f.write("%s,%d,%.1f,%.1f\n" % (k, c, m, s))
    means[counter] = {k: m for k, c, m, s in rows}
bench = json.load(open(os.path.join(src, "bench_unprofiled.json")))
kern = [k for k in means["FETCH_SIZE"] if k.startswith("chi2_scan_kernel")][0]
fetch, write = means["FETCH_SIZE"][kern], means["WRITE_SIZE"].get(kern, 0.0)
out = {"round": int(tag[1:3]), "kernel": kern, "workload": bench["config"]["workload"],
       "rows": bench["config"]["rows_per_gpu"], "words_per_row_stored": bench["config"]["words_per_row_stored"],
       "FETCH_SIZE_KB_mean_per_launch": fetch, "WRITE_SIZE_KB_mean_per_launch": write,
       "correction": "MI355X_MICROARCH.md HBM section: rocprofv3 FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE "
                     "reports exactly half of the bytes of a wide coalesced (16 B/lane) streaming read, so read bytes = "
                     "2 x FETCH_SIZE x 1024; WRITE_SIZE is exact. Separate --pmc passes (FETCH_SIZE, then WRITE_SIZE), no "
                     "tracing flags.",
       "hbm_bytes_per_launch": int(round(2 * fetch * 1024 + write * 1024)),
       "source": ["profiles/%s_bench_cfg2_pmc_FETCH_SIZE.csv" % tag, "profiles/%s_bench_cfg2_pmc_WRITE_SIZE.csv" % tag]}
json.dump(out, open(os.path.join(dst, "%s_traffic_chi2_scan.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1))
with open(os.path.join(dst, tag + "_bench_cfg2_kernel_stats.csv")) as f:
    for row in csv.DictReader(f):
        if "chi2_scan_kernel" in row["Name"]:
            print("rocprof avg of the scan kernel: %.1f us over %s calls; bench.py (HIP events): %.1f us" % (
                float(row["AverageNs"]) / 1e3, row["Calls"], bench["roofline"]["kernel_ms"] * 1e3))
