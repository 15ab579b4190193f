#!/usr/bin/env python3
"""Turns gpurun_out/profiles_<tag>/ (tools/make_profiles.sh) into the committed summaries under profiles/:
per workload the rocprofv3 kernel-stats CSV and the FETCH_SIZE / WRITE_SIZE means per kernel; one table
(<tag>_rooflines.md / .json) with, per kernel: calls, average duration, algorithmic bytes per launch (the workload's
own figure, tools/profile_workloads.py), achieved GB/s and its fraction of the 8 TB/s HBM peak, and the HBM traffic
per launch from the counters (gfx950 correction: read bytes = 2 x FETCH_SIZE for wide streaming reads,
MI355X_MICROARCH.md); and the scan kernel's traffic file that bench.py quotes as roofline.traffic (with the hash of
the kernel source it was measured on).
usage: tools/summarise_profiles.py r02"""
import csv
import glob
import hashlib
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
PEAK = 8.0e12


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def base(name):
    return short(name).split("<")[0]


def first(pattern):
    g = glob.glob(pattern, recursive=True)
    return g[0] if g else None


def counter_means(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


table = []
for wl in sorted(os.listdir(src)):
    d = os.path.join(src, wl)
    if not os.path.isdir(d):
        continue
    stats = first(os.path.join(d, "trace", "**", "*kernel_stats.csv"))
    if not stats:
        continue
    shutil.copy(stats, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, wl)))
    info = {}
    try:
        with open(os.path.join(d, "stdout.txt")) as f:
            lines = [l for l in f.read().splitlines() if l.startswith("{")]
        info = json.loads(lines[-1]) if lines else {}
    except (OSError, ValueError):
        pass
    with open(os.path.join(dst, "%s_%s_output.json" % (tag, wl)), "w") as f:
        json.dump(info, f, indent=1, sort_keys=True)
    alg = info.get("algorithmic_bytes_per_launch", {})
    if wl == "bench" and info:
        alg = {"chi2_scan_kernel": info["roofline"]["algorithmic_bytes_per_launch"]}
    fetch = write = {}
    fp, wp = first(os.path.join(d, "pmc_fetch", "**", "*counter_collection.csv")), first(os.path.join(d, "pmc_write", "**", "*counter_collection.csv"))
    if fp and wp:
        fetch, write = counter_means(fp, "FETCH_SIZE"), counter_means(wp, "WRITE_SIZE")
        with open(os.path.join(dst, "%s_%s_pmc.csv" % (tag, wl)), "w") as f:
            f.write("kernel,calls,mean_FETCH_SIZE_KB,mean_WRITE_SIZE_KB,hbm_bytes_per_launch(2xFETCH+WRITE)\n")
            for k in sorted(set(fetch) | set(write), key=lambda k_: -(fetch.get(k_, (0, 0))[1] + write.get(k_, (0, 0))[1])):
                fe, wr = fetch.get(k, (0, 0.0)), write.get(k, (0, 0.0))
                f.write("%s,%d,%.1f,%.1f,%d\n" % (k, max(fe[0], wr[0]), fe[1], wr[1], int((2 * fe[1] + wr[1]) * 1024)))
    with open(stats) as f:
        for row in csv.DictReader(f):
            name = short(row["Name"])
            if name.startswith("__amd_rocclr"):
                continue
            avg_us = float(row["AverageNs"]) / 1e3
            a = alg.get(base(name))
            ach = a / (avg_us * 1e-6) if a else None
            fe, wr = fetch.get(name), write.get(name)
            traffic = int((2 * fe[1] + (wr[1] if wr else 0.0)) * 1024) if fe else None
            table.append({"workload": wl, "kernel": name, "calls": int(row["Calls"]), "avg_us": round(avg_us, 2),
                          "total_ms": round(float(row["TotalDurationNs"]) / 1e6, 3), "algorithmic_bytes_per_launch": a,
                          "achieved_GBps": round(ach / 1e9, 1) if ach else None, "frac_of_8TBps": round(ach / PEAK, 4) if ach else None,
                          "hbm_traffic_bytes_per_launch": traffic})
    if wl == "bench" and info:
        shutil.copy(os.path.join(src, "bench_unprofiled.json"), os.path.join(dst, "%s_bench_cfg2_unprofiled.json" % tag))
        scan = [k for k in fetch if k.startswith("chi2_scan_kernel")]
        if scan:
            k = max(scan, key=lambda k_: fetch[k_][0])
            with open(os.path.join(root, "phenotypeseeker_amd", "csrc", "assoc_scan.hip"), "rb") as f:
                sha = hashlib.sha256(f.read()).hexdigest()[:16]
            cfg = info["config"]
            with open(os.path.join(dst, "%s_traffic_chi2_scan.json" % tag), "w") as f:
                json.dump({"round": int(tag[1:3]), "kernel": k, "workload": cfg["workload"], "rows": cfg["rows_per_gpu"],
                           "words_per_row_stored": cfg["words_per_row_stored"], "kernel_source_sha16": sha,
                           "FETCH_SIZE_KB_mean_per_launch": fetch[k][1], "WRITE_SIZE_KB_mean_per_launch": write.get(k, (0, 0.0))[1],
                           "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports half "
                                         "of the bytes of a wide (16 B per lane) streaming read, so read bytes = 2 x FETCH_SIZE x 1024; "
                                         "WRITE_SIZE is exact.  Separate --pmc passes, no tracing flags.",
                           "hbm_bytes_per_launch": int((2 * fetch[k][1] + write.get(k, (0, 0.0))[1]) * 1024)}, f, indent=1)

with open(os.path.join(dst, tag + "_rooflines.json"), "w") as f:
    json.dump(table, f, indent=1)
with open(os.path.join(dst, tag + "_rooflines.md"), "w") as f:
    f.write("# %s: kernels by workload (rocprofv3 --kernel-trace --stats; bytes: tools/profile_workloads.py; peak 8 TB/s)\n\n" % tag)
    f.write("| workload | kernel | calls | avg us | algorithmic bytes / launch | achieved GB/s | frac of 8 TB/s | HBM traffic / launch (PMC) |\n")
    f.write("|---|---|---:|---:|---:|---:|---:|---:|\n")
    for r in table:
        if r["total_ms"] < 0.05 and not r["algorithmic_bytes_per_launch"]:
            continue
        f.write("| %s | `%s` | %d | %.2f | %s | %s | %s | %s |\n" % (
            r["workload"], r["kernel"], r["calls"], r["avg_us"],
            "%d" % r["algorithmic_bytes_per_launch"] if r["algorithmic_bytes_per_launch"] else "-",
            "%.0f" % r["achieved_GBps"] if r["achieved_GBps"] else "-",
            "%.3f" % r["frac_of_8TBps"] if r["frac_of_8TBps"] else "-",
            "%d" % r["hbm_traffic_bytes_per_launch"] if r["hbm_traffic_bytes_per_launch"] else "-"))
print(open(os.path.join(dst, tag + "_rooflines.md")).read())
