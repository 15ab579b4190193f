#!/usr/bin/env python3
"""Turns gpurun_out/profiles_<tag>/ (tools/make_profiles.sh) into the committed summaries under profiles/:

  <tag>_<workload>_kernel_stats.csv   the rocprofv3 kernel-stats CSV of the workload
  <tag>_<workload>_pmc.csv            per kernel the mean per launch of every counter collected for it (one --pmc pass each)
  <tag>_<workload>_output.json        what the workload printed (its algorithmic bytes, wall-clock notes)
  <tag>_rooflines.md / .json          ONE table: calls, average duration, algorithmic bytes per launch (the workload's own
                                      figure, tools/profile_workloads.py), achieved GB/s and fraction of the 8 TB/s HBM peak, the
                                      HBM-side traffic per launch from the counters, and HOW the counters were turned into bytes
  <tag>_pmc_calibration.md / .json    the calibration behind that conversion: tools/calib/pmc_calib's known-byte-count access
                                      patterns under the same counters
  <tag>_moments_lds.md                the SQ's LDS / wait counters of the moment scans (what bounds them)
  <tag>_traffic_chi2_scan.json        the scan kernel's traffic, quoted by bench.py as roofline.traffic (with the hash of the
                                      kernel source it was measured on)

Counter -> bytes (r05, after calibration; MI355X_MICROARCH.md calibrates the wide coalesced read only and says "calibrate on a
known byte count in your own access pattern before trusting an absolute"):
  reads   2 x FETCH_SIZE x 1024 = TCC_EA0_RDREQ x 128 B.  Every L2 -> fabric read request of these kernels is a whole 128-byte
          line (TCC_EA0_RDREQ_32B = 0 throughout) and FETCH_SIZE tallies it at 64 B.  Verified on known byte counts for coalesced
          reads of 16 / 8 / 4 B per lane and for 64-byte per-lane segments (exactly bytes / 128 requests); a lane walking its own
          list re-fetches lines (1.19 x its bytes in the calibration); a random 4- or 8-byte probe costs one request -- and 67 M of
          them take 1.35 ms, which is the HBM ceiling (6.3 TB/s) at 128 B apiece and half of it at 64.
  writes  WRITE_SIZE x 1024 as counted = (WRREQ - WRREQ_64B) x 32 B + WRREQ_64B x 64 B: exact for stores that fill 64-byte
          requests (coalesced, any lane width); a store that fills less is charged ONE 32-byte sector (8-byte scatter: 4 x its
          bytes; runs of 16 bytes: 2 x); a global integer atomic is a 32-byte write request and no read.
usage: tools/summarise_profiles.py r05"""
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
READ_RULE = "reads = 2 x FETCH_SIZE (= RDREQ x 128 B: a request is a 128-B line, tallied at 64 B)"
WRITE_RULE = "writes = WRITE_SIZE as counted (64-B requests exact; a partial store is charged one 32-B sector)"


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def base(name):
    return short(name).split("<")[0]


def first(pattern):
    g = glob.glob(pattern, recursive=True)
    return g[0] if g else None


def counters_of(workload_dir):
    """{kernel: {counter: (launches, mean per launch)}} over every pmc_* pass of a workload."""
    acc = defaultdict(lambda: defaultdict(list))
    for d in sorted(glob.glob(os.path.join(workload_dir, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        f = first(os.path.join(d, "**", "*counter_collection.csv"))
        if not f:
            continue
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: (len(v), sum(v) / len(v)) for c, v in cs.items()} for k, cs in acc.items()}


def traffic_of(c):
    """HBM-side bytes per launch of one kernel from its counters, or None."""
    if "FETCH_SIZE" not in c:
        return None
    return int((2 * c["FETCH_SIZE"][1] + c.get("WRITE_SIZE", (0, 0.0))[1]) * 1024)


def stats_rows(workload_dir):
    stats = first(os.path.join(workload_dir, "trace", "**", "*kernel_stats.csv"))
    if not stats:
        return None, []
    with open(stats) as f:
        return stats, list(csv.DictReader(f))


def calibration():
    d = os.path.join(src, "calib")
    stats, rows = stats_rows(d)
    if not stats:
        return
    shutil.copy(stats, os.path.join(dst, "%s_calib_kernel_stats.csv" % tag))
    with open(os.path.join(d, "stdout.txt")) as f:
        patterns = json.load(f)["patterns"]
    cnt = counters_of(d)
    us = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}
    out = []
    for kernel, p in patterns.items():
        c = {k: v[1] for k, v in cnt.get(kernel, {}).items()}
        rd, wr = p["read_bytes"], p["write_bytes"]
        rec = {"kernel": kernel, "pattern": p["pattern"], "avg_us": round(us.get(kernel, 0.0), 1), "true_read_bytes": rd,
               "true_write_bytes": wr, "ops": p["ops"], "FETCH_SIZE_bytes": c.get("FETCH_SIZE", 0.0) * 1024,
               "WRITE_SIZE_bytes": c.get("WRITE_SIZE", 0.0) * 1024, "RDREQ": c.get("TCC_EA0_RDREQ_sum"),
               "RDREQ_32B": c.get("TCC_EA0_RDREQ_32B_sum"), "WRREQ": c.get("TCC_EA0_WRREQ_sum"),
               "WRREQ_64B": c.get("TCC_EA0_WRREQ_64B_sum"), "TCC_HIT": c.get("TCC_HIT_sum"), "TCC_MISS": c.get("TCC_MISS_sum")}
        out.append(rec)
    with open(os.path.join(dst, "%s_pmc_calibration.json" % tag), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(dst, "%s_pmc_calibration.md" % tag), "w") as f:
        f.write("# %s: what FETCH_SIZE / WRITE_SIZE count on gfx950, by access pattern (tools/calib/pmc_calib, known bytes per launch)\n\n" % tag)
        f.write("Buffers of 2 GiB (streams) / 6 GiB (probe and scatter targets): far beyond the 256-MiB Infinity Cache.  One `--pmc` pass per "
                "counter group (FETCH_SIZE | WRITE_SIZE | RDREQ, RDREQ_32B | WRREQ, WRREQ_64B | TCC_HIT, TCC_MISS), means over 3 launches.\n\n")
        f.write("| pattern | us | true bytes | FETCH_SIZE bytes | RDREQ (32-B ones) | true / FETCH | B per request | WRITE_SIZE bytes | WRREQ (64-B ones) | WRITE / true | L2 hit / miss |\n")
        f.write("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        for r in out:
            true = r["true_read_bytes"] + r["true_write_bytes"]
            fe, wr = r["FETCH_SIZE_bytes"], r["WRITE_SIZE_bytes"]
            f.write("| %s | %.1f | %d | %s | %s | %s | %s | %s | %s | %s | %s |\n" % (
                r["pattern"], r["avg_us"], true, "%d" % fe if fe else "-",
                "%.3g (%.3g)" % (r["RDREQ"], r["RDREQ_32B"] or 0) if r["RDREQ"] else "-",
                "%.3f" % (r["true_read_bytes"] / fe) if fe and r["true_read_bytes"] else "-",
                "%.1f true; %.0f if lines" % (r["true_read_bytes"] / r["RDREQ"], 128) if r["RDREQ"] and r["true_read_bytes"] else "-",
                "%d" % wr if wr else "-", "%.3g (%.3g)" % (r["WRREQ"], r["WRREQ_64B"] or 0) if r["WRREQ"] else "-",
                "%.3f" % (wr / r["true_write_bytes"]) if wr and r["true_write_bytes"] else "-",
                "%.3g / %.3g" % (r["TCC_HIT"] or 0, r["TCC_MISS"] or 0) if r["TCC_MISS"] is not None else "-"))
        f.write("\nReading: (1) every coalesced read -- 16, 8 or 4 B per lane -- and the 64-byte per-lane segments issue exactly bytes / 128 "
                "requests and FETCH_SIZE reports exactly half the bytes: a request is a 128-byte line tallied at 64 B, whatever the lane "
                "width, so **read bytes = 2 x FETCH_SIZE = RDREQ x 128 B** for every kernel (no 32-byte requests appear anywhere).  (2) "
                "A lane walking its own stream (pm_mark's shape) issues 1.19 x bytes / 128 requests: lines are evicted between a lane's "
                "eight visits and fetched again -- over-fetch that is real, not a counting artefact.  (3) A random 4- or 8-byte probe is one "
                "request; 67 M of them take 1.34-1.36 ms = 6.3 TB/s at 128 B apiece (the stream ceiling of MI355X_MICROARCH.md) and 3.2 TB/s "
                "at 64: the line reading again.  (4) WRITE_SIZE is exact for stores that fill 64-byte requests (any lane width); a store "
                "that does not is charged one 32-byte sector -- 8-byte scatter (pm_replay's shape) 4.0 x its bytes, 4-byte scatter 8.0 x, "
                "16-byte runs 2.0 x, 64-byte runs 1.0 x -- and reads nothing.  (5) A global integer atomic is one 32-byte write request, "
                "no read request.\n")


table = []
os.makedirs(dst, exist_ok=True)
calibration()
for wl in sorted(os.listdir(src)):
    d = os.path.join(src, wl)
    if not os.path.isdir(d) or wl == "calib":
        continue
    stats, rows = stats_rows(d)
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
    if wl in ("bench", "hbmonly") and info:
        alg = {"chi2_scan_kernel": info["roofline"]["algorithmic_bytes_per_launch"]}
    cnt = counters_of(d)
    if cnt:
        names = sorted({c for cs in cnt.values() for c in cs})
        with open(os.path.join(dst, "%s_%s_pmc.csv" % (tag, wl)), "w") as f:
            f.write("kernel,launches," + ",".join("mean_" + c for c in names) + ",hbm_bytes_per_launch(2xFETCH_SIZE_KB+WRITE_SIZE_KB)x1024\n")
            for k in sorted(cnt, key=lambda k_: -(traffic_of(cnt[k_]) or 0)):
                t = traffic_of(cnt[k])
                f.write("%s,%d,%s,%s\n" % ('"%s"' % k, max(v[0] for v in cnt[k].values()),
                                            ",".join("%.1f" % cnt[k][c][1] if c in cnt[k] else "" for c in names), t if t is not None else ""))
    for row in rows:
        name = short(row["Name"])
        if name.startswith("__amd_rocclr"):
            continue
        avg_us = float(row["AverageNs"]) / 1e3
        a = alg.get(base(name))
        ach = a / (avg_us * 1e-6) if a else None
        c = cnt.get(name, {})
        rec = {"workload": wl, "kernel": name, "calls": int(row["Calls"]), "avg_us": round(avg_us, 2),
               "total_ms": round(float(row["TotalDurationNs"]) / 1e6, 3), "algorithmic_bytes_per_launch": a,
               "achieved_GBps": round(ach / 1e9, 1) if ach else None, "frac_of_8TBps": round(ach / PEAK, 4) if ach else None,
               "hbm_traffic_bytes_per_launch": traffic_of(c),
               "read_bytes_per_launch": int(2 * c["FETCH_SIZE"][1] * 1024) if "FETCH_SIZE" in c else None,
               "write_bytes_per_launch": int(c["WRITE_SIZE"][1] * 1024) if "WRITE_SIZE" in c else None}
        for key, cn in (("RDREQ", "TCC_EA0_RDREQ_sum"), ("RDREQ_32B", "TCC_EA0_RDREQ_32B_sum"), ("WRREQ", "TCC_EA0_WRREQ_sum"),
                        ("WRREQ_64B", "TCC_EA0_WRREQ_64B_sum")):
            if cn in c:
                rec[key] = round(c[cn][1], 1)
        table.append(rec)
    if wl in ("bench", "hbmonly") and info:
        if wl == "bench":
            unprof, kept = os.path.join(src, "bench_unprofiled.json"), os.path.join(dst, "%s_bench_cfg2_unprofiled.json" % tag)
            if not os.path.exists(kept) or os.path.getmtime(unprof) > os.path.getmtime(kept):   # (a later record committed by hand stays)
                shutil.copy(unprof, kept)
        scan = [k for k in cnt if k.startswith("chi2_scan_kernel") and "FETCH_SIZE" in cnt[k]]
        if scan:
            k = max(scan, key=lambda k_: cnt[k_]["FETCH_SIZE"][0])
            with open(os.path.join(root, "phenotypeseeker_amd", "csrc", "assoc_scan.hip"), "rb") as f:
                sha = hashlib.sha256(f.read()).hexdigest()[:16]
            cfg = info["config"]
            fe, wr = cnt[k]["FETCH_SIZE"][1], cnt[k].get("WRITE_SIZE", (0, 0.0))[1]
            # (one file per matrix: bench.py quotes the one whose rows and row width are its own)
            with open(os.path.join(dst, "%s_traffic_chi2_scan%s.json" % (tag, "" if wl == "bench" else "_" + wl)), "w") as f:
                json.dump({"round": int(tag[1:3]), "kernel": k, "workload": cfg["workload"], "rows": cfg["rows_per_gpu"],
                           "words_per_row_stored": cfg["words_per_row_stored"], "kernel_source_sha16": sha,
                           "FETCH_SIZE_KB_mean_per_launch": fe, "WRITE_SIZE_KB_mean_per_launch": wr,
                           "correction": "MI355X_MICROARCH.md HBM section + profiles/%s_pmc_calibration.md: FETCH_SIZE / WRITE_SIZE are KB; on "
                                         "gfx950 a read request is a 128-B line tallied at 64 B, so read bytes = 2 x FETCH_SIZE x 1024; "
                                         "WRITE_SIZE is exact for full 64-B requests.  Separate --pmc passes, no tracing flags." % tag,
                           "hbm_bytes_per_launch": int((2 * fe + wr) * 1024)}, f, indent=1)
    # the device inflate: what its kernels wait for
    if wl == "gzinflate" and any("SQ_WAVE_CYCLES" in c for c in cnt.values()):
        us = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}
        with open(os.path.join(dst, "%s_gzinflate_sq.md" % tag), "w") as f:
            f.write("# %s: the kernels of the device inflate under the SQ's wait and LDS counters (one --pmc pass; means of three launches)\n\n" % tag)
            f.write("Workload: `tools/gz_bench.py fastq 64 128 6 3 noverify` -- 64 .fastq.gz images, 1.93 GB -> 9.2 GB of text.  Shares of "
                    "SQ_WAVE_CYCLES: parked = SQ_WAIT_ANY (s_waitcnt: memory, LDS), issue-stalled = SQ_WAIT_INST_ANY, issuing = SQ_ACTIVE_INST_ANY; "
                    "LDS busy = SQ_LDS_IDX_ACTIVE, of which bank conflicts = SQ_LDS_BANK_CONFLICT.\n\n")
            f.write("| kernel | us | parked | issue-stalled | issuing | LDS busy / wave cycles | bank conflicts / LDS busy | read bytes (2 x FETCH_SIZE) | write bytes |\n")
            f.write("|---|---:|---:|---:|---:|---:|---:|---:|---:|\n")
            for k in ("gz_find_kernel", "gz_decode_kernel<false>", "gz_decode_kernel<true>", "gz_copy_kernel", "gz_tails_kernel", "gz_resolve_kernel", "gz_crc_kernel"):
                c = {n: v[1] for n, v in cnt.get(k, {}).items()}
                if "SQ_WAVE_CYCLES" not in c:
                    continue
                wc = c["SQ_WAVE_CYCLES"]
                f.write("| `%s` | %.0f | %.2f | %.2f | %.2f | %.3f | %.2f | %.2f GB | %.2f GB |\n" % (
                    k, us.get(k, 0.0), c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                    c.get("SQ_LDS_IDX_ACTIVE", 0) / wc, c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                    2 * c.get("FETCH_SIZE", 0) * 1024 / 1e9, c.get("WRITE_SIZE", 0) * 1024 / 1e9))
            f.write("\nReading: the two decoding passes are ISSUING for about two fifths of their wave cycles with ONE wave a SIMD (their LDS tables allow no "
                    "more): what a lane waits for is mostly its own chain of dependent instructions (a symbol is ~40 instructions of the literal path plus ~100 "
                    "of the match path, both executed by a wave whose lanes disagree) and, for 40-50 %, the s_waitcnt of an LDS look-up or of the 4-byte input "
                    "load of whichever lane has run dry -- in the writing pass that wait also covers the stores issued since (one counter for loads and "
                    "stores); the LDS itself is idle (1 % busy).  `gz_copy_kernel` is parked on memory for three quarters of its cycles at 5 waves a SIMD and "
                    "fetches 4.5 x the 16-bit text it copies from -- latency, hidden only by the number of resident waves (4 / 8 / 16 waves a CU: 105 / 79 / "
                    "53 ms).  `gz_resolve_kernel` and `gz_crc_kernel` stream (0.66 / 0.61 of 8 TB/s in algorithmic bytes); the search is two fifths issuing, two "
                    "fifths waiting.\n")
    # the moment scans: what the SQ says bounds them
    if wl == "moments" and any("SQ_WAVE_CYCLES" in c for c in cnt.values()):
        with open(os.path.join(dst, "%s_moments_lds.md" % tag), "w") as f:
            f.write("# %s: the moment scans (16 M x 1,024) under the SQ's LDS and wait counters (one --pmc pass, means per launch)\n\n" % tag)
            f.write("Shares of SQ_WAVE_CYCLES (quad-cycles summed over waves): parked = SQ_WAIT_ANY (s_waitcnt / barrier), issue-stalled = "
                    "SQ_WAIT_INST_ANY, of which on the LDS = SQ_WAIT_INST_LDS, issuing = SQ_ACTIVE_INST_ANY; LDS busy = SQ_LDS_IDX_ACTIVE, of "
                    "which bank conflicts = SQ_LDS_BANK_CONFLICT.\n\n")
            f.write("| kernel | us | LDS instr. | LDS busy / wave cycles | bank conflicts / LDS busy | stalled on LDS | parked (memory) | issue-stalled | issuing |\n")
            f.write("|---|---:|---:|---:|---:|---:|---:|---:|---:|\n")
            us = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}
            for k in sorted(cnt, key=lambda k_: -us.get(k_, 0)):
                c = {n: v[1] for n, v in cnt[k].items()}
                if "SQ_WAVE_CYCLES" not in c or not ("scan_kernel" in k or "finalize" in k):
                    continue
                wc = c["SQ_WAVE_CYCLES"]
                f.write("| `%s` | %.1f | %.3g | %.3f | %.3f | %.3f | %.3f | %.3f | %.3f |\n" % (
                    k, us.get(k, 0.0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_LDS_IDX_ACTIVE", 0) / wc,
                    c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1), c.get("SQ_WAIT_INST_LDS", 0) / wc,
                    c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc))
            f.write("\nReading: the LDS is busy 6-14 % of the wave cycles and holds up issue for 0.5-2 % of them; the waves of every scan -- "
                    "plain or with moments -- are parked on memory for 52-57 %.  The moment scans are NOT LDS-bound (DESIGN.md said so until "
                    "r04 without a counter): they are the plain scan of this shape (0.66-0.68 of the HBM peak at 1 % surviving rows) plus "
                    "10-15 % more issue slots for the f64 moment work, under the same memory-latency bound.  The weighted forms do conflict in "
                    "the LDS (17-31 % of its busy cycles: the per-wave row queue), which is 2-4 % of their wave cycles.\n")

with open(os.path.join(dst, tag + "_rooflines.json"), "w") as f:
    json.dump(table, f, indent=1)
with open(os.path.join(dst, tag + "_rooflines.md"), "w") as f:
    f.write("# %s: kernels by workload (rocprofv3 --kernel-trace --stats; bytes: tools/profile_workloads.py; peak 8 TB/s)\n\n" % tag)
    f.write("HBM-side traffic per launch from the counters: %s; %s -- calibrated on known byte counts in `%s_pmc_calibration.md`.  "
            "`read req` = TCC_EA0_RDREQ per launch (x 128 B = the read column; 32-byte requests: none in any kernel), `write req (64 B)` = "
            "TCC_EA0_WRREQ and how many of them are full 64-byte requests (the rest are 32-byte sectors of partial stores).\n\n" % (READ_RULE, WRITE_RULE, tag))
    f.write("| workload | kernel | calls | avg us | algorithmic bytes / launch | achieved GB/s | frac of 8 TB/s | read bytes (2 x FETCH) | write bytes (WRITE_SIZE) | traffic / algorithmic | read req | write req (64 B) | correction and why |\n")
    f.write("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---|\n")
    for r in table:
        if r["total_ms"] < 0.05 and not r["algorithmic_bytes_per_launch"]:
            continue
        tr, a = r["hbm_traffic_bytes_per_launch"], r["algorithmic_bytes_per_launch"]
        why = "-"
        if r["read_bytes_per_launch"] is not None:
            why = "reads x 2: 128-B line requests tallied at 64 B (every access shape, calibration rows 1-9)"
            if "WRREQ" in r and r["WRREQ"] > 0:
                full = r.get("WRREQ_64B", 0) / r["WRREQ"]
                why += "; writes x 1: %.0f %% of the write requests are full 64-B ones" % (100 * full) if full >= 0.9 else \
                       "; writes as counted, but %.0f %% of the requests are 32-B sectors of partial stores (x 4 for 8-byte stores, calibration row 14)" % (100 * (1 - full))
            elif r["write_bytes_per_launch"]:
                why += "; writes x 1 (request sizes not collected for this workload)"
        f.write("| %s | `%s` | %d | %.2f | %s | %s | %s | %s | %s | %s | %s | %s | %s |\n" % (
            r["workload"], r["kernel"], r["calls"], r["avg_us"], "%d" % a if a else "-",
            "%.0f" % r["achieved_GBps"] if r["achieved_GBps"] else "-", "%.3f" % r["frac_of_8TBps"] if r["frac_of_8TBps"] else "-",
            "%d" % r["read_bytes_per_launch"] if r["read_bytes_per_launch"] is not None else "-",
            "%d" % r["write_bytes_per_launch"] if r["write_bytes_per_launch"] is not None else "-",
            "%.2f" % (tr / a) if tr and a else "-",
            "%.4g" % r["RDREQ"] if "RDREQ" in r else "-",
            "%.4g (%.4g)" % (r["WRREQ"], r.get("WRREQ_64B", 0)) if "WRREQ" in r else "-", why))
print(open(os.path.join(dst, tag + "_rooflines.md")).read())
