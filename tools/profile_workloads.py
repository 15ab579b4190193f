#!/usr/bin/env python3
"""The workloads whose kernels are profiled for profiles/ (tools/make_profiles.sh runs each under rocprofv3):
  cfg3slab   one rank's full-size share of config 3: 2,048 x 5 Mbp, k = 16, balanced slab 0 of 8 (counted by the bucketed
             sort of bucket_count.hip); 10 scans of the 6-GB matrix (well beyond the 256-MiB Infinity Cache)
  moments    the f64 scans on a device-generated 16 M x 1024 matrix: Welch t, GSC-weighted chi2, weighted Welch t
  fastq      one config-5 sample: 2 M x 150-bp reads, 0.63 GB of FASTQ, framed and counted
  fastqwrapped r06: that sample with its lines wrapped at 60 columns: the device's scan of line kinds against the host's state machine
  fastqgz    r05: N (default 16) config-5 samples as .fastq.gz files: counted from plain files, from the .gz files (inflated on the
             device), the inflate alone, zlib on one host thread beside it
  cfg5gz     r05: `phenotypeseeker modeling` on N (default 64) config-5 read sets as .fastq.gz files, with the device inflate and with
             the r04 route (a Python thread pool inflates)
  ingest     config 2's ingest alone: 256 x 5 Mbp, k = 13, counted three times + presence build
  solver     the L1 (grid value, fold) fits of three recorded runs (143 fits each)
  predict    `prediction` counting (f1): 256 x 5 Mbp against a 1,000-word model dictionary, and a 5,000-word one (global table)
  weights    the -w side path (f2) at 1,024 samples: MinHash sketches beside the counting (hash filter + select), the
             523,776 pair merges of mash_pairs_kernel, neighbour joining of the 1,024 leaves
  solver4096 the same grid on a 4,096 x 1,000 design (the 64-word instance of the L1 kernel)
  lasso      the bit-packed Lasso grid (13 alphas x 10 folds + refits) of a 1,024-sample continuous run
Each prints one JSON line: the algorithmic bytes per launch of its kernels (what `frac` in profiles/ is computed from).
usage: tools/profile_workloads.py NAME"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.engine import PskContext  # noqa: E402
from phenotypeseeker_amd.synth import GenomeSet  # noqa: E402

what = sys.argv[1]
out = {"workload": what, "algorithmic_bytes_per_launch": {}, "notes": {}}
alg = out["algorithmic_bytes_per_launch"]

if what == "cfg3slab":
    from phenotypeseeker_amd import dist
    n, L, k, world = 2048, 5_000_000, (int(sys.argv[2]) if len(sys.argv) > 2 else 16), 8     # (r05: `cfg3slab 21` / `cfg3slab 31`: the same slab at a 64-bit word length)
    gs = GenomeSet(n, L, seed=12345)
    with PskContext(0) as ctx:
        ctx.begin(k, 1)
        nu0, _ = ctx.count_kmers(0, gs.sample(0)[1])
        bounds = dist.quantile_bounds(dist.pilot_points(ctx.get_list(0, nu0)[0]), k, world)
        ctx.begin(k, n, bounds[0], bounds[1])
        t0 = time.time()
        pairs = 0
        for s0 in range(0, n, 64):
            nu, _ = ctx.count_kmers_batch(s0, [gs.sample(i)[1] for i in range(s0, min(s0 + 64, n))], 8)
            pairs += sum(nu)
        t1 = time.time()
        m = ctx.build_presence()
        t2 = time.time()
        ph = np.array([gs.phenotype(i) for i in range(n)], dtype=np.int8)
        ctx.chi2_scan(ph, None, 2, n - 2, 0.05, False, 8 * m)
        ms = ctx.rescan_timed(10)
        out["notes"] = {"rows": m, "pairs": pairs, "generate_and_count_s": round(t1 - t0, 2), "presence_s": round(t2 - t1, 4), "k": k,
                        "scan_ms": ms, "matrix_GB": m * 256 / 1e9}
        alg["chi2_scan_kernel"] = m * 256
        kept = pairs // n                       # words of a sample inside the slab (its unique words: a genome has few repeats)
        wb = 4 if k <= 16 else 8                # bytes of a partitioned word (bucket_count.hip: 32-bit words up to k = 16)
        alg["bs_hist_kernel"] = L
        alg["bs_partition_kernel"] = L + wb * kept
        alg["bs_sort_kernel"] = wb * kept + (wb + 4) * kept
        alg["bs_compact_kernel"] = (wb + 4) * kept + 12 * kept
        for name in ("bs_hist", "bs_partition", "bs_sort", "bs_compact"):   # a launch of the grouped chain covers eight genomes
            alg[name + "_batch_kernel"] = 8 * alg[name + "_kernel"]
        # presence build, streaming merge (12 B x pairs read + the matrix written; the merge reads the 8-byte
        # words only, and twice -- once per pass): pm_mark = 8 B x pairs + the occupancy bitmap, pm_fill = 8 B x pairs +
        # the matrix + the union words
        alg["pm_mark_kernel"] = 8 * pairs
        alg["pm_fill_kernel"] = 8 * pairs + m * 256 + m * 8
        # r04: pass 2 replays the records of pass 1 -- 12 B per (word, ballot) record read, one 8-byte word per record written
        # into the zeroed matrix; the records of this slab: one per merge iteration, ~ pairs / 20 (the trace line says how many
        # chunks of 64); the union kernel reads the bitmap + ranks and writes the words
        recs = pairs // 20
        alg["pm_replay_kernel"] = 12 * recs + 8 * recs
        alg["pm_union_kernel"] = (bounds[1] - bounds[0]) // 64 * 12 + m * 8
elif what == "kwide":
    # r05: counting and presence build beyond the 32-bit word spaces (k = 17..32: `-l` accepts them, scripts/phenotypeseeker:89-92) next
    # to k = 16 -- a rank's 1/8 slab of 256 x 5 Mbp (the kernels per genome under the slab filter), then the whole space of 64 genomes
    from phenotypeseeker_amd import dist
    k = int(sys.argv[2])
    n, L, world = 256, 5_000_000, 8
    gs = GenomeSet(n, L, seed=12345)
    fas = [gs.sample(i)[1] for i in range(n)]
    with PskContext(0) as ctx:
        ctx.begin(k, 1)
        nu0, _ = ctx.count_kmers(0, fas[0])
        bounds = dist.quantile_bounds(dist.pilot_points(ctx.get_list(0, nu0)[0]), k, world)
        res = {}
        modes = sys.argv[3:] or ["slab", "whole"]
        for label, lo, hi, m_ in (("slab", bounds[0], bounds[1], n), ("whole", 0, 0, 64)):
            if label not in modes:
                continue
            ctx.begin(k, m_, lo, hi)
            t0 = time.time()
            pairs = 0
            for s0 in range(0, m_, 64):
                nu, _ = ctx.count_kmers_batch(s0, fas[s0:s0 + 64], 8)
                pairs += sum(nu)
            t1 = time.time()
            m = ctx.build_presence()
            t2 = time.time()
            res[label] = {"genomes": m_, "pairs": pairs, "rows": m, "count_wall_s": round(t1 - t0, 4), "count_wall_us_per_genome": round(1e6 * (t1 - t0) / m_, 1),
                          "presence_s": round(t2 - t1, 4)}
        out["notes"] = {"k": k, **res}
elif what == "moments":
    M, N = 16_000_000, 1024
    rng = np.random.default_rng(3)
    with PskContext(0) as ctx:
        ctx.begin(16, N)
        ctx.synth_presence(M, N, 7)
        ph = (np.arange(N) % 2).astype(np.int8)
        vals = rng.normal(0, 1, N) + ph * 0.3
        w = rng.uniform(0.5, 1.5, N)
        ones = np.ones(N, np.uint8)
        t, kept = {}, {}
        for rep in range(3):
            kept["chi2"] = ctx.chi2_scan(ph, None, 2, N - 2, 0.05, False, M); t["chi2"] = ctx.last_scan_ms()
            kept["chi2_weighted"] = ctx.chi2_scan(ph, w, 2, N - 2, 0.05, False, M); t["chi2_weighted"] = ctx.last_scan_ms()
            kept["ttest"] = ctx.ttest_scan(vals, ones, None, 2, N - 2, 0.05, M); t["ttest"] = ctx.last_scan_ms()
            kept["ttest_weighted"] = ctx.ttest_scan(vals, ones, w, 2, N - 2, 0.05, M); t["ttest_weighted"] = ctx.last_scan_ms()
        out["notes"] = {"rows": M, "samples": N, "event_ms": t, "rows_kept": {k_: int(v) for k_, v in kept.items()}}
        for name in ("chi2_scan_kernel", "ttest_scan_kernel"):
            alg[name] = M * (N // 8)
elif what == "fastq":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_configs import _fastq_sample
    gs = GenomeSet(1, 5_000_000, seed=99)
    data = _fastq_sample(gs.codes(0), 2_000_000, 150, seed=[5, 0])
    with PskContext(0) as ctx:
        ctx.begin(13, 1)
        ts = []
        for rep in range(4):
            t0 = time.time()
            nu, nt = ctx.count_kmers_batch(0, [data], 8)
            ts.append(round(time.time() - t0, 4))
        tb = []
        for rep in range(3):            # the same sample four times in one call: the copies into pinned memory run beside the uploads
            t0 = time.time()
            ctx.begin(13, 4)
            ctx.count_kmers_batch(0, [data] * 4, 8)
            tb.append(round((time.time() - t0) / 4, 4))
        out["notes"] = {"file_bytes": len(data), "windows": int(nt[0]), "unique": int(nu[0]), "wall_s": ts,
                        "GBps_of_file_bytes": round(len(data) / min(ts) / 1e9, 1), "batch_of_4_wall_s_per_sample": tb,
                        "batch_GBps_of_file_bytes": round(len(data) / min(tb) / 1e9, 1)}
        clean = 2_000_000 * 151
        for name in ("fq_lines_kernel", "fq_pass_kernel"):
            alg[name] = len(data)
        alg["dc_hist_kernel"] = clean
        alg["dc_partition_kernel"] = clean + 2 * int(nt[0])
        alg["dc_count_kernel"] = 2 * int(nt[0]) + (1 << 26) // 8
elif what == "fastqwrapped":
    # r06: the same config-5 sample with its sequence and quality lines wrapped at 60 columns -- FASTQ that is not four lines a record:
    # framed on the device by the scan of line kinds (frame_gpu.hip, format 3) against r05's route (PSK_HOST_WRAPPED_FASTQ=1: the host's
    # state machine) and against the four-line file; the lists are equal by construction of the file only where the machine's quirk (the
    # first byte of a continued sequence line is swallowed) allows -- so the wrapped file is compared with ITSELF on the two routes
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_configs import _fastq_sample
    gs = GenomeSet(1, 5_000_000, seed=99)
    data = _fastq_sample(gs.codes(0), 2_000_000, 150, seed=[5, 0])
    lines = data.split(b"\n")
    wrapped = bytearray()
    for q in range(0, len(lines) - 3, 4):
        h, sq, pl, ql = lines[q:q + 4]
        wrapped += h + b"\n" + sq[:60] + b"\n" + sq[60:120] + b"\n" + sq[120:] + b"\n" + pl + b"\n" + ql[:60] + b"\n" + ql[60:120] + b"\n" + ql[120:] + b"\n"
    wrapped = bytes(wrapped)
    res = {}
    with PskContext(0) as ctx:
        for label, payload, env in (("four_line", data, None), ("wrapped_device", wrapped, None), ("wrapped_host", wrapped, "PSK_HOST_WRAPPED_FASTQ")):
            if env:
                os.environ[env] = "1"
            ts = []
            for rep in range(4):
                ctx.begin(13, 1)
                t0 = time.time()
                nu, nt = ctx.count_kmers_batch(0, [payload], 8)
                ts.append(round(time.time() - t0, 4))
            w, f = ctx.get_list(0, nu[0])
            if env:
                os.environ.pop(env)
            res[label] = {"file_bytes": len(payload), "wall_s": ts, "GBps_of_file_bytes": round(len(payload) / min(ts) / 1e9, 1), "unique": int(nu[0]),
                          "windows": int(nt[0]), "list_sha": __import__("hashlib").sha256(w.tobytes() + f.tobytes()).hexdigest()[:16]}
    res["wrapped_routes_agree"] = res["wrapped_device"]["list_sha"] == res["wrapped_host"]["list_sha"]
    out["notes"] = res
    for name in ("fqg_pass_kernel",):
        alg[name] = len(wrapped)
elif what == "fastqgz":
    # r05: config-5 samples as sequencers ship them -- .fastq.gz.  N samples of 2 M x 150-bp reads (0.63 GB of text each,
    # qualities in runs as Illumina bins them), gzip level 6, counted from files: plain, then compressed (the images cross
    # PCIe and are inflated on the device: csrc/gz_inflate.hip), with zlib on one host thread beside it
    import gzip
    import shutil
    import tempfile
    import zlib
    from concurrent.futures import ProcessPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gz_bench import make_fastq_like
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    tmp = tempfile.mkdtemp(prefix="psk_fastqgz_")
    try:
        t0 = time.time()
        with ProcessPoolExecutor(min(n, os.cpu_count() or 4)) as ex:
            made = list(ex.map(make_fastq_like, [(tmp, i, 2_000_000) for i in range(n)]))
        t_make = time.time() - t0
        plain, packed = [m[0] for m in made], [m[1] for m in made]
        text_bytes, comp_bytes = sum(m[2] for m in made), sum(m[3] for m in made)
        t0 = time.time()
        with open(packed[0], "rb") as f:
            one = zlib.decompress(f.read(), 31)
        t_zlib = time.time() - t0
        with PskContext(0) as ctx:
            res = {}
            for name, paths in (("plain", plain), ("gz", packed), ("plain_again", plain), ("gz_again", packed)):
                ctx.begin(13, n)
                t0 = time.time()
                nu, nt = ctx.count_kmers_files(0, paths, 8)
                res[name] = (round(time.time() - t0, 3), list(nu), list(nt))
            assert res["plain"][1:] == res["gz"][1:] == res["gz_again"][1:], "the lists of the .gz samples differ"
            images = [open(p, "rb").read() for p in packed]
            _, _, routes, ms = ctx.gz_inflate(images, want_text=False)     # (the first call of this size allocates its buffers: ~30 ms per GB)
            ms = min(ms, ctx.gz_inflate(images, want_text=False)[3])
        out["notes"] = {"samples": n, "text_bytes": text_bytes, "gz_bytes": comp_bytes, "made_in_s": round(t_make, 1),
                        "count_from_plain_files_s": [res["plain"][0], res["plain_again"][0]],
                        "count_from_gz_files_s": [res["gz"][0], res["gz_again"][0]],
                        "inflate_alone_ms": round(ms, 1), "inflate_alone_GBps_of_text": round(text_bytes / ms / 1e6, 1),
                        "routes": sorted(set(routes)),
                        "zlib_one_host_thread_GBps_of_text": round(len(one) / t_zlib / 1e9, 3)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
elif what == "cfg5gz":
    # r05: `phenotypeseeker modeling` end to end on read sets as sequencers ship them: N (default 64) samples of 2 M x 150-bp reads
    # (60 x coverage of related 5-Mbp genomes with a phenotype: synth.GenomeSet) as .fastq.gz files, k = 13 -- with the device
    # inflate, and with PSK_NO_GPU_GZ=1 (the r04 route: a thread pool of this process inflates, the text crosses PCIe)
    import shutil
    import tempfile
    from concurrent.futures import ProcessPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gz_bench import make_fastq_like
    from phenotypeseeker_amd.cli import build_parser
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    reads = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
    tmp = tempfile.mkdtemp(prefix="psk_cfg5gz_")
    cwd = os.getcwd()
    try:
        gs_par = (n, 5_000_000, 4242)
        gs = GenomeSet(*gs_par)
        t0 = time.time()
        with ProcessPoolExecutor(min(n, os.cpu_count() or 4)) as ex:
            made = list(ex.map(make_fastq_like, [(tmp, i, reads, gs_par) for i in range(n)]))
        t_make = time.time() - t0
        with open(os.path.join(tmp, "data.pheno"), "w") as f:
            f.write("ID\tAddresses\tPheno\n" + "".join("%s\t%s\t%d\n" % (gs.name(i), os.path.basename(made[i][1]), gs.phenotype(i)) for i in range(n)))
        walls = {}
        # r06: `cfg5gz N READS 12288,6144,4096` = the device route at these run sizes (PSK_GZ_GROUP_MB) instead of the three routes
        groups = [g for g in (sys.argv[4].split(",") if len(sys.argv) > 4 else []) if g]
        for route in (["device_g%s_%d" % (g, q) for q, g in enumerate(groups)] if groups else ["device", "device_again", "r04_python_pool"]):
            d = os.path.join(tmp, route)
            os.mkdir(d)
            for m in made:
                os.symlink(m[1], os.path.join(d, os.path.basename(m[1])))
            shutil.copy(os.path.join(tmp, "data.pheno"), d)
            os.chdir(d)
            if route == "r04_python_pool":
                os.environ["PSK_NO_GPU_GZ"] = "1"
            if route.startswith("device_g"):
                os.environ["PSK_GZ_GROUP_MB"] = route[len("device_g"):].split("_")[0]
                time.sleep(8)      # (the driver clears what the run before held: hipMalloc waits for that)
            args = build_parser().parse_args(["modeling", "data.pheno", "-l", "13", "--num_threads", "16"])
            t0 = time.time()
            try:
                args.func(args)
            except SystemExit:      # (no k-mer passes the test on a toy set: the CLI says so and leaves)
                pass
            walls[route] = round(time.time() - t0, 2)
            os.environ.pop("PSK_NO_GPU_GZ", None)
            os.environ.pop("PSK_GZ_GROUP_MB", None)
            try:
                with open("phases_rank0.json") as f:
                    walls[route + "_phases"] = {k_: round(v, 3) for k_, v in json.load(f)["phases_s"].items() if v >= 0.01}
            except (OSError, ValueError, KeyError):
                pass
            walls[route + "_made"] = sorted(x for x in os.listdir(".") if x.endswith(".pkl"))
            os.chdir(cwd)
        def table(route):
            try:
                with open(os.path.join(tmp, route, "chi2_results_Pheno.tsv"), "rb") as f:
                    return f.read()
            except OSError:
                return None
        same = (table("device") is not None and table("device") == table("r04_python_pool")) if not groups else \
            all(table("device_g%s_%d" % (g, q)) is not None and table("device_g%s_%d" % (g, q)) == table("device_g%s_0" % groups[0]) for q, g in enumerate(groups))
        out["notes"] = {"samples": n, "reads_per_sample": reads, "text_bytes": sum(m[2] for m in made), "gz_bytes": sum(m[3] for m in made),
                        "made_in_s": round(t_make, 1), "modeling_wall_s": walls, "chi2_tables_identical": same}
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
elif what == "ingest":
    n, L, k = 256, 5_000_000, 13
    gs = GenomeSet(n, L, seed=12345)
    fas = [gs.sample(i)[1] for i in range(n)]
    with PskContext(0) as ctx:
        ts = []
        for rnd in range(3):
            ctx.begin(k, n)
            t0 = time.time()
            tot = 0
            for lo in range(0, n, 64):
                nu, _ = ctx.count_kmers_batch(lo, fas[lo:lo + 64], 8)
                tot += sum(nu)
            t1 = time.time()
            m = ctx.build_presence()
            ts.append((round(t1 - t0, 4), round(time.time() - t1, 4)))
        out["notes"] = {"count_s, presence_s per round": ts, "rows": m, "pairs": tot}
        raw = len(fas[0])
        alg["fa_summary_kernel"] = raw
        alg["fa_emit_kernel"] = raw + L
        alg["dc_hist_kernel"] = L
        alg["dc_partition_kernel"] = L + 2 * L
        alg["dc_count_sparse_kernel"] = 2 * L + (1 << 26) // 8
        for name in ("dc_hist", "dc_partition", "dc_count_sparse"):   # a launch of the grouped chain covers eight genomes
            alg[name + "_batch_kernel"] = 8 * alg[name + "_kernel"]
        alg["pd_or_kernel"] = n * ((1 << 26) // 8)
        alg["pd_transpose_kernel"] = n * ((1 << 26) // 8) + m * 32 + m * 8
elif what == "solver":
    # the (grid value, fold) fits of two recorded runs: 256 samples x 138 distinct columns, 2048 x 169
    with PskContext(0) as ctx:
        for tag in ("fit256", "fit2048"):
            z = np.load(os.path.join(ROOT, "tools", "data", tag + ".npz"))
            X, y = z["X"].astype(np.float32), z["y"].astype(np.int32)
            ts = []
            for rep in range(2):
                t0 = time.time()
                ctx.logreg_l1_fit(X, y, z["fold"].astype(np.int32), z["fit_param"], z["fit_fold"].astype(np.int32), float(z["tol"]),
                                  int(z["max_iter"]))
                ts.append(round(time.time() - t0, 4))
            out["notes"][tag] = {"X": list(X.shape), "fits": int(len(z["fit_param"])), "wall_s": ts}
        # the grid of a 2048-genome run whose 1000 selected k-mers have 907 distinct patterns (register form of the descent)
        d = np.load(os.path.join(ROOT, "tests", "golden", "fit2048_907.npz"))
        X = np.unpackbits(d["Xbits"], axis=1)[:, : int(d["p"])].astype(np.float32)
        t0 = time.time()
        ctx.logreg_l1_fit(X, d["y"], d["fold"], d["fit_param"], d["fit_fold"], float(d["tol"]), int(d["max_iter"]))
        out["notes"]["fit2048_907"] = {"X": list(X.shape), "fits": int(len(d["fit_param"])), "wall_s": [round(time.time() - t0, 3)]}
elif what == "solver4096":
    # the 33..64-word instance of the L1 kernel (2,049 .. 4,096 samples).  A 4,096 x 1,000 design grown from the
    # recorded 2,048 x 907 one: every genome once more with 2 % of its k-mer calls flipped, 93 further columns = existing ones
    # with 1 % flips, labels of the second half with 5 % noise; the reference's grid (13 C x 10 folds + 13 refits = 143 fits)
    d = np.load(os.path.join(ROOT, "tests", "golden", "fit2048_907.npz"))
    X0 = np.unpackbits(d["Xbits"], axis=1)[:, : int(d["p"])].astype(bool)
    rng = np.random.default_rng(4096)
    X1 = np.vstack([X0, X0 ^ (rng.random(X0.shape) < 0.02)])
    extra = X1[:, rng.integers(0, X1.shape[1], 93)] ^ (rng.random((4096, 93)) < 0.01)
    X = np.hstack([X1, extra]).astype(np.float32)
    y0 = d["y"].astype(np.int32)
    y = np.concatenate([y0, y0 ^ (rng.random(2048) < 0.05)]).astype(np.int32)
    fold = np.concatenate([d["fold"], d["fold"]]).astype(np.int32)
    with PskContext(0) as ctx:
        ctx.logreg_l1_fit(X[:, :50], y, fold, d["fit_param"][:2], d["fit_fold"][:2], 1e-4, 50)
        ts, its = [], None
        for rep in range(2):
            t0 = time.time()
            coef, icpt, its = ctx.logreg_l1_fit(X, y, fold, d["fit_param"], d["fit_fold"], float(d["tol"]), int(d["max_iter"]))
            ts.append(round(time.time() - t0, 3))
        out["notes"] = {"X": list(X.shape), "fits": int(len(d["fit_param"])), "wall_s": ts, "newton_max": int(its.max()),
                        "newton_mean": round(float(its.mean()), 1), "nnz_mean": round(float((coef != 0).sum(axis=1).mean()), 1)}
elif what == "predict":
    n, L, k = 256, 5_000_000, 13
    gs = GenomeSet(n, L, seed=12345)
    fas = [gs.sample(i)[1] for i in range(n)]
    with PskContext(0) as ctx:
        ctx.begin(k, 1)
        nu, _ = ctx.count_kmers(0, fas[0])
        words = ctx.get_list(0, nu)[0]
        ts = {}
        for nd in (1000, 5000):
            d = words[:: max(1, len(words) // nd)][:nd]
            for rep in range(2):
                t0 = time.time()
                for lo in range(0, n, 64):
                    ctx.count_dict_batch(fas[lo:lo + 64], k, d, 8)
                ts[nd] = round(time.time() - t0, 4)
        out["notes"] = {"samples": n, "wall_s_by_dictionary_size": ts}
        alg["dict_count_kernel"] = L                    # 1 B per base: the clean stream is read once, the table sits in LDS / L2
elif what == "weights":
    n, L = 1024, 5_000_000
    gs = GenomeSet(n, L, seed=4242)
    from phenotypeseeker_amd import weights as Wt
    with PskContext(0) as ctx:
        ctx.begin(13, n)
        sketches, t0 = [], time.time()
        for lo in range(0, n, 64):
            _, _, sk = ctx.count_kmers_batch(lo, [gs.sample(i)[1] for i in range(lo, lo + 64)], 8, sketch=(21, 1000, 42))
            sketches += sk
        t1 = time.time()
        common, denom = ctx.mash_pairs(sketches, 1000)
        t2 = time.time()
        dist = Wt.distances_from_counts(common, denom, 21)
        t3 = time.time()
        ctx.nj_merges(dist)
        t4 = time.time()
        out["notes"] = {"samples": n, "generate_count_sketch_s": round(t1 - t0, 2), "mash_pairs_s": round(t2 - t1, 4),
                        "distances_host_s": round(t3 - t2, 4), "nj_s": round(t4 - t3, 4)}
        alg["kmer_hash_filter_kernel"] = 8 * L        # one 8-byte window word per base position (k = 21 extract output) read once
        alg["mash_pairs_kernel"] = n * (n + 1) // 2 * 2 * 8 * 1000      # two sketches of 1,000 hashes per pair (L2-resident: a latency bound, not HBM)
        alg["nj_kernel"] = sum(m * m // 2 * 8 for m in range(3, n + 1))   # the lower triangle once per join (L2-resident)
        alg["nj_grid_kernel"] = alg["nj_kernel"]
        alg["nj_lds_kernel"] = n * n * 8                                    # r04: the matrix read ONCE from memory; every join works in LDS
elif what == "lasso":
    d = np.load(os.path.join(ROOT, "tests", "golden", "fit2048_907.npz"))
    X = np.unpackbits(d["Xbits"], axis=1)[:1024, : int(d["p"])].astype(np.float32)
    rng = np.random.default_rng(5)
    y = 2.0 * X[:, 3] - 1.5 * X[:, 40] + rng.normal(0, 0.5, 1024)
    fold = (np.arange(1024) * 10 // 1024).astype(np.int32)
    alphas = np.logspace(-3, 3, 13)
    fp = np.concatenate([np.repeat(alphas, 10), alphas])
    ff = np.concatenate([np.tile(np.arange(10), 13), np.full(13, -1)]).astype(np.int32)
    with PskContext(0) as ctx:
        ts = []
        for rep in range(2):
            t0 = time.time()
            ctx.lasso_fit(X, y, fold, fp, ff, 1e-4, 1000)
            ts.append(round(time.time() - t0, 4))
        out["notes"] = {"X": list(X.shape), "fits": int(len(fp)), "wall_s": ts}
else:
    raise SystemExit("unknown workload " + what)
print(json.dumps(out), flush=True)
