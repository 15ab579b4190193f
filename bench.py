#!/usr/bin/env python3
"""bench.py -- the north-star metric on MI355X: k-mer x sample chi-squared cells per second.

  python bench.py [--gpus N --steps K --warmup W]
      N = 1: plain process.  N > 1 with no WORLD_SIZE in the environment: this process -- which makes no GPU call --
      starts one child per GPU itself (phenotypeseeker_amd/launch.py: RANK / LOCAL_RANK / WORLD_SIZE, a private
      rendezvous directory), relays rank 0's JSON line as the last line of stdout and exits with the worst child's
      code; the reference's own parallel axis needs no launcher either (Pool(num_threads), modeling.py:1649-1663).
  <any launcher that exports RANK / LOCAL_RANK / WORLD_SIZE> bench.py --gpus N ...   (one rank per GPU; the driver's
      launcher does: only those variables are read -- this program never imports torch, the collectives are RCCL
      calls made by libpsk.so, phenotypeseeker_amd/dist.py)
  With every rank on its own GPU the collectives are RCCL or the run FAILS (rc != 0): the host-file transport is an
  opt-in of the one-GPU test boxes (--share-gpu), and a line measured over it says "scaling": "invalid ...".

N = 1 (BASELINE.json configs[1]): synthetic 256 x 5-Mbp FASTA, binary phenotype, k = 13.  The genome set is
generated on the host, every sample is counted on the GPU (psk_count_kmers_batch) and the union + bit-packed
presence matrix is built on the GPU (psk_build_presence) BEFORE the timed region; the matrix is then resident
in HBM.  One "step" = one pass of the chi-squared scan + filter over the whole resident matrix (kernel,
survivor count read-back).  value = rows x samples x steps / wall time.

N > 1 (BASELINE.json configs[2]): ONE synthetic 2,048 x 5-Mbp set, k = 16 -- the same genomes on every rank --
with the canonical word space range-sharded over the N ranks at the quantiles of a pilot of the lists
(dist.balanced_bounds), every rank keeping the words of its slab (--ingest filter: each rank tokenises every
sample with the slab filter, no data-path collective; --ingest exchange: each sample is counted on one rank and
the slab ranges of the lists are exchanged with one all-to-all over xGMI).  The union sizes are all-reduced once
(the global Bonferroni denominator); a step = the scan of the rank's slab + the export and the all-gather of its
survivors.  value = (global rows) x samples x steps / max-over-ranks wall time: the whole job.  The dataset is the
same at every N > 1 (strong scaling over 2 / 4 / 8); the line carries the rows of every rank and max / mean.

Printed JSON also carries "roofline" (dominant kernel = chi2_scan_kernel, HBM-bound; achieved =
algorithmic bytes / mean HIP-event duration of the kernel over the timed steps) and
"cpu_baseline" (the C oracle's scan on the same rows, one row chunk per host thread, a bounded number of passes).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured float4 copy)


def scan_kernel_hash():
    """sha256[:16] of the scan kernel's source: a committed traffic figure is only quoted for the kernel it was
    measured on."""
    import hashlib
    with open(os.path.join(ROOT, "phenotypeseeker_amd", "csrc", "assoc_scan.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def measured_traffic(rows, wpr):
    """HBM bytes per launch of the scan kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate runs, corrected as MI355X_MICROARCH.md prescribes) committed under
    profiles/ -- counters cannot be read from inside this process.  None when no committed
    profile matches this workload's matrix shape, or when the profile was taken on another version of the kernel
    (its "kernel_source_sha16" differs from the source's hash)."""
    import glob
    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_chi2_scan.json"))):
        try:
            with open(fn) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("rows") == rows and d.get("words_per_row_stored") == wpr and d.get("kernel_source_sha16") == scan_kernel_hash():
            best = d.get("hbm_bytes_per_launch")
    return best


def host_cpus():
    """CPUs this process may really use: the affinity mask, cut to the cgroup's CPU quota when there is one (a
    container that sees 256 hardware threads but is allowed 16 CPUs' worth of time runs 16 threads, not 256)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota|max> <period>"
            q, period = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = int(f.read()), int(g.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.999)))
    return max(1, min(n, 512))


def _drop_from_page_cache(path):
    """fsync + POSIX_FADV_DONTNEED: the file's pages leave the page cache, the next reader gets them from the device."""
    fd = os.open(path, os.O_RDONLY)
    try:
        os.fsync(fd)
        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
    finally:
        os.close(fd)


def _read_phases(d, world):
    """The per-rank phase tables `phenotypeseeker modeling` leaves in its working directory (modeling.Phases): where the
    wall-clock of the end-to-end leg went -- process start, HIP start-up, rendezvous, ingest, presence, scan, gather, model,
    teardown -- per rank: {"rank<r>": {"total_s": ..., "phases_s": {...}}}."""
    out = {}
    for r in range(world):
        try:
            with open(os.path.join(d, "phases_rank%d.json" % r)) as f:
                rec = json.load(f)
            out["rank%d" % r] = {"total_s": rec["total_s"], "phases_s": rec["phases_s"]}
        except (OSError, ValueError, KeyError):
            out["rank%d" % r] = None
    return out


def e2e_modeling(gs, n, k, cold=False):
    """BASELINE.json's second figure: wall-clock of `phenotypeseeker modeling data.pheno` from FASTA files on
    disk to the written .pkl, on the same synthetic genomes, through the CLI entry point in this process
    (its own context on the same GPU).  Reported beside `value`, never part of it.  The FASTA files were written by this
    process a moment earlier: the run reads them from the PAGE CACHE, not from a device -- the record says so
    (`input_files`); `--e2e-cold` drops them from the cache first (fsync + posix_fadvise(DONTNEED)) and says that."""
    import shutil
    import tempfile
    from phenotypeseeker_amd.cli import build_parser
    tmp = tempfile.mkdtemp(prefix="psk_bench_e2e_")
    cwd = os.getcwd()
    try:
        rows = ["ID\tAddresses\tPheno"]
        t0 = time.time()
        for i in range(n):
            name, fa = gs.sample(i)
            with open(os.path.join(tmp, name + ".fasta"), "wb") as f:
                f.write(fa)
            rows.append("%s\t%s.fasta\t%d" % (name, name, gs.phenotype(i)))
        with open(os.path.join(tmp, "data.pheno"), "w") as f:
            f.write("\n".join(rows) + "\n")
        t_write = time.time() - t0
        if cold:
            for i in range(n):
                _drop_from_page_cache(os.path.join(tmp, gs.name(i) + ".fasta"))
        os.chdir(tmp)
        args = build_parser().parse_args(["modeling", "data.pheno", "-l", str(k)])
        err = sys.stderr
        sys.stderr = open(os.devnull, "w")
        try:
            t0 = time.time()
            args.func(args)
            wall = time.time() - t0
        finally:
            sys.stderr.close()
            sys.stderr = err
        made = sorted(f for f in os.listdir(".") if f.endswith(".pkl"))
        return {"modeling_wall_s": round(wall, 3), "what": "phenotypeseeker modeling data.pheno: %d FASTA files on disk -> %s"
                % (n, ", ".join(made) or "no model"), "write_dataset_s": round(t_write, 2), "phases": _read_phases(".", 1),
                "input_files": "dropped from the page cache before the run (fsync + posix_fadvise DONTNEED)" if cold else
                               "page cache (written by this process a moment earlier; --e2e-cold drops them first)"}
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


def e2e_modeling_sharded(grp, gs, n, k, args):
    """The same figure with several ranks: rank 0 writes the FASTA files once, then EVERY rank starts
    `phenotypeseeker modeling data.pheno` as a child process (RANK / LOCAL_RANK / WORLD_SIZE as this launch has them, a
    rendezvous directory of its own) in that directory; the wall-clock is the slowest rank's, process start to exit.  The
    children share the GPUs with this process, which has released its matrix by then.  Any failure is reported, not raised: the
    line's `value` does not depend on this leg."""
    import shutil
    import subprocess
    import tempfile
    rank = grp.rank
    tmp, tmp_made, t_write = "", "", 0.0
    try:
        made_dir, write_err = "", ""
        if rank == 0:
            try:
                made_dir = tempfile.mkdtemp(prefix="psk_bench_e2e_")
                rows = ["ID\tAddresses\tPheno"]
                t0 = time.time()
                for i in range(n):
                    name, fa = gs.sample(i)
                    with open(os.path.join(made_dir, name + ".fasta"), "wb") as f:
                        f.write(fa)
                    rows.append("%s\t%s.fasta\t%d" % (name, name, gs.phenotype(i)))
                with open(os.path.join(made_dir, "data.pheno"), "w") as f:
                    f.write("\n".join(rows) + "\n")
                t_write = time.time() - t0
            except Exception as e:   # noqa: BLE001 -- every rank learns it through the all-gather below
                write_err = "%s: %s" % (type(e).__name__, e)
        tmp = grp.allgather_bytes(("" if write_err else made_dir).encode())[0].decode()
        if rank == 0:
            tmp_made = made_dir
        if not tmp:
            return {"error": "the dataset could not be written" + (": " + write_err if write_err else "")}
        env = dict(os.environ, PSK_RDZV_DIR=os.path.join(tmp, ".rendezvous"),     # a meeting place of their own
                   PSK_REDUNDANT_INGEST="1" if args.ingest == "filter" else "0")
        here = os.path.dirname(os.path.abspath(__file__))
        env["PYTHONPATH"] = here + os.pathsep + env.get("PYTHONPATH", "")
        cmd = [sys.executable, os.path.join(here, "scripts", "phenotypeseeker"), "modeling", "data.pheno", "-l", str(k)]
        # whatever happens to this rank's child, every rank makes the same two collectives below
        t0 = time.time()
        rc, err_tail = 0, ""
        try:
            r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=None if os.environ.get("PSK_TRACE") else subprocess.PIPE,
                               timeout=900)
            rc, err_tail = r.returncode, (r.stderr or b"").decode(errors="replace")[-600:]
        except Exception as e:   # noqa: BLE001 -- a child that hangs is killed by the timeout
            rc, err_tail = -1, "%s: %s" % (type(e).__name__, e)
        wall = time.time() - t0
        worst = grp.allreduce_max(wall)
        failed = grp.allreduce_sum(1 if rc != 0 else 0)
        made = sorted(f for f in os.listdir(tmp) if f.endswith(".pkl")) if rank == 0 else []
        res = {"modeling_wall_s": round(worst, 3), "ranks": grp.world, "ingest": args.ingest,
               "phases": _read_phases(tmp, grp.world) if rank == 0 else None,
               "what": "phenotypeseeker modeling data.pheno as %d child processes (one per rank, started after the files were "
                       "written): %d FASTA files on disk -> %s" % (grp.world, n, ", ".join(made) or "no model"),
               "write_dataset_s": round(t_write, 2)}
        if rank == 0 and not made and not failed:
            res["error"] = "no .pkl written; rank 0: %s" % err_tail
        if failed:
            res["error"] = "%d rank(s) failed; rank %d: %s" % (failed, rank, err_tail[-400:])
        return res
    except Exception as e:   # noqa: BLE001 -- reported in the line
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        try:
            grp.barrier()
        except Exception:   # noqa: BLE001
            pass
        if rank == 0 and tmp_made:
            shutil.rmtree(tmp_made, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--samples", type=int, default=None, help="default 256 (N = 1) / 2048 (N > 1)")
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--kmer", type=int, default=None, help="default 13 (N = 1) / 16 (N > 1)")
    ap.add_argument("--ingest", default="auto", choices=["auto", "filter", "exchange"],
                    help="N > 1: every rank tokenises every sample and keeps its slab (filter), or every sample is "
                         "counted on one rank and the list ranges are exchanged with an all-to-all (exchange); auto = the "
                         "CLI's rule: exchange when the collectives are RCCL, filter on a host transport (DESIGN.md section 8)")
    ap.add_argument("--workload", default="fasta", choices=["fasta", "matrix"],
                    help="fasta: count synthetic genomes on the GPU (default, BASELINE cfg 2); "
                         "matrix: device-generated presence matrix of --rows rows (quick runs)")
    ap.add_argument("--rows", type=int, default=1 << 25)
    ap.add_argument("--cpu-sample-rows", type=int, default=40_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the end-to-end `phenotypeseeker modeling` wall-clock (FASTA files -> .pkl) on the same data")
    ap.add_argument("--e2e-cold", action="store_true",
                    help="drop the FASTA files of the end-to-end leg from the page cache before the run (N = 1)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="map ranks onto the visible GPUs modulo their count (tests on a one-GPU box, with the host "
                         "transport named by PSK_DIST_TRANSPORT; RCCL refuses two ranks on one device)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N = 1 only: run the N > 1 step (scan + export + all-gather) on a one-rank group, to measure "
                         "what the exchange adds per step on one GPU; the line carries \"exchange\": \"forced\"")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launcher-free: fan out here, before anything of this process has touched the GPU (launch.py imports the
        # standard library only; the children are fresh processes, not an exec of a process that has initialised HIP)
        from phenotypeseeker_amd import launch
        sys.exit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus, share_gpu=args.share_gpu))
    from phenotypeseeker_amd import dist as psk_dist
    from phenotypeseeker_amd.engine import PskContext
    from phenotypeseeker_amd.synth import GenomeSet

    from phenotypeseeker_amd import watchdog

    grp = psk_dist.Group()
    if grp.world != args.gpus:
        sys.exit("bench.py --gpus %d but WORLD_SIZE=%d in the environment" % (args.gpus, grp.world))
    # a launch that runs into its deadline (launch.spawn_ranks: PSK_LAUNCH_TIMEOUT) asks every rank where it is: the rank
    # answers SIGUSR1 with phases_rank<r>.json -- the phase and the call it is stuck in
    t_proc = time.time()
    done_phases = {}

    def phase(name, _last=[None, t_proc]):
        if _last[0] is not None:
            done_phases[_last[0]] = round(done_phases.get(_last[0], 0.0) + time.time() - _last[1], 4)
        _last[0], _last[1] = name, time.time()
        watchdog.enter(name)
    watchdog.install(grp.rank, grp.world, lambda: {"total_s": round(time.time() - t_proc, 4), "phases_s": dict(done_phases)})
    if grp.world > 1 and os.environ.get("PSK_LAUNCHER") != "psk":
        # ranks of somebody else's launcher (the driver's torch.distributed.run): nobody above them keeps the deadline, so
        # every rank keeps it itself -- table, then exit 124
        from phenotypeseeker_amd import launch as _launch
        watchdog.self_deadline(_launch.launch_timeout())
    phase("rendezvous, communicator (ncclCommInitRank)")
    if args.share_gpu:
        os.environ["PSK_SHARE_GPU"] = "1"     # ranks modulo the visible GPUs; opts into the host-file transport (tests)
    grp.init(force=args.force_exchange)       # RCCL, or an error: see dist._rccl_or_host_files
    rank, world = grp.rank, grp.world
    sharded = world > 1
    if args.ingest == "auto":
        # what a rank moves at config 3 (2,048 x 5 Mbp, k = 16, 8 ranks): filter = ALL 10.2 GB of FASTA read from the host's
        # files by every rank and sent over its PCIe link (8 x 10 GB of host reads in all); exchange = an eighth of the files
        # per rank (1.3 GB), then 7/8 of 12 B x 1.28 G pairs = 13.4 GB per rank GPU to GPU over seven xGMI links at once.
        # The host side is the scarce one: exchange wherever the collectives are RCCL (as the CLI does)
        args.ingest = "exchange" if getattr(grp, "backend", None) == "rccl" else "filter"
    n = args.samples if args.samples is not None else (2048 if sharded else 256)
    k = args.kmer if args.kmer is not None else (16 if sharded else 13)

    phase("HIP runtime, context")
    ctx = PskContext(grp.device)
    info = ctx.device_info()
    t_setup = time.time()
    phase("ingest: pilot, k-mer lists, list exchange")
    pheno = np.array([1 if i % 2 == 0 else 0 for i in range(n)], dtype=np.int8)
    ingest = {}
    shard = None
    if args.workload == "fasta":
        gs = GenomeSet(n, args.length, seed=12345)        # ONE dataset: the same genomes on every rank
        t_gen = t_cnt = t_xch = 0.0
        call_s = []          # seconds of every 64-sample counting call: the first one carries the process's first-touch costs
        tot_unique = 0
        lo_w = hi_w = 0
        if sharded:
            # pilot: rank r counts sample r whole; the quantiles of these lists cut the word space into slabs of equal
            # row share (dist.balanced_bounds; one all-gather)
            t0 = time.time()
            with PskContext(grp.device) as pc:
                pc.begin(k, 1)
                nu0, _ = pc.count_kmers(0, gs.sample(rank % n)[1])
                pilot = pc.get_list(0, nu0)[0]
            bounds = psk_dist.balanced_bounds(grp, k, [pilot])
            lo_w, hi_w = bounds[rank], bounds[rank + 1]
            t_pilot = time.time() - t0
        ctx.begin(k, n, lo_w, hi_w)
        if sharded and args.ingest == "exchange":
            own = [i for i in range(n) if psk_dist.owner_of(i, world) == rank]
            with PskContext(grp.device) as cnt:
                cnt.begin(k, max(len(own), 1))
                tots = []
                for lo in range(0, len(own), 64):
                    t0 = time.time()
                    fas = [gs.sample(i)[1] for i in own[lo:lo + 64]]
                    t1 = time.time()
                    _, nts = cnt.count_kmers_batch(lo, fas, 8)
                    t2 = time.time()
                    tots += list(nts)
                    t_gen += t1 - t0
                    t_cnt += t2 - t1
                t0 = time.time()
                tot_unique = psk_dist.ListExchange(grp, k, bounds).run(cnt, ctx, n, tots)
                t_xch = time.time() - t0
        else:
            for lo in range(0, n, 64):
                t0 = time.time()
                fas = [gs.sample(i)[1] for i in range(lo, min(lo + 64, n))]
                t1 = time.time()
                nus, _ = ctx.count_kmers_batch(lo, fas, 8)
                t2 = time.time()
                t_gen += t1 - t0
                t_cnt += t2 - t1
                call_s.append(round(t2 - t1, 4))
                tot_unique += sum(nus)
        phase("presence matrix")
        t0 = time.time()
        M = ctx.build_presence()
        t_build = time.time() - t0
        ingest = {"generate_s": round(t_gen, 2), "count_s": round(t_cnt, 3), "presence_s": round(t_build, 3),
                  "pairs": tot_unique, "bases": n * args.length}
        if call_s:
            ingest["count_first_call_s"] = call_s[0]
            ingest["count_steady_call_s"] = round(float(np.median(call_s[1:])), 4) if len(call_s) > 1 else call_s[0]
        if sharded:
            ingest.update({"mode": args.ingest, "pilot_s": round(t_pilot, 2), "exchange_s": round(t_xch, 2)})
        workload = "synthetic %d x %.1f-Mbp FASTA, binary phenotype, k=%d" % (n, args.length / 1e6, k)
        if sharded:
            workload += ", canonical word space range-sharded over %d GPUs" % world
    else:
        M = args.rows
        ctx.synth_presence(M, n, seed=7 + rank)
        workload = "device-generated presence matrix %d rows x %d samples" % (M, n)
    _, wpr, _ = ctx.presence_shape()
    t_setup = time.time() - t_setup

    phase("all-reduce of the union size")
    M_global = grp.allreduce_sum(int(M))
    rows_per_rank = [int(x) for x in grp.allgather_i64(np.array([int(M)]))[:, 0]]

    # The sharded path: the union size is all-reduced ONCE (Bonferroni denominator, above); every scan
    # is followed by one all-gather of its survivors.  The gather is device-to-device (RCCL) and
    # double-buffered, so the collective of scan i runs while scan i+1 streams the matrix -- as it
    # does in a run with several phenotypes.  N = 1 has no exchange.
    xch = psk_dist.SurvivorExchange(grp, wpr) if world > 1 or args.force_exchange else None
    pending = []
    gathered = [0]

    def drain(limit, read_counts=False):
        while len(pending) > limit:
            s_ = pending.pop(0)
            if read_counts or not pending:
                counts = xch.finish_counts(s_)       # gathered table stays on the device
                if (counts > xch.cap).any():
                    raise RuntimeError("survivor buffer overflow in bench")
                gathered[0] = int(counts.sum())
            else:
                xch.wait(s_)                         # completion only: the slot's buffers are reused next step

    scan_args = (pheno, None, 2, n - 2, 0.05, False, M_global)
    # Untimed, before the W warm-up steps: ~40 ms of back-to-back scans so that the GPU's clocks have settled.  The
    # ingest ends with a few light kernels; measured right after it, the first ~50 scans run 4-5 % slower than the
    # steady state (r01: 117-121 us against 111-112 us per launch, the same for a 1000-step run either way).
    phase("clock-settling scans")
    ctx.chi2_scan(*scan_args)          # the first launch also loads the kernel's code object
    warm_ms = ctx.rescan_timed(3)
    ctx.rescan_timed(int(min(300, max(3, 40.0 / max(warm_ms, 0.01)))))

    def run_steps(count):
        """`count` steps; returns (survivors of the last scan, kernel ms of every scan).  Two scans are kept in flight
        (psk_chi2_scan_begin twice, two result sets -- what modeling.py does over the phenotypes of a run), so the
        host's part of a step -- wait for scan i and read its survivor count; N > 1: pack the survivors and queue the
        all-gather -- happens while scan i + 1 streams its matrix.  Every step is one whole scan whose count reaches
        the host inside the timed region, plus, for N > 1, one export and one all-gather."""
        ms_all, npass = [], 0
        for _ in range(min(2, count)):
            ctx.chi2_scan_begin(*scan_args)
        for i in range(count):
            npass = ctx.scan_end()
            ms_all.append(ctx.last_scan_ms())
            s_ = xch.export(ctx) if xch is not None else None
            if i + 2 < count:
                ctx.chi2_scan_begin(*scan_args)
            if xch is not None:
                xch.collect(s_)
                pending.append(s_)
                if len(pending) > 1:
                    xch.wait(pending.pop(0))
        return npass, ms_all

    phase("warm-up steps (scan + survivors' all-gather)")
    npass, _ = run_steps(args.warmup) if args.warmup > 0 else (0, [])
    if xch is not None:
        drain(0)
    grp.barrier()
    phase("timed steps (scan + survivors' all-gather)")
    t0 = time.perf_counter()
    npass, kernel_ms = run_steps(args.steps)
    if xch is not None:
        drain(0)   # the last exchange completes inside the timed region
    grp.barrier()
    elapsed = time.perf_counter() - t0
    phase("reductions of the result line, roofline, CPU baseline, e2e leg")
    elapsed = grp.allreduce_max(elapsed)
    cells_total = grp.allreduce_sum(int(M) * n) * args.steps
    value = cells_total / elapsed

    # roofline of the dominant kernel (this rank): algorithmic bytes = M * 8 * ceil(N/64)
    # (1 bit per cell; the kernel does not read the key array)
    alg_words = (n + 63) // 64
    alg_bytes = M * 8 * alg_words
    mean_ms = float(np.mean(kernel_ms))
    achieved = alg_bytes / (mean_ms * 1e-3) / 1e9
    traffic = measured_traffic(int(M), wpr)
    # the stream-read ceiling measured in this run, on this matrix ("against both the 8 TB/s spec and the measured
    # stream-read ceiling"): a kernel that only reads the matrix, 16 B per lane, as many launches as the timed steps, its clocks
    # settled like the scan's -- outside the timed region
    ctx.stream_read_ceiling(3)
    ceil_ms, ceil_bytes, ceil_shape = ctx.stream_read_ceiling(max(args.steps, 10))
    ceiling = ceil_bytes / (ceil_ms * 1e-3) / 1e9
    stored_rate = int(M) * 8 * wpr / (mean_ms * 1e-3) / 1e9      # bytes of the matrix as stored (= algorithmic unless rows are padded)
    # a multi-rank figure is a scaling point only when its collectives ran on RCCL with one GPU per rank
    real_multi = sharded and grp.backend == "rccl" and grp.rccl_ranks == world and not args.share_gpu
    scaling = "weak" if not sharded else ("strong" if real_multi else
                                          "invalid (%s collectives%s: not a scaling point)" % (grp.backend, ", ranks share GPUs" if args.share_gpu else ""))
    out = {
        "metric": "k-mer x sample chi2 cells/sec", "value": value, "unit": "cells/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "u64 popcount + f64",
        "data": "synthetic",
        "config": {"workload": workload, "n_samples": n, "k": k, "rows_per_gpu": int(M),
                   "words_per_row_stored": wpr, "survivors": int(npass), "survivors_all_slabs": int(gathered[0]) if xch is not None
                   else int(npass), "device": info["name"],
                   "setup_s": round(t_setup, 2), "ingest": ingest,
                   "rows_global": int(M_global), "rows_per_rank": rows_per_rank,
                   "balance_max_over_mean": round(max(rows_per_rank) / (sum(rows_per_rank) / len(rows_per_rank)), 4),
                   "collectives": (grp.backend or "none") +
                   (" (fallback: %s)" % grp.t.fallback_reason[:700] if getattr(grp.t, "fallback_reason", None) else "")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "chi2_scan_kernel",
                     "kernel_ms": mean_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "stored_bytes_per_launch": int(M) * 8 * wpr,
                     "measured_stream_ceiling": {"GBps": ceiling, "kernel": "stream_read_kernel (%s)" % ceil_shape, "kernel_ms": ceil_ms,
                                                 "bytes_per_launch": int(ceil_bytes), "frac_of_peak": ceiling / HBM_PEAK_GBS,
                                                 "what": "the same matrix read once per launch by a kernel that does nothing else (the fastest "
                                                         "of four shapes), timed with HIP events in this run"},
                     "frac_of_measured_ceiling": stored_rate / ceiling},
    }

    if sharded or args.force_exchange:
        out["rccl_ranks"] = grp.rccl_ranks     # ncclCommCount of the communicator the collectives ran on (0: not RCCL)
    if args.force_exchange:
        out["exchange"] = "forced"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle's scan (C restatement of modeling.py:677-858, "port") on the same matrix, checked
        # equal to the GPU's answer first; rows are independent, so they are cut into one range per host thread
        # (orc_chi2_scan_mt, POSIX threads -- the reference runs its chunks in a process pool).  One thread alone is
        # timed too, on a part of the rows.
        from oracle import oracle as O
        ns = int(min(args.cpu_sample_rows, M))
        rows = ctx.get_rows(np.arange(ns, dtype=np.uint64))
        threads = host_cpus()
        ph_list, ones = pheno.tolist(), np.ones(n)
        n1 = min(ns, 4_000_000)
        t0 = time.perf_counter()
        ref1 = O.chi2_scan(rows[:n1], ph_list, ones, n, 2, n - 2, 0.05, False, M_global)
        dt1 = time.perf_counter() - t0
        scratch = {}   # output arrays of the untimed pass, reused by the timed ones
        ref = O.chi2_scan(rows, ph_list, ones, n, 2, n - 2, 0.05, False, M_global, n_threads=threads, scratch=scratch)
        t0 = time.perf_counter()
        reps = 0
        while True:   # bounded sample: repeat the pass until a few seconds of wall-clock have been timed
            O.chi2_scan(rows, ph_list, ones, n, 2, n - 2, 0.05, False, M_global, n_threads=threads, scratch=scratch)
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= 3.0 or reps >= 32:
                break
        res = ctx.get_results(npass)
        sel = res["row"] < ns
        same = bool(np.array_equal(res["row"][sel], np.nonzero(ref["keep"])[0].astype(np.uint64)) and
                    np.allclose(res["stat"][sel], ref["stat"][ref["keep"]], rtol=1e-12) and
                    np.array_equal(ref1["keep"], ref["keep"][:n1]) and np.array_equal(ref1["stat"], ref["stat"][:n1]))
        out["cpu_baseline"] = {"value": reps * ns * n / dt, "unit": "cells/s", "cores": threads, "kind": "port",
                               "sample": "%d pass(es) over the first %d rows of the same matrix (%d samples), one row range "
                                         "per thread, oracle/psk_oracle.c orc_chi2_scan_mt, %.1f s" % (reps, ns, n, dt),
                               "single_thread_value": n1 * n / dt1,
                               "matches_gpu": same,
                               "reference_python_8proc_cells_per_s": 7.4e6}
    if rank == 0 and world == 1 and args.workload == "fasta" and not args.no_e2e:
        out["e2e"] = e2e_modeling(gs, n, k, cold=args.e2e_cold)
    ctx.close()   # the matrix and the lists go before the CLI children of the next leg bring their own
    if world > 1 and args.workload == "fasta" and not args.no_e2e:
        out["e2e"] = e2e_modeling_sharded(grp, gs, n, k, args)
    grp.close()
    if rank == 0:
        # RCCL prints a version banner through C stdio, which (piped) is flushed at exit, after Python's own
        # buffer: flush it now so that the JSON line is the LAST line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
