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
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_chi2_scan*.json"))):
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
            for key in ("stuck_in", "stuck_in_for_s", "blocked_in_call", "blocked_for_s"):   # a rank that was asked where it hangs
                if rec.get(key) is not None:
                    out["rank%d" % r][key] = rec[key]
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


def _run_child_with_budget(cmd, cwd, env, budget_s):
    """One CLI child: (rc, stderr tail).  A child still running after budget_s is first asked where it is (SIGUSR1: it writes
    phases_rank<r>.json with the phase and the call it is stuck in, watchdog.py), then terminated by its pid; rc 124 then."""
    import signal
    import subprocess
    import tempfile
    with tempfile.TemporaryFile() as errf:
        p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.DEVNULL, stderr=None if os.environ.get("PSK_TRACE") else errf)
        rc = None
        try:
            rc = p.wait(timeout=budget_s)
        except subprocess.TimeoutExpired:
            for sig, wait_s in ((signal.SIGUSR1, 2.0), (signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
                try:
                    p.send_signal(sig)
                except OSError:
                    pass
                if sig != signal.SIGUSR1:
                    try:
                        p.wait(timeout=wait_s)
                        break
                    except subprocess.TimeoutExpired:
                        pass
                else:
                    time.sleep(wait_s)
            rc = 124
        finally:
            if p.poll() is None:
                p.kill()
                p.wait()
        errf.seek(0, os.SEEK_END)
        size = errf.tell()
        errf.seek(max(0, size - 600))
        return rc, errf.read().decode(errors="replace")


def e2e_budget_s(t_launch):
    """Seconds the end-to-end leg may take: min(900, what is left of the launch's deadline - 60 s), so that the leg's own
    time-out fires -- and is reported in the line -- before the launch's deadline does."""
    from phenotypeseeker_amd import launch
    deadline = launch.launch_timeout()
    if not deadline or deadline <= 0:
        return 900.0
    return min(900.0, deadline - (time.time() - t_launch) - 60.0)


def e2e_modeling_sharded(grp, gs, n, k, args, budget_s=900.0):
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
            rc, err_tail = _run_child_with_budget(cmd, tmp, env, budget_s)
        except Exception as e:   # noqa: BLE001
            rc, err_tail = -1, "%s: %s" % (type(e).__name__, e)
        wall = time.time() - t0
        worst = grp.allreduce_max(wall)
        failed = grp.allreduce_sum(1 if rc != 0 else 0)
        rcs = [int(x) for x in grp.allgather_i64(np.array([rc]))[:, 0]]
        made = sorted(f for f in os.listdir(tmp) if f.endswith(".pkl")) if rank == 0 else []
        res = {"modeling_wall_s": round(worst, 3), "ranks": grp.world, "ingest": args.ingest, "rc": rcs, "budget_s": round(budget_s, 1),
               "phases": _read_phases(tmp, grp.world) if rank == 0 else None,
               "what": "phenotypeseeker modeling data.pheno as %d child processes (one per rank, started after the files were "
                       "written): %d FASTA files on disk -> %s" % (grp.world, n, ", ".join(made) or "no model"),
               "write_dataset_s": round(t_write, 2)}
        if rank == 0 and not made and not failed:
            res["error"] = "no .pkl written; rank 0: %s" % err_tail
        if failed:
            res["error"] = "%d rank(s) failed (rc per rank %s; 124 = over the leg's budget of %.0f s, < 0 = killed by that signal); " \
                           "rank %d: %s" % (failed, rcs, budget_s, rank, err_tail[-400:])
            if rank == 0:    # the first phase some rank did not get through
                stuck = [(r_, v.get("stuck_in")) for r_, v in ((r_, res["phases"].get("rank%d" % r_) or {}) for r_ in range(grp.world))
                         if v.get("stuck_in")]
                missing = [r_ for r_ in range(grp.world) if res["phases"].get("rank%d" % r_) is None]
                res["first_failing_phase"] = {"stuck": dict(("rank%d" % r_, p_) for r_, p_ in stuck), "no_table_from_ranks": missing}
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
                         "CLI's rule: exchange when the collectives are RCCL, filter on a host transport (DESIGN.md section 7)")
    ap.add_argument("--workload", default="fasta", choices=["fasta", "matrix"],
                    help="fasta: count synthetic genomes on the GPU (default, BASELINE cfg 2); "
                         "matrix: device-generated presence matrix of --rows rows (quick runs)")
    ap.add_argument("--rows", type=int, default=1 << 25)
    ap.add_argument("--cpu-sample-rows", type=int, default=40_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the end-to-end `phenotypeseeker modeling` wall-clock (FASTA files -> .pkl) on the same data")
    ap.add_argument("--no-hbm-only", action="store_true",
                    help="skip the beyond-cache block (the same scan over a device-generated 4.3-GB matrix): what tools/make_profiles.sh "
                         "passes, so that the profiler's average for the headline kernel covers the headline matrix only")
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
    # when the launch began: launch.spawn_ranks says (PSK_LAUNCH_T0); under somebody else's launcher, this process's start
    try:
        t_launch = float(os.environ["PSK_LAUNCH_T0"]) if os.environ.get("PSK_LAUNCHER") == "psk" else t_proc
    except (KeyError, ValueError):
        t_launch = t_proc
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
        ctx.synth_presence(M, n, seed=(80 << 48) | (7 + rank))     # (bits 48..63 = 80: config 2's share of associated rows, so the host picks the headline form of the kernel)
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
    phase("reductions of the result line")
    elapsed = grp.allreduce_max(elapsed)
    cells_total = grp.allreduce_sum(int(M) * n) * args.steps
    value = cells_total / elapsed
    # what every rank spent where before the timed region (one small all-gather): the first thing to look at when an N > 1
    # line disappoints -- which rank's count / exchange / matrix build was the slow one
    mine = dict((key, ingest.get(key)) for key in ("count_s", "exchange_s", "presence_s", "pilot_s", "generate_s"))
    mine["setup_s"] = round(t_setup, 3)
    per_rank = [json.loads(b.decode()) for b in grp.allgather_bytes(json.dumps(mine).encode())] if sharded else [mine]

    # roofline of the dominant kernel (this rank): algorithmic bytes = M * 8 * ceil(N/64)
    # (1 bit per cell; the kernel does not read the key array)
    alg_words = (n + 63) // 64
    alg_bytes = M * 8 * alg_words
    mean_ms = float(np.mean(kernel_ms))
    achieved = alg_bytes / (mean_ms * 1e-3) / 1e9 if mean_ms > 0 else 0.0
    traffic = measured_traffic(int(M), wpr)
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
                   "setup_s": round(t_setup, 2), "ingest": ingest, "ingest_per_rank": per_rank,
                   "rows_global": int(M_global), "rows_per_rank": rows_per_rank,
                   "balance_max_over_mean": round(max(rows_per_rank) / max(sum(rows_per_rank) / len(rows_per_rank), 1e-9), 4),
                   "collectives": (grp.backend or "none") +
                   (" (fallback: %s)" % grp.t.fallback_reason[:700] if getattr(grp.t, "fallback_reason", None) else "")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "chi2_scan_kernel",
                     "kernel_ms": mean_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "stored_bytes_per_launch": int(M) * 8 * wpr},
    }
    if args.workload == "fasta" and n == 2048 and k == 16 and args.length == 5_000_000:
        # BASELINE config 3: the N = 1 run of it is the base of the strong-scaling curve, the N > 1 lines say so
        out["config"]["scaling_base"] = ("this line: the same 2,048 x 5-Mbp, k = 16 dataset on one GPU" if not sharded else
                                         "`bench.py --gpus 1 --samples 2048 --kmer 16` on the same dataset "
                                         "(profiles/r06_cfg3_n1.json); the default N = 1 line is BASELINE config 2, another workload")
    if sharded or args.force_exchange:
        out["rccl_ranks"] = grp.rccl_ranks     # ncclCommCount of the communicator the collectives ran on (0: not RCCL)
    if args.force_exchange:
        out["exchange"] = "forced"

    # `value` and the roofline of the timed steps exist: the line goes out NOW (rank 0, flushed), before any of the legs below
    # can hang or die (an 8-GPU line must not be lost to a leg that does not define it).  Every leg that
    # completes re-emits the augmented line: the LAST JSON line of stdout is the record, and `legs` says what it lacks.
    line = ResultLine(rank, out)
    watchdog.on_dump(line.on_deadline)       # a launch that runs into its deadline re-emits what there is, with the stuck phase
    legs = out["legs"] = {}
    todo = ["kernel_spread", "stream_ceiling"] + ([] if args.no_hbm_only else ["hbm_only"])
    if world == 1 and not args.no_cpu_baseline:
        todo.append("cpu_baseline")
    if args.workload == "fasta" and not args.no_e2e:
        todo.append("e2e")
    for name_ in todo:
        legs[name_] = "pending"
    line.emit()
    rf = out["roofline"]

    def leg(name_, fn):
        """A leg may fail; the line may not.  (Legs that make collectives handle their own failures so that every rank still
        makes them -- e2e_modeling_sharded; the legs before it are this rank's alone.)"""
        if name_ not in legs:
            return
        phase("leg: " + name_)
        try:
            legs[name_] = fn() or "ok"
        except Exception as e:   # noqa: BLE001 -- reported in the line
            legs[name_] = "failed: %s: %s" % (type(e).__name__, str(e)[:300])
        line.emit()

    def leg_kernel_spread():
        # >= 200 more launches of the same scan, each timed with its own pair of HIP events (20 launches
        # = 2.3 ms carry no spread): `value` stays what the K timed steps gave
        more = ctx.rescan_times(max(200, args.steps))
        rf["kernel_ms_min"], rf["kernel_ms_p50"], rf["kernel_ms_p95"], rf["kernel_ms_max"] = (
            float(np.min(more)), float(np.percentile(more, 50)), float(np.percentile(more, 95)), float(np.max(more)))
        rf["kernel_ms_launches"] = int(len(more))
        rf["frac_p50"] = alg_bytes / (rf["kernel_ms_p50"] * 1e-3) / 1e9 / HBM_PEAK_GBS if rf["kernel_ms_p50"] > 0 else None

    def leg_stream_ceiling():
        # the stream-read ceiling measured in this run, on this matrix ("against both the 8 TB/s spec and the
        # measured stream-read ceiling"): a kernel that only reads the matrix, 16 B per lane, as many launches as the timed
        # steps, its clocks settled like the scan's -- outside the timed region
        ctx.stream_read_ceiling(3)
        ceil_ms, ceil_bytes, ceil_shape = ctx.stream_read_ceiling(max(args.steps, 10))
        if ceil_bytes == 0 or ceil_ms <= 0 or mean_ms <= 0:      # an empty (or < 16-byte) slab has no ceiling to quote
            rf["measured_stream_ceiling"], rf["frac_of_measured_ceiling"] = None, None
            return "ok (matrix too small for a ceiling)"
        ceiling = ceil_bytes / (ceil_ms * 1e-3) / 1e9
        stored_rate = int(M) * 8 * wpr / (mean_ms * 1e-3) / 1e9      # bytes of the matrix as stored (= algorithmic unless rows are padded)
        rf["measured_stream_ceiling"] = {"GBps": ceiling, "kernel": "stream_read_kernel (%s)" % ceil_shape, "kernel_ms": ceil_ms,
                                         "bytes_per_launch": int(ceil_bytes), "frac_of_peak": ceiling / HBM_PEAK_GBS,
                                         "what": "the same matrix read once per launch by a kernel that does nothing else (the fastest "
                                                 "of four shapes), timed with HIP events in this run"}
        rf["frac_of_measured_ceiling"] = stored_rate / ceiling

    def leg_hbm_only():
        # the same kernel on a matrix the 256-MiB Infinity Cache cannot hold (the headline matrix of config 2 is 0.73 GB, of
        # which a third stays cached between launches): >= 4 GB of device-generated rows of the same width, rank 0 only
        if rank != 0:
            return "ok (rank 0 only)"
        rows_big = int(max(4.3e9 // (8 * alg_words), 1))
        with PskContext(grp.device) as big:
            # (seed bits 48..63 = 80: the generator then plants config 2's share of associated rows, 0.008 % -- with its default 1 % the
            # host would pick the queued form of the kernel, chi2_scan_kernel<G, 2>, and this block would measure another instantiation)
            big.synth_presence(rows_big, n, seed=(80 << 48) | 11)
            big_npass = big.chi2_scan(pheno, None, 2, n - 2, 0.05, False, rows_big)
            w_ms = big.rescan_timed(3)
            big.rescan_timed(int(min(100, max(3, 40.0 / max(w_ms, 0.01)))))
            ms_big = big.rescan_times(50)
            _, wpr_big, _ = big.presence_shape()
            big.stream_read_ceiling(2)
            c_ms, c_bytes, c_shape = big.stream_read_ceiling(10)
        b_big = rows_big * 8 * alg_words
        p50 = float(np.percentile(ms_big, 50))
        rf["hbm_only"] = {"rows": rows_big, "survivors": int(big_npass), "bytes_per_launch": b_big, "stored_bytes_per_launch": rows_big * 8 * wpr_big,
                          "kernel_ms_mean": float(np.mean(ms_big)), "kernel_ms_min": float(np.min(ms_big)), "kernel_ms_p50": p50,
                          "kernel_ms_p95": float(np.percentile(ms_big, 95)), "launches": int(len(ms_big)),
                          "traffic": measured_traffic(rows_big, wpr_big),
                          "achieved": b_big / (float(np.mean(ms_big)) * 1e-3) / 1e9, "frac": b_big / (float(np.mean(ms_big)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "stream_ceiling_GBps": c_bytes / (c_ms * 1e-3) / 1e9 if c_ms > 0 else None, "stream_ceiling_kernel": c_shape,
                          "what": "the same scan over a device-generated matrix of %.1f GB (psk_synth_presence, same row width): "
                                  "beyond the Infinity Cache, every byte comes from HBM" % (b_big / 1e9)}

    def leg_cpu_baseline():
        # CPU baseline: the oracle's scan (C restatement of modeling.py:677-858, "port") on the same matrix, checked
        # equal to the GPU's answer first; rows are independent, so they are cut into one range per host thread
        # (orc_chi2_scan_mt, POSIX threads -- the reference runs its chunks in a process pool).  One thread alone is
        # timed too, on a part of the rows.  The sample is bounded in CELLS (about 1e10 per pass), whatever the row width.
        from oracle import oracle as O
        ns = int(max(1, min(args.cpu_sample_rows, M, 10_240_000_000 // n)))
        rows = ctx.get_rows(np.arange(ns, dtype=np.uint64))
        threads = host_cpus()
        ph_list, ones = pheno.tolist(), np.ones(n)
        n1 = int(max(1, min(ns, 1_024_000_000 // n)))
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

    # the last scan of the timed steps is the one whose results the CPU leg checks: the spread leg re-runs the SAME scan into
    # the same result set, so `npass` stays what it was
    leg("kernel_spread", leg_kernel_spread)
    leg("stream_ceiling", leg_stream_ceiling)
    leg("cpu_baseline", leg_cpu_baseline)

    def leg_e2e():
        budget = e2e_budget_s(t_launch)
        if world == 1:
            out["e2e"] = e2e_modeling(gs, n, k, cold=args.e2e_cold)
        else:
            if grp.allreduce_sum(1 if budget < 30.0 else 0):     # every rank takes the same branch
                out["e2e"] = {"skipped": "%.0f s left of the launch's deadline (PSK_LAUNCH_TIMEOUT) once 60 s are set aside" % budget}
                return "skipped: no time left before the launch's deadline"
            out["e2e"] = e2e_modeling_sharded(grp, gs, n, k, args, budget_s=budget)
        if "error" in out["e2e"]:
            return "failed: " + out["e2e"]["error"][:300]

    ctx.close()   # the matrix and the lists go before the hbm-only matrix and the CLI children of the e2e leg bring their own
    leg("hbm_only", leg_hbm_only)
    leg("e2e", leg_e2e)
    phase("teardown")
    grp.close()
    line.emit(final=True)


class ResultLine:
    """Rank 0's JSON line.  Printed -- flushed, as the last line of stdout so far -- as soon as `value` exists, and again
    whenever a leg has added to it: whatever happens afterwards (a leg that hangs into the launch's deadline, a child that is
    killed), the LAST JSON line of stdout is a complete metric line, and its `line` / `legs` fields say what it lacks."""

    def __init__(self, rank, out):
        import threading
        self.rank, self.out, self.lock = rank, out, threading.Lock()

    def emit(self, final=False, aborted=None):
        if self.rank != 0:
            return
        with self.lock:
            pending = [k_ for k_, v in self.out.get("legs", {}).items() if v == "pending"]
            self.out["line"] = ("final" if final and not pending else
                                "provisional: legs pending: " + ", ".join(pending) if pending else "provisional")
            if aborted is not None:
                self.out["aborted"] = aborted
            # RCCL prints a version banner through C stdio, which (piped) is flushed at exit, after Python's own
            # buffer: flush it now so that the JSON line is the LAST line of stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
            sys.stdout.write(json.dumps(self.out) + "\n")
            sys.stdout.flush()

    def on_deadline(self, rec):
        """watchdog: the launch ran into its deadline (or this rank was asked where it is): the line again, with the phase."""
        self.emit(aborted={"stuck_in": rec.get("stuck_in"), "blocked_in_call": rec.get("blocked_in_call"),
                           "what": "this rank was asked where it is (SIGUSR1 / PSK_LAUNCH_TIMEOUT): the legs marked pending did not finish"})


if __name__ == "__main__":
    main()
