"""bench.py prints ONE JSON line with the fields the driver and the judge read (-m gpu: needs the device)."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                        "--samples", "64", "--length", "200000", "--cpu-sample-rows", "200000"], cwd=ROOT, timeout=600,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    # r06: the line goes out as soon as `value` exists and again after every leg -- every line of stdout is a complete metric
    # line with the same `value`; the LAST one is the record, and says "final"
    recs = [json.loads(l) for l in lines]
    assert len(recs) >= 2 and all(x["value"] == recs[0]["value"] and x["roofline"]["frac"] == recs[0]["roofline"]["frac"] for x in recs)
    assert recs[0]["line"].startswith("provisional: legs pending: ") and "cpu_baseline" not in recs[0] and "e2e" not in recs[0]
    d = recs[-1]
    assert d["line"] == "final" and set(d["legs"]) == {"kernel_spread", "stream_ceiling", "hbm_only", "cpu_baseline", "e2e"}
    assert all(v.startswith("ok") for v in d["legs"].values()), d["legs"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["achieved"] > 0
    # "against both the 8 TB/s spec and the measured stream-read ceiling" -- a plain read of the same matrix timed
    # in the same run (this test's matrix is a few megabytes: it sits in the caches, so the ceiling may exceed the HBM spec here --
    # but not by the factor of r05's first probe, which lost its loads to the optimiser and claimed 123 TB/s at full size)
    ceil = rf["measured_stream_ceiling"]
    assert 0 < ceil["GBps"] < 60000 and 0 <= rf["stored_bytes_per_launch"] - ceil["bytes_per_launch"] < 16 and "stream_read_kernel" in ceil["kernel"]
    assert abs(rf["frac_of_measured_ceiling"] - (rf["stored_bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9) / ceil["GBps"]) < 1e-9
    assert 0.05 < rf["frac_of_measured_ceiling"] < 3.0
    # the spread of the headline kernel over >= 200 further launches, and the same kernel beyond the Infinity Cache (r06)
    assert rf["kernel_ms_launches"] >= 200 and 0 < rf["kernel_ms_min"] <= rf["kernel_ms_p50"] <= rf["kernel_ms_p95"] <= rf["kernel_ms_max"]
    hb = rf["hbm_only"]
    assert hb["bytes_per_launch"] >= 4.0e9 and hb["launches"] >= 50 and 0 < hb["kernel_ms_min"] <= hb["kernel_ms_p50"] <= hb["kernel_ms_p95"]
    assert abs(hb["frac"] - hb["achieved"] / 8000.0) < 1e-9 and 0.3 < hb["frac"] < 1.0, hb
    assert d["config"]["ingest_per_rank"][0]["count_s"] == d["config"]["ingest"]["count_s"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["single_thread_value"] > 0 and cb["matches_gpu"] is True
    assert d["value"] > cb["value"]
    e2e = d["e2e"]  # BASELINE.json's second figure rides along, outside `value`
    assert e2e["modeling_wall_s"] > 0 and "log_reg_model_Pheno.pkl" in e2e["what"]
    assert e2e["input_files"].startswith("page cache")          # (the files were written a moment earlier: it says so)
    # ... with its phase table: the phases of the run sum to its wall-clock
    ph = e2e["phases"]["rank0"]
    assert abs(sum(ph["phases_s"].values()) - ph["total_s"]) < 1e-2
    assert abs(ph["total_s"] - e2e["modeling_wall_s"]) <= 0.05 * e2e["modeling_wall_s"] + 0.01, (ph, e2e["modeling_wall_s"])
    for key in ("HIP runtime, context", "ingest: k-mer lists", "presence matrix", "scan", "model: result tables, grid search, model files"):
        assert key in ph["phases_s"], ph


def test_bench_two_ranks_run_the_sharded_pipeline():
    """N > 1: ONE dataset, the word space cut at pilot quantiles, every rank its slab (both ingest modes), survivors
    all-gathered; the line carries the rows of every rank, whose sum is the one-rank union.  Two ranks share the one
    GPU here, so the collectives go through the gloo transport of tests/ (RCCL refuses two ranks per device)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PSK_DIST_TRANSPORT="_gloo_transport:GlooTransport", OMP_NUM_THREADS="1",
               PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    size = ["--samples", "64", "--length", "200000", "--kmer", "16", "--steps", "4", "--warmup", "1"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-e2e"] + size, cwd=ROOT,
                         timeout=600, capture_output=True, text=True)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    for port, ingest, transport in ((29631, "filter", "gloo"), (29633, "exchange", "gloo"), (29635, "filter", "host-files"),
                                    (29637, "exchange", "host-files")):
        # `python bench.py --gpus 2` with NO outside launcher: bench.py itself starts the two ranks (launch.py).  Without
        # a named transport RCCL is tried, refuses the shared GPU on both ranks, and -- because --share-gpu opts into it --
        # the run falls back, loudly, to the host-file transport of dist.py
        env_t = dict(env, MASTER_PORT=str(port)) if transport == "gloo" else {k_: v for k_, v in env.items() if k_ != "PSK_DIST_TRANSPORT"}
        for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env_t.pop(var, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--ingest", ingest] + size
        if transport == "gloo":
            cmd.append("--no-e2e")   # the CLI children of the e2e leg would open a second gloo group on the same port
        r = subprocess.run(cmd, env=env_t, cwd=ROOT, timeout=900, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert lines[-1].startswith("{") and sum(l.startswith("{") for l in lines) >= 2     # rank 0's line, relayed last
        d = json.loads(lines[-1])
        assert d["line"] == "final" and all(json.loads(l)["value"] == d["value"] for l in lines if l.startswith("{"))
        cfg = d["config"]
        assert len(cfg["ingest_per_rank"]) == 2 and all(x["presence_s"] is not None for x in cfg["ingest_per_rank"])
        # two ranks on one GPU over a host transport: a correctness run, and the line says so
        assert d["n_gpus"] == 2 and d["scaling"].startswith("invalid (") and d["rccl_ranks"] == 0
        if transport == "gloo":
            assert cfg["collectives"] == "gloo"
        else:
            assert cfg["collectives"].startswith("host-files (fallback: ") and "RCCL communicator not formed" in r.stderr
        assert len(cfg["rows_per_rank"]) == 2 and sum(cfg["rows_per_rank"]) == cfg["rows_global"] == d1["config"]["rows_per_gpu"]
        assert cfg["balance_max_over_mean"] <= 1.10
        assert cfg["survivors_all_slabs"] == d1["config"]["survivors"]
        assert cfg["ingest"]["mode"] == ingest and "range-sharded over 2 GPUs" in cfg["workload"]
        assert d["value"] > 0 and "cpu_baseline" not in d
        if transport != "gloo":   # BASELINE's second figure with several ranks: the CLI as one child process per rank
            e2e = d["e2e"]
            assert "error" not in e2e, e2e
            assert e2e["rc"] == [0, 0] and 0 < e2e["budget_s"] <= 900 and d["legs"]["e2e"] == "ok"
            assert e2e["modeling_wall_s"] > 0 and e2e["ranks"] == 2 and "log_reg_model_Pheno.pkl" in e2e["what"]
            # every rank's phase table, process start to teardown: the slowest rank's total is the leg's wall-clock less what
            # the interpreter and the HIP runtime need to exit -- r05: the CLI no longer leaves through os._exit, so that part
            # is the honest one now (0.3-0.6 s for two ranks that share a GPU; 15 % + 0.5 s allowed)
            ph = e2e["phases"]
            assert set(ph) == {"rank0", "rank1"} and all(v is not None for v in ph.values()), ph
            for v in ph.values():
                assert abs(sum(v["phases_s"].values()) - v["total_s"]) < 1e-2
                assert any(k.startswith("process start") for k in v["phases_s"]) and any(k.startswith("rendezvous") for k in v["phases_s"])
                assert any(k.startswith("ingest: k-mer lists (") for k in v["phases_s"]) and "survivor all-gather" in v["phases_s"]
            slowest = max(v["total_s"] for v in ph.values())
            assert slowest <= e2e["modeling_wall_s"] + 0.01 and e2e["modeling_wall_s"] - slowest <= 0.15 * e2e["modeling_wall_s"] + 0.5, (slowest, e2e["modeling_wall_s"])


def _children_of(pids):
    """pid -> argv of the live children of `pids` (one pass over /proc; exact parent pids, no pattern matching on names)."""
    out = {}
    for ent in os.listdir("/proc"):
        if not ent.isdigit():
            continue
        try:
            with open("/proc/%s/stat" % ent) as f:
                ppid = int(f.read().rsplit(")", 1)[1].split()[1])
            if ppid in pids:
                with open("/proc/%s/cmdline" % ent, "rb") as f:
                    out[int(ent)] = f.read().split(b"\0")
        except (OSError, ValueError, IndexError):
            pass
    return out


def test_bench_line_survives_the_death_of_the_e2e_leg():
    """VERDICT r05 next #1 'done' criterion: the e2e leg's CLI children are KILLED mid-run (their exact pids, found as children of
    the two rank processes) -- the last JSON line of stdout is still a valid metric line, the same `value` as the line printed
    before the legs, with the children's exit codes recorded in it.  Second half: the children hang (SIGSTOP) -- the leg's
    budget ends them before the launch's deadline.  (The deadline itself re-emitting the line: tests/test_bench_line.py.)"""
    import signal
    import time
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PSK_DIST_TRANSPORT", "PSK_RDZV_DIR",
                                                            "PSK_RDZV_FILE", "PSK_LAUNCH_NONCE")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--samples", "64", "--length", "200000",
           "--kmer", "16", "--steps", "4", "--warmup", "1"]

    def run(action, launch_timeout):
        p = subprocess.Popen(cmd, env=dict(env, PSK_LAUNCH_TIMEOUT=str(launch_timeout)), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        hit = set()
        t0 = time.time()
        try:
            while p.poll() is None and time.time() - t0 < 600:
                ranks = [pid for pid, argv in _children_of({p.pid}).items() if any(a.endswith(b"bench.py") for a in argv)]
                for pid, argv in _children_of(set(ranks)).items():
                    if pid not in hit and any(a.endswith(b"scripts/phenotypeseeker") for a in argv):
                        os.kill(pid, action)
                        hit.add(pid)
                time.sleep(0.005)
            out, err = p.communicate(timeout=60)
        finally:
            if p.poll() is None:
                p.kill()
            for pid in hit:          # (stopped children of a launch that was torn down)
                try:
                    os.kill(pid, signal.SIGKILL)
                except OSError:
                    pass
        return p.returncode, out, err, hit

    rc, out, err, hit = run(signal.SIGKILL, 600)
    assert len(hit) == 2, (hit, err[-2000:])
    recs = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert out.strip().splitlines()[-1].startswith("{") and len(recs) >= 2
    d = recs[-1]
    assert rc == 0 and d["line"] == "final" and d["value"] == recs[0]["value"] > 0 and d["roofline"]["frac"] > 0
    assert d["legs"]["e2e"].startswith("failed: 2 rank(s) failed") and d["e2e"]["rc"] == [-9, -9], d["e2e"]
    assert d["e2e"]["first_failing_phase"]["no_table_from_ranks"] == [0, 1]       # SIGKILL leaves no table; a hang does (below)
    assert all(v.startswith("ok") for k_, v in d["legs"].items() if k_ != "e2e"), d["legs"]

    # children that hang (stopped): the leg's own budget -- min(900, deadline - elapsed - 60) -- fires before the launch's
    # deadline does; the children are killed by pid, the codes (124) are in the line, the launch ends normally
    rc, out, err, hit = run(signal.SIGSTOP, 110)
    assert len(hit) == 2, (hit, err[-2000:])
    recs = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    d = recs[-1]
    assert rc == 0 and "deadline of 110 s passed" not in err
    assert out.strip().splitlines()[-1].startswith("{") and d["line"] == "final" and d["value"] == recs[0]["value"] > 0
    assert d["e2e"]["rc"] == [124, 124] and 30 <= d["e2e"]["budget_s"] <= 50 and d["legs"]["e2e"].startswith("failed: 2 rank(s) failed")


def test_bench_eight_ranks_on_one_gpu_exchange_and_filter():
    """The 8-GPU preflight: `bench.py --gpus 8` exactly as the driver's scaling run will start it -- no outside
    launcher, eight ranks of launch.py -- with the one GPU of this box shared (--share-gpu: RCCL refuses, all ranks fall back
    together to the host-file transport and the line says "invalid").  Both ingest modes: the union is the one-rank union, cut
    into eight balanced slabs, the survivors of all slabs are the one-rank survivors, `ingest.exchange_s` and `rccl_ranks` are
    in the line; the refusal quotes what RCCL itself said."""
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PSK_DIST_TRANSPORT", "PSK_RDZV_DIR",
                                                            "PSK_RDZV_FILE", "PSK_LAUNCH_NONCE")}
    size = ["--samples", "64", "--length", "200000", "--kmer", "16", "--steps", "3", "--warmup", "1", "--no-e2e", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + size, cwd=ROOT, env=env, timeout=600, capture_output=True, text=True)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    for ingest in ("exchange", "filter"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--ingest", ingest] + size,
                           env=dict(env, PSK_LAUNCH_TIMEOUT="600"), cwd=ROOT, timeout=900, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        cfg = d["config"]
        assert d["n_gpus"] == 8 and d["scaling"].startswith("invalid (") and d["rccl_ranks"] == 0
        assert cfg["collectives"].startswith("host-files (fallback: ") and "RCCL communicator not formed" in r.stderr
        assert len(cfg["rows_per_rank"]) == 8 and sum(cfg["rows_per_rank"]) == cfg["rows_global"] == d1["config"]["rows_per_gpu"]
        assert cfg["balance_max_over_mean"] <= 1.15
        assert cfg["survivors_all_slabs"] == d1["config"]["survivors"]
        assert cfg["ingest"]["mode"] == ingest and "exchange_s" in cfg["ingest"] and "range-sharded over 8 GPUs" in cfg["workload"]
        if ingest == "exchange":
            print("eight ranks, one GPU:", cfg["collectives"])
            assert "RCCL said: " in cfg["collectives"] and "NCCL WARN" in cfg["collectives"]    # the refusal in RCCL's own words


def test_bench_without_rccl_and_without_the_opt_in_fails():
    """Un-fakeable N > 1 (VERDICT r02 / ADVICE r02): two ranks that CLAIM a GPU each (no --share-gpu) on a one-GPU box --
    rank 1's device does not exist, RCCL cannot form the communicator -- must not fall back to host files: rc != 0 and
    no JSON line.  With `--force-exchange` on one GPU the step runs on a one-rank RCCL communicator and quotes
    ncclCommCount."""
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PSK_DIST_TRANSPORT", "PSK_SHARE_GPU",
                                                            "PSK_DIST_ALLOW_HOST_FILES")}
    size = ["--samples", "64", "--length", "200000", "--kmer", "16", "--steps", "3", "--warmup", "1", "--no-e2e", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + size, env=env, cwd=ROOT, timeout=600,
                       capture_output=True, text=True)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-exchange"] + size, env=env, cwd=ROOT,
                       timeout=600, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["rccl_ranks"] == 1 and d["config"]["collectives"] == "rccl" and d["scaling"] == "weak"


def test_bench_under_the_drivers_launcher():
    """The driver starts N > 1 as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: the ranks then
    come with RANK / LOCAL_RANK / WORLD_SIZE in their environment and bench.py must NOT start ranks of its own; they meet in
    the per-user rendezvous directory keyed by the launcher process (no PSK_RDZV_DIR).  Two ranks on the one GPU here
    (--share-gpu: host-file collectives, so the line says "invalid"); torch is the launcher only -- the ranks never import it."""
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PSK_DIST_TRANSPORT", "PSK_RDZV_DIR",
                                                            "PSK_RDZV_FILE", "PSK_LAUNCH_NONCE")}
    launcher = "torch.distributed" + ".run"
    cmd = [sys.executable, "-m", launcher, "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29641",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--samples", "64", "--length", "200000", "--kmer", "16",
           "--steps", "3", "--warmup", "1", "--no-e2e"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, timeout=900, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) >= 2 and json.loads(lines[-1])["line"] == "final"      # rank 0 alone prints; its last line is the record
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"].startswith("invalid (") and d["config"]["collectives"].startswith("host-files")
    assert len(d["config"]["rows_per_rank"]) == 2 and d["value"] > 0
