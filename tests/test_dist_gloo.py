"""N > 1 path on CPU: two gloo ranks, each owning one slab of the k-mer word space, must
reproduce the single-rank result byte for byte (SURVEY.md 8(e) invariant)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, load_dataset


def test_slab_bounds_tile_the_word_space():
    from phenotypeseeker_amd.dist import slab_bounds
    for k in (1, 5, 13, 16, 31, 32):
        for world in (1, 2, 3, 4, 8):
            if world > 4 ** k:
                continue
            edges = [slab_bounds(k, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == 0
            for (lo, hi), (lo2, _) in zip(edges[:-1], edges[1:]):
                assert hi == lo2 and hi > lo
            assert all(lo < (1 << 64) and hi < (1 << 64) for lo, hi in edges)


def _slab_shares(union_words, bounds):
    edges = [0] + [int(np.searchsorted(union_words, np.uint64(b))) for b in bounds[1:-1]] + [len(union_words)]
    return np.diff(edges).astype(float)


def test_balanced_bounds_on_uniform_and_at_rich_genomes(oracle):
    """uniform cuts of the word space are 1.87x imbalanced at 8 ranks (canonical words have
    density ~2(1-u)); the quantile cuts of a four-list pilot must hold max/mean <= 1.10 for W in {2, 4, 8}, on uniform
    ACGT and on a 29 %-GC set (the reference's example organism), and so must the closed form on uniform sequence."""
    from phenotypeseeker_amd import dist
    from phenotypeseeker_amd.synth import GenomeSet
    for gc, k in ((0.5, 13), (0.29, 13), (0.29, 16)):
        gs = GenomeSet(6, 300_000, seed=5, gene_len=500, gc=gc, contigs=3)
        lists = [oracle.count_kmers(gs.sample(i)[1], k)[0] for i in range(6)]
        uw = oracle.union(lists)
        for world in (2, 4, 8):
            uni = _slab_shares(uw, [dist.slab_bounds(k, world, r)[0] for r in range(world)] + [0])
            assert uni.max() / uni.mean() > 1.3                                    # what is being fixed
            pts = np.concatenate([dist.pilot_points(w) for w in lists[:4]])
            b = dist.quantile_bounds(pts, k, world)
            assert b[0] == 0 and b[-1] == 0 and all(x < y for x, y in zip(b[:-2], b[1:-1]))
            sh = _slab_shares(uw, b)
            assert sh.max() / sh.mean() <= 1.10, (gc, k, world, sh)
            if gc == 0.5:
                sh = _slab_shares(uw, dist.canonical_cdf_bounds(k, world))
                assert sh.max() / sh.mean() <= 1.10, ("closed form", k, world, sh)
    # degenerate inputs still give legal bounds
    assert dist.quantile_bounds(np.zeros(0, np.uint64), 2, 4)[1:-1] == sorted(set(dist.quantile_bounds(np.zeros(0, np.uint64), 2, 4)[1:-1]))
    b = dist.quantile_bounds(np.full(100, 7, np.uint64), 3, 4)
    assert all(x < y for x, y in zip(b[:-2], b[1:-1])) and b[-2] < 64


def test_rendezvous_carries_the_unique_id_and_ignores_stale_or_foreign_files(tmp_path):
    """The RCCL unique id travels through a private directory: rank 0 publishes atomically, the others poll.  A blob
    without this launch's nonce (a stale file of a crashed run, somebody else's file) is never accepted, rank 0 clears
    what an earlier transport of the same name left behind, and a directory others can write is refused."""
    import threading
    from phenotypeseeker_amd import dist
    d = dist._private_dir(os.path.join(tmp_path, "rdzv"))
    assert os.stat(d).st_mode & 0o077 == 0
    # leftovers of an earlier launch under the same names: an id with another nonce, a collective directory
    dist._publish(os.path.join(d, "id.0"), dist._MAGIC + (3).to_bytes(2, "little") + b"old" + (4).to_bytes(4, "little") + b"dead")
    os.mkdir(os.path.join(d, "coll.0"))
    got = {}

    def reader(r):
        got[r] = dist.exchange_unique_id(r, 3, None, timeout=20, rdzv=(d, "nonce-of-this-launch"))[0]

    ts = [threading.Thread(target=reader, args=(r,)) for r in (1, 2)]
    for t in ts:
        t.start()
    import time
    time.sleep(0.3)
    assert got == {}                       # the stale id is there, and nobody took it
    uid = bytes(range(128))
    assert dist.exchange_unique_id(0, 3, lambda: uid, rdzv=(d, "nonce-of-this-launch"))[0] == uid
    for t in ts:
        t.join()
    assert got == {1: uid, 2: uid}
    assert sorted(os.listdir(d)) == ["id.0"]      # rank 0 removed the leftovers before publishing
    with pytest.raises(RuntimeError):
        dist.exchange_unique_id(1, 2, None, timeout=0.3, rdzv=(d, "another-launch"))
    loose = os.path.join(tmp_path, "loose")
    os.mkdir(loose, 0o777)
    os.chmod(loose, 0o777)
    with pytest.raises(RuntimeError):
        dist._private_dir(loose)
    # the default meeting place is keyed by the launcher process: pid AND start time of the parent
    for var in ("PSK_RDZV_DIR", "PSK_RDZV_FILE", "PSK_LAUNCH_NONCE"):
        os.environ.pop(var, None)
    d1, n1 = dist._rendezvous()
    assert str(os.getppid()) in os.path.basename(d1) and dist._parent_start_ticks() in os.path.basename(d1) and n1
    assert os.stat(os.path.dirname(d1)).st_mode & 0o077 == 0
    os.rmdir(d1)


def test_launcher_starts_ranks_relays_rank0_and_returns_the_worst_code(tmp_path):
    """launch.spawn_ranks (what `bench.py --gpus N` and `PSK_GPUS=N phenotypeseeker ...` use instead of an outside
    launcher): one child per rank with RANK / LOCAL_RANK / WORLD_SIZE, a private rendezvous directory and a nonce;
    rank 0's stdout is the launcher's, the other ranks' goes to stderr; the exit code is the worst rank's, and a rank
    that hangs after another has failed is terminated."""
    prog = os.path.join(tmp_path, "rank.py")
    with open(prog, "w") as f:
        f.write("import os, sys, time\n"
                "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
                "d = os.environ['PSK_RDZV_DIR']\n"
                "assert os.stat(d).st_mode & 0o077 == 0 and len(os.environ['PSK_LAUNCH_NONCE']) == 32 and os.environ['LOCAL_RANK'] == str(r)\n"
                "print('line of rank %d of %d' % (r, w), flush=True)\n"
                "mode = sys.argv[1]\n"
                "if mode == 'fail' and r == 1: sys.exit(7)\n"
                "if mode == 'fail' and r == 2: time.sleep(600)\n")
    code = "import sys; sys.path.insert(0, %r); from phenotypeseeker_amd import launch; sys.exit(launch.spawn_ranks([%r, sys.argv[1]], 3, grace_s=1.0))" % (ROOT, prog)
    r = subprocess.run([sys.executable, "-c", code, "ok"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.splitlines() == ["line of rank 0 of 3"]
    assert "line of rank 1 of 3" in r.stderr and "line of rank 2 of 3" in r.stderr
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code, "fail"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 7 and time.time() - t0 < 60 and "rank 1 exited with code 7" in r.stderr


def test_status_rounds_make_a_failed_rank_everybodys_failure(tmp_path, monkeypatch):
    """dist._rccl_or_host_files without a GPU (the context is stubbed): a rank that has no device says so BEFORE anybody
    would enter ncclCommInitRank, and every rank raises -- no fallback unless the run opted in; with the
    opt-in all ranks take the host-file transport together and carry the reason."""
    import threading
    from phenotypeseeker_amd import dist

    class Ctx:
        def __init__(self, device):
            if device == 1 and Ctx.break_rank1:
                raise RuntimeError("no such device")

        def close(self):
            pass
    Ctx.break_rank1 = True

    class NoRccl:
        def __init__(self, rank, world, device, rdzv=None, seq=0):
            raise AssertionError("ncclCommInitRank must not be reached when a rank has reported a failure")
    monkeypatch.setattr(dist, "PskContext", Ctx)
    monkeypatch.setattr(dist, "RcclTransport", NoRccl)
    monkeypatch.setenv("PSK_RDZV_DIR", str(tmp_path / "meet"))
    monkeypatch.setenv("PSK_LAUNCH_NONCE", "n1")
    for var in ("PSK_SHARE_GPU", "PSK_DIST_ALLOW_HOST_FILES", "PSK_DIST_STRICT"):
        monkeypatch.delenv(var, raising=False)
    out = {}

    class FirstTransport(list):        # the two "processes" are threads of one module: both form their FIRST transport
        def __getitem__(self, i):
            return 0

        def __setitem__(self, i, v):
            pass
    monkeypatch.setattr(dist, "_rdzv_seq", FirstTransport([0]))

    def rank_main(r):
        try:
            t = dist._rccl_or_host_files(r, 2, r)
            out[r] = (t.name, getattr(t, "fallback_reason", None))
            t.close()
        except Exception as e:  # noqa: BLE001
            out[r] = repr(e)
    for allow in (False, True):
        if allow:
            monkeypatch.setenv("PSK_DIST_ALLOW_HOST_FILES", "1")
            monkeypatch.setenv("PSK_RDZV_DIR", str(tmp_path / "meet2"))      # another launch: another meeting place
            Ctx.break_rank1 = False       # the stand-in for "RCCL refuses": the device is there, the communicator is not

            class Refuses:
                def __init__(self, rank, world, device, rdzv=None, seq=0):
                    raise RuntimeError("two ranks on one device")
            monkeypatch.setattr(dist, "RcclTransport", Refuses)
        out.clear()
        th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
        for x in th:
            x.start()
        for x in th:
            x.join(60)
        if not allow:
            assert all("RCCL communicator not formed (rank 1" in out[r] and "no such device" in out[r] for r in range(2)), out
        else:
            assert out == {0: ("host-files", "RuntimeError: two ranks on one device"), 1: ("host-files", "RuntimeError: two ranks on one device")}


def test_pack_merge_round_trip():
    from phenotypeseeker_amd import dist
    rng = np.random.default_rng(0)
    parts = []
    for n in (0, 3, 5):
        res = {"word": np.sort(rng.integers(0, 1 << 40, n)).astype(np.uint64), "stat": rng.random(n), "p": rng.random(n),
               "mean_x": rng.random(n), "mean_y": rng.random(n), "n_with": rng.integers(0, 99, n).astype(np.int32)}
        parts.append((res, rng.integers(0, 1 << 62, (n, 4)).astype(np.uint64)))
    merged, bits = dist.merge_candidates([dist.pack_candidates(r, b) for r, b in parts])
    assert len(merged["word"]) == 8 and bits.shape == (8, 4)
    assert np.array_equal(merged["stat"], np.concatenate([p[0]["stat"] for p in parts]))
    assert np.array_equal(bits, np.concatenate([p[1] for p in parts]))


@pytest.mark.parametrize("tag,port", [("ds_omitB", "29617"), ("ds_k21", "29627")])      # (ds_k21: 64-bit words, slab bounds beyond 2^32)
def test_two_rank_gloo_run_equals_single_rank(tmp_path, oracle, tag, port):
    out = os.path.join(tmp_path, "merged.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", PSK_DIST_TRANSPORT="_gloo_transport:GlooTransport",
               PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env["MASTER_PORT"] = port
    env["PSK_TEST_DATASET"] = tag
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(var, None)
    # the package's own launcher (launch.spawn_ranks): no outside launcher anywhere in the tests
    code = "import sys; sys.path.insert(0, %r); from phenotypeseeker_amd import launch; sys.exit(launch.spawn_ranks(sys.argv[1:], 2))" % ROOT
    cmd = [sys.executable, "-c", code, os.path.join(ROOT, "tests", "_dist_worker.py"), out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, timeout=300, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(out)
    assert int(z["world"]) == 2 and float(z["tmax"]) == 2.0
    shares = z["shares"].astype(float)
    assert shares.max() / shares.mean() <= 1.10, shares      # quantile cuts: the two slabs hold the same share of the rows
    ds = load_dataset(tag)
    k, names, n = ds["meta"]["k"], ds["names"], len(ds["names"])
    wl = [oracle.count_kmers(ds["files"][nm], k)[0] for nm in names]
    uw = oracle.union(wl)
    assert int(z["m_global"]) == len(uw) == ds["meta"]["n_union"]
    assert int(z["pairs"]) == sum(len(w) for w in wl)   # the list exchange moved every (word, sample) pair once
    bits = oracle.presence_bits(wl, uw, wpr=z["bits"].shape[1])
    ref = oracle.chi2_scan(bits, ds["pheno"], np.ones(n), n, 2, n - 2, 0.05, True, len(uw))
    keep = np.nonzero(ref["keep"])[0]
    assert np.array_equal(z["word"], uw[keep])
    assert np.array_equal(z["stat"], ref["stat"][keep]) and np.array_equal(z["p"], ref["p"][keep])
    assert np.array_equal(z["n_with"], ref["n_with"][keep]) and np.array_equal(z["bits"], bits[keep])


def test_host_file_transport_collectives_and_close(tmp_path):
    """The transport the ranks fall back to when RCCL cannot form the communicator: all-reduce, all-gather and the
    closing handshake with four threads as ranks of unequal speed (the context that stages device buffers is stubbed:
    no GPU here).  A rank used to remove its last file at close() before a slower rank had read it -- the slower one
    then sat out the whole timeout."""
    import threading
    import time
    from phenotypeseeker_amd import dist

    class NoCtx:
        def __init__(self, device):
            pass

        def close(self):
            pass

    real = dist.PskContext
    dist.PskContext = NoCtx
    try:
        world, out, errs = 4, {}, []
        rd = dist._private_dir(str(tmp_path / "rdzv"))

        def rank_main(r):
            try:
                t = dist.HostFileTransport(r, world, 0, rdzv=(rd, ""), seq=3, timeout=30.0)
                for it in range(6):
                    s = t.allreduce(np.array([r + it], dtype=np.uint64), "sum")
                    m = t.allreduce(np.array([float(r * it)]), "max")
                    g = t.allgather_host(np.full(5 + it, r, dtype=np.uint8))
                    assert int(s[0]) == sum(range(world)) + world * it and float(m[0]) == float((world - 1) * it)
                    assert g.shape == (world, 5 + it) and all((g[q] == q).all() for q in range(world))
                    if r == it % world:
                        time.sleep(0.05)          # a straggler, a different one every round
                if r != 0:
                    time.sleep(0.2 * r)           # the others reach close() long after rank 0's last collective
                t0 = time.time()
                t.close()
                out[r] = time.time() - t0
            except Exception as e:  # noqa: BLE001
                errs.append((r, repr(e)))

        th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for x in th:
            x.start()
        for x in th:
            x.join(60)
        assert not errs, errs
        assert sorted(out) == list(range(world)) and max(out.values()) < 5.0
        assert os.listdir(rd) == []      # rank 0 removed the collective directory last
    finally:
        dist.PskContext = real


def test_psk_gpus_fans_out_modeling_only(monkeypatch, capsys):
    """`PSK_GPUS=N phenotypeseeker ...` starts ranks for `modeling` only -- `prediction` is not rank-aware (N copies
    would race on predictions_<name>.txt and log.txt), --help / --version need none.  The decision is taken from the raw
    arguments before anything GPU-bound is imported."""
    from phenotypeseeker_amd import cli, launch
    calls = []
    monkeypatch.setattr(launch, "spawn_ranks", lambda argv, n, **kw: calls.append((list(argv), n)) or 0)
    monkeypatch.setenv("PSK_GPUS", "4")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        cli.main(["modeling", "data.pheno", "-w"])
    assert e.value.code == 0 and calls == [(["-m", "phenotypeseeker_amd.cli", "modeling", "data.pheno", "-w"], 4)]
    for argv in (["--help"], ["modeling", "--help"], ["--version"]):
        with pytest.raises(SystemExit):
            cli.main(argv)
    assert len(calls) == 1                       # no ranks for help / version
    ran = []
    from phenotypeseeker_amd import prediction
    monkeypatch.setattr(prediction, "prediction", lambda args: ran.append(args))
    cli.main(["prediction", "in1.txt", "in2.txt"])
    assert len(calls) == 1 and len(ran) == 1     # one process, the sub-command itself
    assert "not sharded" in capsys.readouterr().err
    monkeypatch.setenv("WORLD_SIZE", "4")        # under an outside launcher the ranks come here directly
    ran_m = []
    from phenotypeseeker_amd import modeling
    monkeypatch.setattr(modeling, "modeling", lambda args: ran_m.append(args))
    cli.main(["modeling", "data.pheno"])
    assert len(calls) == 1 and len(ran_m) == 1
    monkeypatch.setenv("PSK_GPUS", "many")
    monkeypatch.delenv("WORLD_SIZE")
    with pytest.raises(SystemExit) as e:
        cli.main(["modeling", "data.pheno"])
    assert "PSK_GPUS" in str(e.value.code)


def test_status_files_of_another_launch_are_not_read(tmp_path):
    """the status rounds of the RCCL bring-up (`rd.<seq>.<rank>`, `st.<seq>.<rank>`) carry the launch's nonce like
    the id blob does: what a crashed earlier launch left in a caller-supplied rendezvous directory -- an 'ok' without a
    tag, or with another launch's -- is waited out instead of being taken for this launch's answer."""
    import threading
    from phenotypeseeker_amd import dist
    d = str(tmp_path)
    with open(os.path.join(d, "rd.0.1"), "wb") as f:
        f.write(b"ok")                                            # r03's format, stale
    with pytest.raises(RuntimeError, match="no status"):
        dist._exchange_status(d, "rd.0", 0, 2, "", timeout=0.3, nonce="this-launch")
    tag = dist._MAGIC + (5).to_bytes(2, "little") + b"other"
    with open(os.path.join(d, "rd.0.1"), "wb") as f:
        f.write(tag + b"ok")                                      # another launch's
    with pytest.raises(RuntimeError, match="no status"):
        dist._exchange_status(d, "rd.0", 0, 2, "", timeout=0.3, nonce="this-launch")
    got = {}
    t = threading.Thread(target=lambda: got.setdefault(1, dist._exchange_status(d, "rd.0", 1, 2, "no GPU", timeout=5.0, nonce="this-launch")))
    t.start()
    got[0] = dist._exchange_status(d, "rd.0", 0, 2, "", timeout=5.0, nonce="this-launch")
    t.join()
    assert got[0] == got[1] == ["ok", "no GPU"]


@pytest.mark.parametrize("mode", ["all-hang", "rank1-hangs"])
def test_launch_deadline_leaves_stuck_phase_tables_and_a_nonzero_code(tmp_path, mode):
    """ranks that hang in a collective (all of them, or one while the others have left) are not waited for
    until somebody's 1,800-s kill: the launch has its own deadline (PSK_LAUNCH_TIMEOUT), at which every live rank is asked
    where it is (SIGUSR1), answers with phases_rank<r>.json -- finished phases, the phase it is in, the call it is blocked in;
    written by a thread, the main thread never comes back from the call -- and is then terminated by pid; the launcher
    returns 124 well inside the deadline + report + kill budget."""
    import json
    import time
    code = ("import sys, os\nsys.path.insert(0, %r)\nfrom phenotypeseeker_amd import launch\n"
            "sys.exit(launch.spawn_ranks([%r, %r], 2))\n" % (ROOT, os.path.join(ROOT, "tests", "_hang_worker.py"), mode))
    env = dict(os.environ, PSK_LAUNCH_TIMEOUT="2")
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=120)
    took = time.time() - t0
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert took < 2 + 3 + 5 + 10, took
    assert "deadline of 2 s passed (PSK_LAUNCH_TIMEOUT)" in r.stderr
    stuck = [1] if mode == "rank1-hangs" else [0, 1]
    for rank in stuck:
        with open(os.path.join(tmp_path, "phases_rank%d.json" % rank)) as f:
            rec = json.load(f)
        assert rec["stuck"] is True and rec["rank"] == rank and rec["world"] == 2
        assert rec["stuck_in"] == "all-reduce of the union size"
        assert rec["blocked_in_call"] == "all-reduce(sum) (test-transport, 2 ranks)" and rec["blocked_for_s"] >= 1.0
        assert list(rec["phases_s"]) == ["arguments, data.pheno", "presence matrix"]
        assert "rank %d/2: stuck in phase `all-reduce of the union size`" % rank in r.stderr
    if mode == "rank1-hangs":
        assert not os.path.exists(os.path.join(tmp_path, "phases_rank0.json")) and "rank 0 done" in r.stdout
    # no deadline: PSK_LAUNCH_TIMEOUT=0 is honoured (the launch then ends with its ranks); a value that is no number is refused
    from phenotypeseeker_amd import launch
    os.environ["PSK_LAUNCH_TIMEOUT"] = "soon"
    try:
        with pytest.raises(SystemExit):
            launch.launch_timeout()
    finally:
        del os.environ["PSK_LAUNCH_TIMEOUT"]
    assert launch.launch_timeout() == 900.0


def test_ranks_started_by_different_parents_meet_in_a_supplied_directory(tmp_path):
    """PSK_RDZV_DIR without PSK_LAUNCH_NONCE is for ranks that do NOT share a parent (one wrapper per
    rank).  Each rank here is the child of its own shell; the tag of the id / status files must not depend on the parent
    (r04 derived it from the parent's pid + start time: such ranks rejected each other's files until the timeout)."""
    meet = str(tmp_path / "meet")
    code = ("import os, sys\nsys.path.insert(0, %r)\nfrom phenotypeseeker_amd import dist\n"
            "r = int(os.environ['RANK'])\n"
            "d, nonce = dist._rendezvous()\n"
            "print('nonce=%%r ppid=%%d' %% (nonce, os.getppid()))\n"
            "uid, _ = dist.exchange_unique_id(r, 2, lambda: b'U' * 128, timeout=20, rdzv=(d, nonce))\n"
            "st = dist._exchange_status(d, 'rd.0', r, 2, '', timeout=20, nonce=nonce)\n"
            "assert uid == b'U' * 128 and st == ['ok', 'ok'], (uid, st)\n" % ROOT)
    script = tmp_path / "rank.py"
    script.write_text(code)
    procs = []
    for rank in (1, 0):
        env = {k_: v for k_, v in os.environ.items() if k_ not in ("PSK_LAUNCH_NONCE", "PSK_RDZV_FILE")}
        env.update(PSK_RDZV_DIR=meet, RANK=str(rank), WORLD_SIZE="2")
        # `sh -c "python ...; true"`: the shell stays the parent of the rank (no exec), one shell per rank
        procs.append(subprocess.Popen(["sh", "-c", "%s %s; rc=$?; exit $rc" % (sys.executable, script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=60) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    ppids = {o[0].split("ppid=")[1].strip() for o in outs}
    assert len(ppids) == 2 and all("nonce=''" in o[0] for o in outs)


def test_a_rank_of_an_outside_launcher_keeps_the_deadline_itself(tmp_path):
    """Under the driver's `torch.distributed.run` (or srun) no launch.spawn_ranks sits above the ranks: every rank then keeps
    PSK_LAUNCH_TIMEOUT itself -- its own stuck-phase table, then exit code 124 (watchdog.self_deadline)."""
    import json
    import time
    code = ("import os, sys, time\nsys.path.insert(0, %r)\nfrom phenotypeseeker_amd import watchdog\n"
            "watchdog.install(os.environ['RANK'], os.environ['WORLD_SIZE'], lambda: {'phases_s': {'ingest': 0.5}})\n"
            "watchdog.enter('scan')\nwatchdog.self_deadline(1.5)\n"
            "with watchdog.blocking('all-gather (rccl, 8 ranks)'):\n    r, w = os.pipe(); os.read(r, 1)\n" % ROOT)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=dict(os.environ, RANK="3", WORLD_SIZE="8"),
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 124 and time.time() - t0 < 20, (r.returncode, r.stderr[-1000:])
    with open(os.path.join(tmp_path, "phases_rank3.json")) as f:
        rec = json.load(f)
    assert rec["stuck_in"] == "scan" and rec["blocked_in_call"] == "all-gather (rccl, 8 ranks)" and rec["world"] == 8
    assert rec["phases_s"] == {"ingest": 0.5} and "deadline of 2 s passed" in r.stderr or "deadline of 1 s passed" in r.stderr


def test_a_supplied_rendezvous_directory_gets_a_launcher_wide_tag_where_one_exists(monkeypatch, tmp_path):
    """PSK_RDZV_DIR without PSK_LAUNCH_NONCE used to mean an EMPTY tag on the id / status files -- a rank of the next
    launch could read the RCCL unique id a crashed launch left there.  Now the tag is made of what is equal on every rank of one
    launch and differs between launches, when there is such a thing: the elastic launcher's run id + restart count, Slurm's job +
    step, or the common parent when all ranks of the job are children of one launcher on this node; else it stays empty (ranks of
    different parents: the test above)."""
    from phenotypeseeker_amd import dist
    for var in ("PSK_LAUNCH_NONCE", "PSK_RDZV_FILE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "SLURM_JOB_ID", "SLURM_STEP_ID",
                "LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("PSK_RDZV_DIR", str(tmp_path / "meet"))
    assert dist._rendezvous() == (str(tmp_path / "meet"), "")
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")                     # two nodes: no common parent
    assert dist._rendezvous()[1] == ""
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")                     # every rank a child of this node's launcher
    tag = dist._rendezvous()[1]
    assert tag.startswith("pp_%d_" % os.getppid()) and tag == dist._rendezvous()[1]
    monkeypatch.setenv("SLURM_JOB_ID", "4711")
    monkeypatch.setenv("SLURM_STEP_ID", "2")
    assert dist._rendezvous()[1] == "sl_4711_2"
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "run-a")
    assert dist._rendezvous()[1] == "te_run_a_0"
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "3")           # a restarted gang must not meet the files of the attempt before
    assert dist._rendezvous()[1] == "te_run_a_3"
    monkeypatch.setenv("PSK_LAUNCH_NONCE", "given")                 # what the caller gives wins
    assert dist._rendezvous()[1] == "given"
