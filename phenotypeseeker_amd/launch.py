"""Launcher-free fan-out over the GPUs of one node: `python bench.py --gpus N`, `PSK_GPUS=N phenotypeseeker modeling ...`.

The reference's own parallel axis needs no outside launcher -- `Pool(num_threads)` inside the program
(/root/reference/PhenotypeSeeker/modeling.py:1649-1663, :335-342) -- so neither does this one: the PARENT, which
never touches the GPU (this module imports the standard library only: no libpsk.so, no HIP call, so starting
children is not an exec from a process that has initialised the GPU), starts one child process per rank with
RANK / LOCAL_RANK / WORLD_SIZE and a PRIVATE rendezvous directory of this launch (mkdtemp: mode 0700, a name
nobody can predict or pre-create) plus a nonce the rendezvous blob must carry, relays the children's output, and
exits with the worst child's exit code.  Rank 0's stdout is the parent's stdout (its last line is the program's
result line); the other ranks' stdout goes to stderr.  When one rank fails the others get a grace period and are
then terminated -- by their exact pids, never by pattern.  The launch as a whole has a deadline (PSK_LAUNCH_TIMEOUT seconds,
default 900 for bench.py -- well under the 1,800 s after which a driver kills the job without a trace --, none for the CLI): when it passes, every rank still
running is sent SIGUSR1 -- a rank answers by writing phases_rank<r>.json with the phase and the call it is stuck in
(watchdog.py) --, then terminated by pid, and the launcher leaves with 124 (what `timeout` returns).  A launcher that already exported WORLD_SIZE
(`torchrun`-style, srun, mpirun) is honoured instead: the programs only read RANK / LOCAL_RANK / WORLD_SIZE.
"""
import os
import secrets
import shutil
import signal
import subprocess
import sys
import tempfile
import time


def launched_by_outside_launcher():
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def launch_timeout(default=900.0):
    """PSK_LAUNCH_TIMEOUT in seconds (0 or negative: no deadline).  default: what holds without the variable -- 900 for the
    benchmark (a bounded job under a driver that kills at 1,800), none for `phenotypeseeker modeling` (a run over thousands of
    read sets may rightly take hours: there the deadline is the user's to set)."""
    raw = os.environ.get("PSK_LAUNCH_TIMEOUT", str(default))
    try:
        return float(raw)
    except ValueError:
        sys.exit("PSK_LAUNCH_TIMEOUT=%s: expected a number of seconds" % raw)


def spawn_ranks(argv, world, share_gpu=False, env_extra=None, grace_s=15.0, deadline_s=None, report_s=3.0):
    """Runs `sys.executable argv...` as `world` rank processes; returns the exit code to leave with (0 only when
    every rank returned 0; 124 when the launch ran into its deadline -- deadline_s, default PSK_LAUNCH_TIMEOUT)."""
    if deadline_s is None:
        deadline_s = launch_timeout()
    rdzv = tempfile.mkdtemp(prefix="psk_launch_")          # 0700, unpredictable: the ranks' private meeting place
    nonce = secrets.token_hex(16)
    base = dict(os.environ)
    base.update(env_extra or {})
    base.update({"WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world), "PSK_RDZV_DIR": rdzv, "PSK_LAUNCH_NONCE": nonce,
                 "MASTER_ADDR": base.get("MASTER_ADDR", "127.0.0.1"), "PSK_LAUNCHER": "psk",
                 "PSK_LAUNCH_T0": repr(time.time())})      # the deadline's clock: a leg can tell what is left of it
    base.pop("PSK_RDZV_FILE", None)
    if share_gpu:
        base["PSK_SHARE_GPU"] = "1"
    procs = []
    reaped = set()      # ranks this launcher terminated itself (after another rank's failure): their codes say nothing
    try:
        for r in range(world):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            # ranks > 0 must not write into the result stream: their stdout is this process's stderr
            out = None if r == 0 else sys.stderr
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=out))

        def forward(signum, _frame):
            for p in procs:
                if p.poll() is None:
                    try:
                        p.send_signal(signum)
                    except OSError:
                        pass
        old = {s: signal.signal(s, forward) for s in (signal.SIGINT, signal.SIGTERM)}
        try:
            first_fail = None
            t_start = time.time()
            timed_out = False
            while any(p.poll() is None for p in procs):
                if deadline_s and deadline_s > 0 and not timed_out and time.time() - t_start > deadline_s:
                    timed_out = True
                    live = [r for r, p in enumerate(procs) if p.poll() is None]
                    sys.stderr.write("psk launch: deadline of %.0f s passed (PSK_LAUNCH_TIMEOUT) with rank(s) %s still running; "
                                     "asking them where they are (SIGUSR1 -> phases_rank<r>.json), then terminating them\n"
                                     % (deadline_s, ", ".join(map(str, live))))
                    sys.stderr.flush()
                    for p in procs:
                        if p.poll() is None:
                            try:
                                p.send_signal(signal.SIGUSR1)
                            except OSError:
                                pass
                    time.sleep(report_s)
                    first_fail = time.time() - grace_s - 1.0      # (no further grace: straight to terminate / kill below)
                bad = [p for p in procs if p.poll() not in (None, 0)]
                if bad and first_fail is None:
                    first_fail = time.time()
                    sys.stderr.write("psk launch: rank %d exited with code %d; the other ranks have %.0f s to finish\n"
                                     % (procs.index(bad[0]), bad[0].returncode, grace_s))
                if first_fail is not None and time.time() - first_fail > grace_s:
                    for p in procs:
                        if p.poll() is None:
                            reaped.add(p.pid)
                            p.terminate()
                    t0 = time.time()
                    while any(p.poll() is None for p in procs) and time.time() - t0 < 5.0:
                        time.sleep(0.05)
                    for p in procs:
                        if p.poll() is None:
                            p.kill()
                time.sleep(0.02)
        finally:
            for s, h in old.items():
                signal.signal(s, h)
        codes = [p.wait() for p in procs if p.pid not in reaped]
        if timed_out:
            return 124
        # a rank killed by signal n reports 128 + n, as a shell does; the worst rank decides
        return max([0] + [(c if c > 0 else 128 - c) for c in codes if c != 0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(rdzv, ignore_errors=True)
