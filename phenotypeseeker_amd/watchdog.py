"""A rank that hangs says where.

The first run on eight GPUs is the likeliest place for eight ranks that all wait in a collective: without this the
driver's kill leaves no trace of the phase they were in.  launch.spawn_ranks gives a launch
an overall deadline (PSK_LAUNCH_TIMEOUT) and, when it passes, sends every rank SIGUSR1 before it terminates them by pid;
a rank that has called install() answers the signal by writing `phases_rank<r>.json` -- the table of the phases it has
finished plus `stuck_in`: the phase it is in and, inside it, the call it is blocked in (a collective, ncclCommInitRank).

The answer must not depend on the main thread coming back to the interpreter: a rank blocked inside libpsk.so (a ctypes
call: the GIL is released, but Python-level signal handlers only run between bytecodes OF THE MAIN THREAD) would never
write anything.  So the C-level handler's wakeup descriptor (signal.set_wakeup_fd) is read by a daemon thread, and that
thread writes the table.  Standard library only: importing this module loads nothing that could touch the GPU.

The reference has no counterpart (its workers are a multiprocess Pool without any failure handling, SURVEY.md section 5:
"failure detection: none").
"""
import json
import os
import signal
import sys
import threading
import time

_state = {"phase": None, "phase_since": None, "call": None, "call_since": None, "snapshot": None, "rank": 0, "world": 1,
          "installed": False, "dir": None, "on_dump": None}


def enter(phase):
    """The phase this rank is in from now on (modeling.Phases.enter, bench.py)."""
    _state["phase"], _state["phase_since"] = phase, time.time()


def swap_call(call):
    """Names the blocking call this rank is about to make (None: it has returned); returns the name it replaces."""
    prev = _state["call"]
    _state["call"], _state["call_since"] = call, (time.time() if call else None)
    return prev


class blocking:
    """with watchdog.blocking("all-reduce (rccl)"): ..."""

    def __init__(self, call):
        self.call = call

    def __enter__(self):
        self.prev = swap_call(self.call)

    def __exit__(self, *exc):
        swap_call(self.prev)
        return False


def report():
    """What the table of a stuck rank says."""
    now = time.time()
    rec = {"rank": _state["rank"], "world": _state["world"], "stuck": True, "pid": os.getpid(),
           "stuck_in": _state["phase"], "stuck_in_for_s": round(now - _state["phase_since"], 3) if _state["phase_since"] else None,
           "blocked_in_call": _state["call"],
           "blocked_for_s": round(now - _state["call_since"], 3) if _state["call_since"] else None}
    snap = _state["snapshot"]
    if snap is not None:
        try:
            rec.update(snap())
        except Exception as e:          # a half-updated table must not cost the report
            rec["snapshot_error"] = "%s: %s" % (type(e).__name__, e)
    return rec


def dump():
    rec = report()
    path = os.path.join(_state["dir"] or ".", "phases_rank%d.json" % rec["rank"])
    tmp = "%s.%d.tmp" % (path, os.getpid())
    with open(tmp, "w") as f:
        json.dump(rec, f)
    os.replace(tmp, path)
    sys.stderr.write("psk rank %d/%d: stuck in phase `%s`%s -- table in %s\n" % (
        rec["rank"], rec["world"], rec["stuck_in"],
        (" (blocked in %s for %.1f s)" % (rec["blocked_in_call"], rec["blocked_for_s"])) if rec["blocked_in_call"] else "", path))
    sys.stderr.flush()
    hook = _state["on_dump"]
    if hook is not None:
        try:
            hook(rec)
        except Exception as e:
            sys.stderr.write("psk rank %d: the stuck-phase hook failed: %s\n" % (rec["rank"], e))
    return rec


def on_dump(hook):
    """hook(table) runs after a stuck-phase table has been written (bench.py: rank 0 prints its result line once more)."""
    _state["on_dump"] = hook


def install(rank, world, snapshot=None, directory=None):
    """Main thread only (signal.set_wakeup_fd demands it).  snapshot: () -> dict merged into the table (the finished
    phases); directory: where phases_rank<r>.json goes (default: the working directory at the time of the signal)."""
    _state.update(rank=int(rank), world=int(world), snapshot=snapshot, dir=directory)
    if _state["installed"]:
        return
    r, w = os.pipe()
    os.set_blocking(w, False)
    # a Python-level handler has to exist for the C-level one (which writes the signal number to the descriptor) to be
    # installed at all; it has nothing to do
    signal.signal(signal.SIGUSR1, lambda signum, frame: None)
    signal.set_wakeup_fd(w, warn_on_full_buffer=False)

    def listen():
        while True:
            try:
                data = os.read(r, 64)
            except OSError:
                return
            if not data:
                return
            if signal.SIGUSR1 in data:          # (the descriptor is told about EVERY signal with a Python handler)
                try:
                    dump()
                except Exception as e:
                    sys.stderr.write("psk rank %d: could not write the stuck-phase table: %s\n" % (_state["rank"], e))
    threading.Thread(target=listen, name="psk-watchdog", daemon=True).start()
    _state["installed"] = True


def self_deadline(seconds, code=124):
    """For ranks that somebody else's launcher started (torchrun, srun: no launch.spawn_ranks above them to keep the deadline):
    after `seconds` this rank writes its stuck-phase table itself and leaves with `code` -- the other ranks' timers, set to the
    same deadline, do the same within the skew of their starts; a launcher that tears the job down on the first exit finds
    the tables already written.  0 or negative: no deadline."""
    if not seconds or seconds <= 0:
        return

    def run():
        time.sleep(seconds)
        try:
            sys.stderr.write("psk rank %d/%d: deadline of %.0f s passed (PSK_LAUNCH_TIMEOUT)\n" % (_state["rank"], _state["world"], seconds))
            dump()
        finally:
            os._exit(code)
    threading.Thread(target=run, name="psk-deadline", daemon=True).start()
