"""bench.py's result line cannot be lost to what follows the timed region -- the host-side pieces,
no GPU: the line is printed as soon as `value` exists and again after every leg; a leg's child that outlives its budget is asked
where it is and killed, and the line carries its code and phase; a launch that runs into its deadline re-emits the line."""
import io
import json
import os
import sys
import time

from helpers import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_result_line_is_emitted_early_and_again_after_every_leg(capsys):
    out = {"metric": "k-mer x sample chi2 cells/sec", "value": 1.0e13, "legs": {"cpu_baseline": "pending", "e2e": "pending"}}
    line = bench.ResultLine(0, out)
    line.emit()
    out["legs"]["cpu_baseline"] = "ok"
    out["cpu_baseline"] = {"value": 1.0}
    line.emit()
    # the launch's deadline passes while the e2e leg hangs: the watchdog's hook prints the line once more, with the phase
    line.on_deadline({"stuck_in": "leg: e2e", "blocked_in_call": "all-reduce(max) (rccl, 8 ranks)"})
    got = _lines(capsys.readouterr().out)
    assert len(got) == 3 and all(g["value"] == 1.0e13 for g in got)
    assert got[0]["line"] == "provisional: legs pending: cpu_baseline, e2e" and "cpu_baseline" not in got[0]
    assert got[1]["line"] == "provisional: legs pending: e2e" and got[1]["cpu_baseline"] == {"value": 1.0}
    assert got[2]["aborted"]["stuck_in"] == "leg: e2e" and got[2]["legs"]["e2e"] == "pending"
    # the legs done (one of them failed): the last line says final, and which leg failed
    out["legs"]["e2e"] = "failed: 2 rank(s) failed"
    del out["aborted"]
    line.emit(final=True)
    last = _lines(capsys.readouterr().out)[-1]
    assert last["line"] == "final" and last["legs"]["e2e"].startswith("failed")
    # ranks other than 0 print nothing
    bench.ResultLine(1, out).emit(final=True)
    assert capsys.readouterr().out == ""


def test_the_watchdog_hook_re_emits_the_line_when_a_rank_is_asked_where_it_is(tmp_path, capsys):
    from phenotypeseeker_amd import watchdog
    old = dict(watchdog._state)
    try:
        watchdog._state.update(rank=0, world=2, dir=str(tmp_path), snapshot=lambda: {"total_s": 1.0, "phases_s": {"ingest": 1.0}})
        watchdog.enter("leg: e2e")
        line = bench.ResultLine(0, {"value": 3.0, "legs": {"e2e": "pending"}})
        watchdog.on_dump(line.on_deadline)
        watchdog.dump()
        got = _lines(capsys.readouterr().out)
        assert got[-1]["value"] == 3.0 and got[-1]["aborted"]["stuck_in"] == "leg: e2e"
        with open(tmp_path / "phases_rank0.json") as f:
            assert json.load(f)["stuck_in"] == "leg: e2e"
    finally:
        watchdog._state.clear()
        watchdog._state.update(old)


def test_a_child_over_its_budget_is_asked_where_it_is_then_killed(tmp_path):
    """The e2e leg's children: one that hangs in a collective (tests/_hang_worker.py: the product's Phases + watchdog around an
    all-reduce that never returns) is sent SIGUSR1 after the budget -- its table names the phase and the call --, then
    terminated by pid; rc 124, and bench._read_phases hands the stuck phase to the line."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    t0 = time.time()
    rc, tail = bench._run_child_with_budget([sys.executable, os.path.join(ROOT, "tests", "_hang_worker.py"), "both-hang"],
                                            str(tmp_path), env, 1.5)
    assert rc == 124 and time.time() - t0 < 1.5 + 2.0 + 5.0 + 5.0
    assert "stuck in phase `all-reduce of the union size`" in tail
    ph = bench._read_phases(str(tmp_path), 2)
    assert ph["rank1"] is None
    assert ph["rank0"]["stuck_in"] == "all-reduce of the union size" and ph["rank0"]["blocked_in_call"].startswith("all-reduce(sum)")
    assert list(ph["rank0"]["phases_s"]) == ["arguments, data.pheno", "presence matrix"]
    # a child that ends in time: its own code, no signal
    rc, tail = bench._run_child_with_budget([sys.executable, "-c", "import sys; sys.stderr.write('bye'); sys.exit(3)"], str(tmp_path), env, 30.0)
    assert rc == 3 and tail == "bye"


def test_e2e_budget_leaves_a_minute_of_the_launch_deadline(monkeypatch):
    monkeypatch.setenv("PSK_LAUNCH_TIMEOUT", "900")
    assert 835.0 < bench.e2e_budget_s(time.time() - 2.0) <= 838.0            # 900 - 2 - 60
    assert bench.e2e_budget_s(time.time() - 880.0) < 30.0                    # the leg is skipped then, and the line says so
    monkeypatch.setenv("PSK_LAUNCH_TIMEOUT", "5000")
    assert bench.e2e_budget_s(time.time()) == 900.0                          # never more than 900 s
    monkeypatch.setenv("PSK_LAUNCH_TIMEOUT", "0")
    assert bench.e2e_budget_s(time.time() - 1.0e6) == 900.0                  # no deadline
