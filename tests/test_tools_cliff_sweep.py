"""CPU: the standing sweep for performance cliffs (tools/cliff_sweep.py) stays runnable and keeps telling the same story on
a handful of pairs: clean reads at the default scoring are not started over; a burst of errors at the reference's scoring (match 1) sends
pairs back to a checkpoint and probation returns them to value steps; every result the sweep looks at is the oracle's (it asserts that)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cliff_sweep.py"), *args], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if l and not l.startswith("#")]


def test_quick_sweep_runs_and_clean_reads_are_not_started_over():
    rows = _run("--quick", "--shapes", "C0", "--pairs-scale", "0.1")
    cells = [l.split() for l in rows[1:]]
    assert len(cells) == 4                                      # two scorings x two error rates
    for c in cells:
        assert c[-1] == "0" or c[-2] == "0", c                  # mismatch column (a FLAT BATCH remark may follow it)
    m2 = [c for c in cells if c[1] == "m2x4q4r2"]
    assert m2 and all(float(c[8]) <= 5.0 for c in m2), m2       # over-clean %: (almost) nobody at the default scoring


def test_bursts_table_shows_pairs_going_back_and_returning():
    rows = _run("--bursts", "--quick", "--shapes", "C1", "--pairs-scale", "0.17")
    cells = [l.replace("|", " ").split() for l in rows[1:]]
    by = {(c[1], int(c[2])): c for c in cells}
    assert float(by[("m2x4q4r2", 350)][4]) == 0.0               # nobody goes back at match 2
    back_off, back_on, ret = float(by[("m1x4q6r2", 350)][4]), float(by[("m1x4q6r2", 350)][8]), float(by[("m1x4q6r2", 350)][9])
    assert back_off >= 50.0 and back_on >= 50.0 and ret >= 30.0, by[("m1x4q6r2", 350)]
    row = by[("m1x4q6r2", 350)]                                  # shape scoring burst pairs | back over cost tail | back ret over cost tail
    assert float(row[11]) < float(row[6])                       # cost with probation < without
    assert float(row[7]) > 10.0 and float(row[12]) < float(row[7])      # the tail one such pair costs its wave: a cliff; smaller with probation
    assert float(by[("m2x4q4r2", 350)][7]) < 3.0                # ... and none at match 2
