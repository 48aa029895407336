"""bench.py's line-keeping machinery without a GPU: the one JSON line is printed once, on the ORIGINAL stdout, whatever
else a library writes to descriptor 1; a leg that overruns its wall-clock bound makes rank 0 print the line as far as the
run got and the process leave with exit code 4 (VERDICT round 5, next 2 a).  The GPU suite drives the same paths through
real runs with 2 and 8 ranks (tests/test_multi_gpu.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, timeout=60):
    p = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + code], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    return p.returncode, p.stdout, p.stderr


def test_the_line_is_printed_once_and_stdout_carries_nothing_else():
    rc, out, err = _run("""
import os, bench
bench.own_stdout()
print("a library's banner on descriptor 1")            # -> stderr
os.write(1, b"and one written by C code\\n")             # -> stderr
assert bench.emit({"metric": "m", "value": 1.5}) is True
assert bench.emit({"metric": "again"}) is False          # whoever comes second prints nothing
""")
    assert rc == 0, err
    assert out.count("\n") == 1 and json.loads(out) == {"metric": "m", "value": 1.5}
    assert "banner" in err and "C code" in err


def test_a_leg_that_overruns_prints_what_was_measured_and_exits_4():
    rc, out, err = _run("""
import time, bench
bench.own_stdout()
bench.PROGRESS["headline"] = {"metric": "m", "value": 3.0, "self_check": "ok"}
bench.PROGRESS["union8"].update({"merge_only": 2.0, "n_gpus": 8})
g = bench.Guard(0)
g.arm("union8", 0.6)
time.sleep(30)                                           # (a collective that never returns)
""")
    assert rc == 4, (rc, err)
    line = json.loads(out)
    assert line["value"] == 3.0 and line["self_check"] == "ok"
    assert line["union8"]["merge_only"] == 2.0 and "wall-clock bound" in line["union8"]["error"] and "union8" in line["union8"]["error"]
    assert "TIMEOUT" in err


def test_a_rank_other_than_zero_leaves_without_a_line_and_a_disarmed_guard_stays_quiet():
    rc, out, err = _run("""
import time, bench
bench.own_stdout()
g = bench.Guard(3)
g.arm("intersect", 0.4)
time.sleep(30)
""")
    assert rc == 4 and out == "" and "rank 3" in err
    rc, out, err = _run("""
import time, bench
g = bench.Guard(0)
g.arm("c2", 0.3)
g.disarm()
time.sleep(1.2)
print("done")
""")
    assert rc == 0 and out.strip() == "done"


def test_partial_line_without_a_headline_says_so():
    rc, out, err = _run("""
import json, bench
print(json.dumps(bench.partial_line("leg 'intersect' exceeded its wall-clock bound on rank 0")))
""")
    assert rc == 0
    line = json.loads(out)
    assert line["value"] is None and "intersect" in line["error"] and line["unit"] == "k-mers/s"
