"""A slice (12 + 10 seconds: the driver's GPU suite has a wall clock) of the randomised parity soaks in the driver-run
suite: the pair kernels (all rules, cutoffs, output combinations, both single- and two-pass;
tools/soak.py) and the N-way tile kernel (modes, bucket paths, split tiles; tools/soak_nway.py), every case against the
CPU oracle.  `python tools/soak.py <seconds> <seed>` runs them for as long as one likes.
Reference: src/glistcompare.c:433-489, :545-591, :605-717, :843-905."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _soak(tool, seconds, seed, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seconds), str(seed)], capture_output=True, text=True,
                       timeout=seconds + 240, cwd=ROOT, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert "soak ok" in p.stdout or "all equal to the oracle" in p.stdout, p.stdout[-1500:]
    return p.stdout.strip().splitlines()[-1]


def test_pair_kernels_soak():
    print(_soak("soak.py", 12, 501))


def test_nway_kernels_soak():
    print(_soak("soak_nway.py", 10, 502))

