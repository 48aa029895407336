"""Multi-process CPU coverage (gloo, world_size 2 and 3) of the key-range sharded N-way path."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from genometester4_amd import distributed as D

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_nway_gloo(world):
    port = _free_port()
    res = tempfile.mktemp(prefix="gt4dist_")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GT4_DIST_RESULT=res, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert open(res).read() == "OK"
    os.remove(res)


def test_key_ranges_and_slices():
    for k, world in ((16, 8), (25, 8), (32, 8), (32, 3)):
        b = D.key_range_bounds(k, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == 1 << 64
    keys = np.array([0, 5, 1 << 62, (1 << 63) + 1, 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
    sl = D.slice_indices(keys, D.key_range_bounds(32, 4))
    assert sl == [(0, 2), (2, 3), (3, 4), (4, 5)]
    # slices partition the list whatever the key distribution
    rng = np.random.default_rng(0)
    keys = np.unique(rng.integers(0, 1 << 50, size=5000, dtype=np.uint64))
    sl = D.slice_indices(keys, D.key_range_bounds(25, 8))
    assert sl[0][0] == 0 and sl[-1][1] == len(keys) and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
