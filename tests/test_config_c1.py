"""BASELINE.json configs[0] / SURVEY 8d C1: `glistcompare --intersection` on two 10 M-entry synthetic
k=16 .list files through the reference's own CPU path -- plumbing, no GPU: the reference binary
(oracle/_ref, built from /root/reference by oracle/Makefile) and the C restatement must agree on
NUnique / NTotal and on every byte of the output file at this size too (the committed goldens are
tiny).  With a GPU, the drop-in CLI is held to the same file."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import gpu_util as U
import oracle_lib as O
from genometester4_amd.listio import header_bytes, write_list

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "genometester4_amd", "glistcompare")
N, K = 10_000_000, 16


@pytest.fixture(scope="module")
def c1():
    # shared universe: 1.5e7 ascending unique keys < 4^16; every third key is in both
    # lists, the others alternate between A and B: |A| = |B| = 1e7, |A n B| = 5e6
    rng = np.random.default_rng(16)
    idx = np.arange(3 * N // 2, dtype=np.uint64)
    keys = idx * np.uint64(200) + rng.integers(0, 200, size=len(idx), dtype=np.uint64)
    in_a, in_b = idx % 3 != 2, idx % 3 != 1
    a = U.make_records(keys[in_a], rng.integers(1, 9, size=int(in_a.sum()), dtype=np.uint32))
    b = U.make_records(keys[in_b], rng.integers(1, 9, size=int(in_b.sum()), dtype=np.uint32))
    d = tempfile.mkdtemp(prefix="gt4c1_")
    write_list(os.path.join(d, "a.list"), a, K)
    write_list(os.path.join(d, "b.list"), b, K)
    exp = O.compare(a, b, O.OP_INTRSEC)[O.OP_INTRSEC]
    yield d, exp
    import shutil
    shutil.rmtree(d, ignore_errors=True)


def _run(exe, d, args):
    return subprocess.run([exe, "a.list", "b.list"] + args, cwd=d, capture_output=True, timeout=600)


def test_c1_reference_cpu_path_matches_restatement(c1):
    d, (n, total, rec) = c1
    if not os.access(O.REF_GLISTCOMPARE, os.X_OK):
        pytest.skip("oracle/_ref/glistcompare not built (needs /root/reference)")
    assert n == N // 2
    r = _run(O.REF_GLISTCOMPARE, d, ["--intersection", "--count_only"])
    assert r.returncode == 0 and r.stdout.decode() == "NUnique\t%d\nNTotal\t%d\n" % (n, total)
    r = _run(O.REF_GLISTCOMPARE, d, ["--intersection", "-o", "ref"])
    assert r.returncode == 0
    with open(os.path.join(d, "ref_%d_intrsec.list" % K), "rb") as f:
        assert f.read() == header_bytes(K, n, total) + rec.tobytes()


@pytest.mark.gpu
def test_c1_drop_in_cli_writes_the_same_file(c1):
    d, (n, total, rec) = c1
    r = _run(CLI, d, ["--intersection", "-o", "hip"])
    assert r.returncode == 0, r.stderr.decode()
    with open(os.path.join(d, "hip_%d_intrsec.list" % K), "rb") as f:
        assert f.read() == header_bytes(K, n, total) + rec.tobytes()
