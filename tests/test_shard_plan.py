"""CPU-only checks of the chunk planner behind `glistcompare --gpus N` / GT4HIP_HBM_LIMIT
(genometester4_amd/csrc/gt4_shard.c, make_plan), through tests/harness/shard_plan_harness.c: the
cuts partition every input exactly, cut every input at the same KEYS (a chunk of one list never
shares a key with another chunk of another list), the chunk count is a multiple of the worker count,
and chunks respect the budget whenever the key distribution allows it."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from genometester4_amd.listio import make_records, write_list

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness():
    d = tempfile.mkdtemp(prefix="gt4plan_")
    exe = os.path.join(d, "shard_plan_harness")
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "genometester4_amd", "csrc"),
                           os.path.join(ROOT, "tests", "harness", "shard_plan_harness.c"), os.path.join(ROOT, "genometester4_amd", "csrc", "gt4_listfile.c"),
                           "-o", exe, "-lpthread"])
    yield d, exe
    import shutil
    shutil.rmtree(d, ignore_errors=True)


def _plan(exe, cwd, ranks, limit, mode, ops, files):
    out = subprocess.run([exe, str(ranks), str(limit), str(mode), str(ops)] + files, cwd=cwd, capture_output=True, check=True).stdout.decode().split("\n")
    c = int(out[0].split()[1])
    cuts = [[int(x) for x in line.split()] for line in out[1:1 + len(files)]]
    return c, cuts


@pytest.mark.parametrize("ranks", [1, 2, 3, 8])
@pytest.mark.parametrize("limit", [1 << 10, 1 << 16, 1 << 40])
def test_cuts_partition_the_inputs_at_common_keys(harness, ranks, limit):
    d, exe = harness
    rng = np.random.default_rng(ranks * 7 + (limit % 1000))
    keys = np.unique(rng.integers(0, 1 << 40, size=20000, dtype=np.uint64))
    lists = []
    for j, frac in enumerate((0.9, 0.3, 0.02, 0.0)):
        m = rng.random(len(keys)) < frac
        rec = make_records(keys[m], np.ones(int(m.sum()), np.uint32))
        write_list(os.path.join(d, "p%d.list" % j), rec, 20)
        lists.append(rec)
    files = ["p%d.list" % j for j in range(4)]
    for mode, ops, use in ((0, 3, files[:2]), (0, 15, [files[2], files[0]]), (1, 0, files), (2, 0, files[:3])):
        c, cuts = _plan(exe, d, ranks, limit, mode, ops, use)
        assert c >= ranks and c % ranks == 0
        recs = [lists[int(f[1])] for f in use]
        for rec, cut in zip(recs, cuts):
            assert len(cut) == c + 1 and cut[0] == 0 and cut[-1] == len(rec)
            assert all(a <= b for a, b in zip(cut, cut[1:]))
        # chunk boundaries are KEYS: every record of chunk c in any list is below every record of chunk c + 1 in any list
        for ch in range(c - 1):
            hi = [int(rec["key"][cut[ch + 1] - 1]) for rec, cut in zip(recs, cuts) if cut[ch + 1] > 0]
            lo = [int(rec["key"][cut[ch + 1]]) for rec, cut in zip(recs, cuts) if cut[ch + 1] < len(rec)]
            if hi and lo:
                assert max(hi) < min(lo)
        # budget: bytes in flight per input record as the planner counts them
        n_streams = 1 if mode else bin(ops).count("1")
        per_record = 12 * (2 + 2 * n_streams + (0 if mode == 0 else 2))
        worst = max(sum(cut[ch + 1] - cut[ch] for cut in cuts) for ch in range(c))
        longest = max(len(r) for r in recs)
        assert worst * per_record <= limit or c >= 2 * longest, (worst, c)
