"""glistquery's multi-list forms through include/gt4_set_operations.h:
multi-list dump (gt4_union / gt4_is_union), search_lists_multi and the zipper search, printed by
examples/setops_driver.c exactly as the reference's glistquery prints them; expected stdout from
the reference binary (tests/golden/query_cases.json, made by make_golden_query.py).

All of them are merges on the device: the per-key count table is the N-way union plus one streaming
merge per list (no per-key binary search), list membership a second table under rule NUMBER.
Reference: src/glistquery.c:82-106, :702-717, :776-812; src/set-operations.c:131-228."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import golden_util as G
from genometester4_amd.listio import RECORD_DTYPE, write_list

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "genometester4_amd", "setops_driver")
Q = json.load(open(os.path.join(ROOT, "tests", "golden", "query_cases.json")))


@pytest.fixture(scope="module")
def workdir():
    _, inputs, _ = G.load()
    d = tempfile.mkdtemp(prefix="gt4query_")
    for name in Q["inputs"]:
        rec, k, _ = inputs[name]
        write_list(os.path.join(d, name + ".list"), rec, k)
    for name, (hexbytes, k) in Q["extra_inputs"].items():
        write_list(os.path.join(d, name + ".list"), np.frombuffer(bytes.fromhex(hexbytes), dtype=RECORD_DTYPE), k)
    yield d
    import shutil
    shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("case", Q["cases"], ids=lambda c: c["id"])
def test_query_form_prints_what_glistquery_prints(case, workdir):
    p = subprocess.run([DRIVER] + case["driver_argv"], cwd=workdir, capture_output=True, timeout=300)
    assert p.returncode == case["exit"], p.stderr.decode()
    assert p.stdout.decode("latin-1") == case["stdout"]


def test_count_table_by_merge_on_generated_lists():
    """The table of six 3e5-record lists against numpy: keys = sorted union, column j = list j's count or 0."""
    from genometester4_amd import capi
    import gpu_util as U
    ctx = capi.Context(0)
    try:
        rng = np.random.default_rng(3)
        keys = np.unique(rng.integers(0, 1 << 44, size=600000, dtype=np.uint64))
        lists = []
        for j in range(6):
            m = rng.random(len(keys)) < (0.1 + 0.13 * j)
            lists.append(U.make_records(keys[m], rng.integers(0, 9, size=int(m.sum()), dtype=np.uint32)))
        dev = [ctx.upload(x, 22) for x in lists]
        tk, tc = ctx.union_table(dev)
        uni = np.unique(np.concatenate([x["key"] for x in lists]))
        assert tk.tobytes() == uni.tobytes()
        for j, x in enumerate(lists):
            col = np.zeros(len(uni), dtype=np.uint32)
            col[np.searchsorted(uni, x["key"])] = x["count"]
            assert tc[:, j].tobytes() == col.tobytes()
        pk, pc = ctx.union_table(dev, probe=True)
        pk2, pp = ctx.union_table(dev, probe=True, presence=True)
        assert pk.tobytes() == lists[0]["key"].tobytes() == pk2.tobytes()
        for j, x in enumerate(lists):
            idx = np.searchsorted(x["key"], lists[0]["key"])
            idx[idx == len(x)] = 0
            hit = x["key"][idx] == lists[0]["key"]
            assert pp[:, j].tobytes() == hit.astype(np.uint32).tobytes()
            assert pc[:, j].tobytes() == np.where(hit, x["count"][idx], 0).astype(np.uint32).tobytes()
    finally:
        ctx.close()
