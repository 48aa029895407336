"""GT4I index files as glistcompare inputs (reference src/glistcompare.c:269-270,
src/index-map.c:123-175).  Fixtures: index and list files the reference's own glistmaker built from
synthetic sequences, and what the reference glistcompare made of them
(tests/golden/make_golden_index.py).

CPU: the oracle's index decode + set operations reproduce every reference output, and a list made
by plain glistmaker from the same sequences equals the decoded index.  GPU: the drop-in CLI replays
every invocation byte for byte, and the C ABI upload decodes like the oracle."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from genometester4_amd.listio import RECORD_DTYPE

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLI = os.path.join(ROOT, "genometester4_amd", "glistcompare")
with open(os.path.join(HERE, "golden", "index_cases.json")) as _f:
    CASES = json.load(_f)
INPUTS = np.load(os.path.join(HERE, "golden", "index_inputs.npz"))
OUTPUTS = np.load(os.path.join(HERE, "golden", "index_outputs.npz"))


def _records(name):
    data = bytes(INPUTS[name])
    if name.endswith(".index"):
        k, _, rec = O.index_decode(data)
        return rec, k
    k = int(np.frombuffer(data[12:16], dtype=np.uint32)[0])
    start = int(np.frombuffer(data[32:40], dtype=np.uint64)[0])
    return np.frombuffer(data[start:], dtype=RECORD_DTYPE), k


def _expected(case):
    p = G.parse_argv(case["argv"])
    recs, ks = zip(*[_records(f) for f in p["files"]])
    k, out, stats = ks[0], {}, []
    if len(recs) == 2:
        res = O.compare(recs[0], recs[1], p["ops"], p["rule"], p["cutoff"], p["subtract"], p["count_override"])
        for bit, (n, total, r) in sorted(res.items()):
            out["%s_%d_%s.list" % (p["out"], k, G.OP_FILES[bit])] = G.list_file_bytes(k, n, total, r)
            stats.append((n, total))
    else:
        for bit, fn, suffix in ((1, O.union_multi, "union"), (2, O.intersect_multi, "intrsec")):
            if p["ops"] & bit:
                rc, n, total, r = fn(list(recs), p["cutoff"], p["rule"], p["count_override"])
                assert rc == 0
                out["%s_%d_%s.list" % (p["out"], k, suffix)] = G.list_file_bytes(k, n, total, r)
                stats.append((n, total))
    return out, stats


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["id"])
def test_oracle_reproduces_reference_on_index_inputs(case):
    exp, stats = _expected(case)
    if "--count_only" in case["argv"]:
        assert "".join("NUnique\t%d\nNTotal\t%d\n" % s for s in stats) == case["stdout"]
        return
    assert sorted(exp) == sorted(case["files"])
    for name, data in exp.items():
        assert data == bytes(OUTPUTS["%s/%s" % (case["id"], name)]), name


def test_index_decodes_to_the_list_of_the_same_sequences():
    """glistmaker --index and plain glistmaker over the same FASTA: same k-mers, count = locations."""
    for stem in ("a", "b"):
        idx, k = _records("I%s_6.index" % stem)
        lst, k2 = _records("L%s_6.list" % stem)
        assert k == k2 == 6 and idx.tobytes() == lst.tobytes()
    with pytest.raises(ValueError):
        O.index_decode(bytes(INPUTS["La_6.list"]))      # a list is not an index


@pytest.fixture(scope="module")
def workdir():
    d = tempfile.mkdtemp(prefix="gt4idx_")
    for name in INPUTS.files:
        with open(os.path.join(d, name), "wb") as f:
            f.write(bytes(INPUTS[name]))
    yield d
    import shutil
    shutil.rmtree(d, ignore_errors=True)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["plain", "chunks", "gpus2"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: c["id"])
def test_cli_reproduces_reference_on_index_inputs(case, workdir, mode):
    """plain: whole tables in HBM; chunks / gpus2: key-range slices of the index tables (the count of
    a slice's last entry reaches to the NEXT entry's first location) through the chunk pipeline /
    two worker processes."""
    env = dict(os.environ, **{"plain": {}, "chunks": {"GT4HIP_HBM_LIMIT": "2K"}, "gpus2": {"GT4HIP_GPUS": "2", "GT4HIP_HBM_LIMIT": "8K"}}[mode])
    before = set(os.listdir(workdir))
    p = subprocess.run([CLI] + case["argv"], cwd=workdir, capture_output=True, timeout=300, env=env)
    created = sorted(set(os.listdir(workdir)) - before)
    try:
        assert p.returncode == case["exit"], p.stderr.decode("latin-1")
        assert p.stdout.decode("latin-1") == case["stdout"]
        assert p.stderr.decode("latin-1") == case["stderr"]
        assert created == sorted(case["files"])
        for f in created:
            with open(os.path.join(workdir, f), "rb") as fh:
                assert fh.read() == bytes(OUTPUTS["%s/%s" % (case["id"], f)]), f
    finally:
        for f in created:
            os.remove(os.path.join(workdir, f))


@pytest.mark.gpu
def test_capi_upload_index_decodes_like_the_oracle():
    from genometester4_amd import capi
    ctx = capi.Context(0)
    try:
        for name in ("Ia_6.index", "Ib_6.index", "Ic_6.index"):
            data = bytes(INPUTS[name])
            k, nloc, rec = O.index_decode(data)
            hdr = np.frombuffer(data[:72], dtype=np.uint8)
            kmers_start = int(hdr[56:64].view(np.uint64)[0])
            kmers = np.frombuffer(data[kmers_start: kmers_start + 16 * len(rec)], dtype=np.uint64).reshape(-1, 2)
            d = ctx.upload_index(kmers, nloc, k)
            assert d.n_words == len(rec) and d.download().tobytes() == rec.tobytes()
            assert d.sum_counts() == nloc and d.is_sorted()
        empty = ctx.upload_index(np.zeros((0, 2), dtype=np.uint64), 0, 6)
        assert empty.n_words == 0
    finally:
        ctx.close()
