"""Real-shaped inputs: eight lists of 1.2e6 canonical k-mers of a synthetic genome with
tandem repeats, poly-A runs, a satellite and diverged copies (tests/genome_util.py, rebuilt from the seed; pinned
against the reference glistmaker by tests/golden/make_golden_genome.py) through

  * the CPU oracle (not gpu): every file the REFERENCE glistcompare wrote is reproduced byte for byte (sha256);
  * the GPU (gpu): the drop-in CLI writes the same bytes, transcripts and exit codes; the library's N-way calls
    give the same records whichever tile kernel and whichever path (the library's choice, the tile kernel forced,
    the pairwise tree) takes them -- and the test records which one the library chose for these keys.

Reference: src/glistmaker.c:914-924, src/glistcompare.c:545-591, :605-717, :843-905."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import genome_util as GU
import oracle_lib as O
from genometester4_amd.listio import header_bytes, make_records, write_list

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "genometester4_amd", "glistcompare")
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "genome_golden.json")))
RULE = {"default": 0, "add": 1, "max": 4}


@pytest.fixture(scope="module", params=[16, 25])
def lists(request):
    k = request.param
    assert (GOLDEN["seed"], GOLDEN["samples"], GOLDEN["base_length"]) == (GU.GENOME_SEED, GU.N_SAMPLES, GU.BASE_LENGTH)
    recs = [make_records(keys, counts) for keys, counts in GU.sample_lists(k)]
    for rec, g in zip(recs, GOLDEN["k"][str(k)]["inputs"]):
        data = header_bytes(k, len(rec), int(rec["count"].astype(np.uint64).sum())) + rec.tobytes()
        assert GU.sha(data) == g["sha256"], "the rebuilt %s is not the file the reference glistmaker wrote" % g["file"]
    return k, recs


def _parse(argv):
    """(inputs, ops, rule, cutoff) of a golden invocation"""
    files = [a for a in argv if a.endswith(".list")]
    ops = (1 if "-u" in argv else 0) | (2 if "-i" in argv else 0) | (4 if "-d" in argv else 0)
    rule = RULE[argv[argv.index("-r") + 1]] if "-r" in argv else 0
    cutoff = int(argv[argv.index("-c") + 1]) if "-c" in argv else 1
    return files, ops, rule, cutoff


def _file_image(k, rec, n, total):
    return header_bytes(k, n, total) + np.ascontiguousarray(rec).tobytes()


def test_oracle_reproduces_the_reference_files(lists):
    k, recs = lists
    for run in GOLDEN["k"][str(k)]["runs"]:
        files, ops, rule, cutoff = _parse(run["argv"])
        ins = [recs[int(f[1:f.index("_")])] for f in files]
        assert run["exit"] == 0
        if len(ins) == 2:
            res = O.compare(ins[0], ins[1], ops, rule=rule, cutoff=cutoff)
            made = {"union": res.get(1), "intrsec": res.get(2), "0_diff1": res.get(4)}
        else:
            rc, n, t, r = (O.union_multi if ops & 1 else O.intersect_multi)(ins, cutoff, rule, 1)
            assert rc == 0
            made = {"union" if ops & 1 else "intrsec": (n, t, r)}
        for name, g in run["files"].items():
            kind = name[len("g_%d_" % k):-len(".list")]
            n, t, r = made[kind]
            assert (n, t) == (g["n_words"], g["total_count"]), (run["id"], name)
            assert GU.sha(_file_image(k, r, n, t)) == g["sha256"], "%s: %s differs from the reference's file" % (run["id"], name)


@pytest.fixture(scope="module")
def workdir(lists):
    k, recs = lists
    d = tempfile.mkdtemp(prefix="gt4genome_")
    for i, rec in enumerate(recs):
        write_list(os.path.join(d, "s%d_%d.list" % (i, k)), rec, k)
    yield d
    import shutil
    shutil.rmtree(d, ignore_errors=True)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"GT4HIP_HBM_LIMIT": "4M"}], ids=["one piece", "key-range chunks"])
def test_cli_writes_the_reference_bytes(lists, workdir, env):
    k, _ = lists
    for run in GOLDEN["k"][str(k)]["runs"]:
        before = set(os.listdir(workdir))
        p = subprocess.run([CLI] + run["argv"], cwd=workdir, capture_output=True, timeout=600, env=dict(os.environ, **env))
        created = sorted(set(os.listdir(workdir)) - before)
        try:
            assert p.returncode == run["exit"], p.stderr.decode("latin-1")
            assert p.stdout.decode("latin-1") == run["stdout"] and p.stderr.decode("latin-1") == run["stderr"]
            assert created == sorted(run["files"])
            for name in created:
                data = open(os.path.join(workdir, name), "rb").read()
                assert GU.sha(data) == run["files"][name]["sha256"], "%s: %s differs from the reference's file" % (run["id"], name)
        finally:
            for name in created:
                os.remove(os.path.join(workdir, name))


@pytest.mark.gpu
def test_library_paths_on_genomic_keys(lists):
    """the eight-way unions through the library: what it chooses for these keys (recorded in the test's output),
    the tile kernel forced (both tile kernels) and the pairwise tree -- all the reference's records"""
    from genometester4_amd import capi
    k, recs = lists
    ctx = capi.Context(0)
    try:
        dev = [ctx.upload(r, k) for r in recs]
        for run in GOLDEN["k"][str(k)]["runs"]:
            files, ops, rule, cutoff = _parse(run["argv"])
            if len(files) != 8 or not ops & 1:
                continue
            g = run["files"]["g_%d_union.list" % k]
            for name, opts in (("library's choice", {"kway": 1}), ("tile kernel", {"kway": 3}), ("tree", {"kway": 0})):
                for o, v in opts.items():
                    ctx.set_option(o, v)
                declined = ctx.get_counter("kway_declined")
                rc, n, t, out = ctx.union_multi(dev, cutoff, rule, 1)
                assert rc == 0 and (n, t) == (g["n_words"], g["total_count"]), (run["id"], name)
                assert GU.sha(_file_image(k, out.download(), n, t)) == g["sha256"], "%s by %s differs from the reference's file" % (run["id"], name)
                out.free()
                if name == "library's choice":
                    print("%s k=%d: one pass %d, declined %d, tiles cut in two %d" % (run["id"], k, ctx.get_counter("nway_one_pass"), ctx.get_counter("kway_declined") - declined, ctx.get_counter("kway_splits")))
        for d in dev:
            d.free()
    finally:
        ctx.close()
