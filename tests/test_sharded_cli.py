"""The key-range sharded modes of the drop-in CLI (genometester4_amd/csrc/gt4_shard.c): several
GPUs (one worker process per GPU, forked before any HIP call) and chunks streamed through the
device memory (GT4HIP_HBM_LIMIT).  Every mode must give the reference's bytes.

  chunks   one worker, a budget of 1 KiB: every golden run is cut into many key-range chunks that go
           through the loader / merger / writer pipeline one after the other
  gpus2    two worker processes (both land on device 0 when one GPU is visible), each writing its
           own extents of the output files with pwrite
  rccl1    one worker, the RCCL gatherv path (communicator of one rank: the code the 8-GPU run uses)

Reference: scripts/MakeUnion.pl:31-95, src/glistcompare.c:366-422 (the multi-list job);
src/utils.c:35-99, src/glistcompare.c:491-496 (how the reference reads and writes)."""
import os
import subprocess

import numpy as np
import pytest

import golden_util as G
from test_cli import CLI, GL_CASES, NO_GPU_IDS, OUTPUTS, workdir  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

MODES = {
    "chunks": {"GT4HIP_HBM_LIMIT": "1K"},
    "gpus2": {"GT4HIP_GPUS": "2", "GT4HIP_HBM_LIMIT": "6K"},
    # (NCCL_DEBUG: an environment that asks RCCL for its version banner gets it on stdout; the transcripts
    # are compared with the library quiet -- the tool itself only sets NCCL_DEBUG=NONE when it is unset)
    "rccl1": {"GT4HIP_GPUS": "1", "GT4HIP_GATHER": "rccl", "GT4HIP_HBM_LIMIT": "64K", "NCCL_DEBUG": "NONE"},
}
GPU_CASES = [c for c in GL_CASES if c["id"] not in NO_GPU_IDS]
# the forked / RCCL modes start several HIP contexts per run: a spread of the cases, not all of them
SUBSET = [c for i, c in enumerate(GPU_CASES) if c["id"].startswith(("edge_", "tree_")) or i % 6 == 0]


def _run_env(argv, cwd, env_extra):
    before = set(os.listdir(cwd))
    p = subprocess.run([CLI] + argv, cwd=cwd, capture_output=True, timeout=600, env=dict(os.environ, **env_extra))
    created = sorted(set(os.listdir(cwd)) - before)
    data = {}
    for f in created:
        with open(os.path.join(cwd, f), "rb") as fh:
            data[f] = fh.read()
        os.remove(os.path.join(cwd, f))
    return p.returncode, p.stdout.decode("latin-1"), p.stderr.decode("latin-1"), data


def _check(case, cwd, env_extra):
    rc, out, err, files = _run_env(case["argv"], cwd, env_extra)
    assert rc == case["exit"], (rc, err)
    assert out == case["stdout"]
    assert err == case["stderr"]
    assert sorted(files) == sorted(case["files"])
    for name, data in files.items():
        assert data == bytes(OUTPUTS["%s/%s" % (case["id"], name)]), "%s differs from the reference output" % name


# (every edge / tree / multi case and every third of the others: the whole set runs in one piece in tests/test_cli.py and through
# 1-KiB chunks under the sanitizers in tests/test_host_sanitizers.py; the driver's GPU suite has a wall clock)
@pytest.mark.parametrize("case", [c for i, c in enumerate(GPU_CASES) if c["id"].startswith(("edge_", "tree_", "multi_")) or i % 3 == 0], ids=lambda c: c["id"])
def test_chunked_run_reproduces_reference(case, workdir):
    _check(case, workdir, MODES["chunks"])


@pytest.mark.parametrize("case", SUBSET[::2], ids=lambda c: c["id"])
def test_two_worker_processes_reproduce_reference(case, workdir):
    _check(case, workdir, MODES["gpus2"])


@pytest.mark.parametrize("case", SUBSET[::8], ids=lambda c: c["id"])  # (2.6 s each: librccl.so is half a gigabyte)
def test_rccl_gather_path_reproduces_reference(case, workdir):
    _check(case, workdir, MODES["rccl1"])


def test_gpus_flag_is_the_same_as_the_environment(workdir):
    case = next(c for c in GPU_CASES if c["id"] == "edge_ragged")
    rc, out, err, files = _run_env(case["argv"] + ["--gpus", "2"], workdir, {})
    assert rc == 0 and sorted(files) == sorted(case["files"])
    for name, data in files.items():
        assert data == bytes(OUTPUTS["%s/%s" % (case["id"], name)])


@pytest.fixture(scope="module")
def big_files():
    """Two 3e7-record k=25 lists (360 MB each) generated on the GPU, written to /dev/shm."""
    import shutil
    import tempfile
    from genometester4_amd import capi
    from genometester4_amd.listio import write_list
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(prefix="gt4shard_", dir=base)
    ctx = capi.Context(0)
    n = 30_000_000
    for name, seeds in (("a", (11, 21, 3, 0, 12, 23, 3, 1)), ("b", (11, 22, 3, 0, 13, 24, 3, 2))):
        s, p = ctx.alloc(n // 2, 25), ctx.alloc(n // 2, 25)
        ctx.generate_ex(s, n // 2, seeds[0], seeds[1], 8, seeds[2], seeds[3])
        ctx.generate_ex(p, n // 2, seeds[4], seeds[5], 8, seeds[6], seeds[7])
        u = ctx.compare(s, p, 1)[1][1]
        write_list(os.path.join(d, name + ".list"), u.download(), 25)
    ctx.close()
    yield d
    shutil.rmtree(d, ignore_errors=True)


def _outputs(d, prefix):
    return {f: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d)) if f.startswith(prefix + "_")}


def test_medium_files_every_mode_writes_the_same_bytes(big_files):
    """-u -i -d -dd on 2 x 3e7 records: the plain run (whole lists in HBM), chunks of ~100 MB through
    the pipeline, two worker processes, and the RCCL gather path all write identical files; the plain
    run's intersection is also checked against the oracle."""
    import oracle_lib as O
    from genometester4_amd.listio import read_list
    d = big_files
    runs = {"plain": {}, "chunks": {"GT4HIP_HBM_LIMIT": "1200M"}, "gpus2": {"GT4HIP_GPUS": "2", "GT4HIP_HBM_LIMIT": "900M"},
            "rccl1": {"GT4HIP_GPUS": "1", "GT4HIP_GATHER": "rccl", "GT4HIP_HBM_LIMIT": "1500M"}}
    got = {}
    for name, env in runs.items():
        p = subprocess.run([CLI, "a.list", "b.list", "-u", "-i", "-d", "-dd", "-c", "2", "-o", name], cwd=d, capture_output=True,
                           timeout=900, env=dict(os.environ, GT4HIP_VERBOSE="1", **env))
        assert p.returncode == 0, p.stderr.decode()
        got[name] = _outputs(d, name)
        assert len(got[name]) == 4
        if name != "plain":
            assert b"chunks" in p.stderr
            for f, data in got[name].items():
                assert data == got["plain"][f.replace(name, "plain", 1)], "%s: %s differs from the plain run" % (name, f)
            for f in got[name]:
                os.remove(os.path.join(d, f))
    _, a = read_list(os.path.join(d, "a.list"))
    _, b = read_list(os.path.join(d, "b.list"))
    exp = O.compare(a, b, 2, cutoff=2)[2]
    _, inter = read_list(os.path.join(d, "plain_25_intrsec.list"))
    assert inter.tobytes() == exp[2].tobytes()
    # the N-way forms on the same files (a, b, a again): union and intersection, chunked vs plain
    for flags in (["-u"], ["-i"]):
        outs = []
        for name, env in (("mp", {}), ("mc", {"GT4HIP_HBM_LIMIT": "1500M"}), ("m2", {"GT4HIP_GPUS": "2", "GT4HIP_HBM_LIMIT": "2G"})):
            p = subprocess.run([CLI, "a.list", "b.list", "plain_25_0_diff1.list"] + flags + ["-o", name], cwd=d, capture_output=True,
                               timeout=900, env=dict(os.environ, **env))
            assert p.returncode == 0, p.stderr.decode()
            o = _outputs(d, name)
            assert len(o) == 1
            outs.append(list(o.values())[0])
            for f in o:
                os.remove(os.path.join(d, f))
        assert outs[0] == outs[1] == outs[2]


def test_io_primitives_round_trip(big_files):
    """gt4hip_list_upload_fd / gt4hip_list_write_fd (pinned staging, several copy threads) move a
    list body bit for bit, at any record offset."""
    from genometester4_amd import capi
    d = big_files
    ctx = capi.Context(0)
    try:
        src = os.path.join(d, "a.list")
        n = (os.path.getsize(src) - 48) // 12
        fd = os.open(src, os.O_RDONLY)
        first, count = 1_000_003, n - 2_000_007
        lst = ctx.upload_fd(fd, 48 + 12 * first, count, 25)
        os.close(fd)
        body = np.fromfile(src, dtype=np.uint8, offset=48 + 12 * first, count=12 * count)
        assert lst.n_words == count and lst.is_sorted()
        assert lst.download_range(0, 5000).tobytes() == body[:60000].tobytes()
        out = os.path.join(d, "copy.bin")
        fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        ctx.write_fd(lst, 7, count - 7, fd, 100)
        os.close(fd)
        back = np.fromfile(out, dtype=np.uint8, offset=100)
        assert back.tobytes() == body[84:].tobytes()
        os.remove(out)
    finally:
        ctx.close()


@pytest.mark.parametrize("mode", ["plain", "chunks", "gpus2", "rccl1"])
def test_unwritable_output_is_an_error_in_every_mode(workdir, mode):
    """The reference writes through a NULL FILE* here (src/glistcompare.c:814-834) and crashes; the
    tool reports the file it cannot create, exits 1, leaves no temporary behind -- and, with several
    worker processes, nobody is left waiting at a barrier."""
    env = {"plain": {}}
    env.update(MODES)
    before = set(os.listdir(workdir))
    rc, out, err, files = _run_env(["A8.list", "B8.list", "-u", "-i", "-o", "no_such_dir/x"], workdir, env[mode])
    assert rc == 1, (rc, err)
    assert "Cannot create output file no_such_dir/x_8_union.list.tmp" in err
    assert not files and set(os.listdir(workdir)) == before


def test_three_workers_and_more_workers_than_records(workdir):
    case = next(c for c in GPU_CASES if c["id"] == "edge_ragged")
    for gpus in ("3", "7"):
        _check(case, workdir, {"GT4HIP_GPUS": gpus})
    tiny = next(c for c in GPU_CASES if c["id"] == "edge_empty_nonempty")
    _check(tiny, workdir, {"GT4HIP_GPUS": "5"})


def test_two_rccl_ranks_on_one_device_fail_cleanly(workdir):
    """RCCL refuses two ranks on one device.  On a one-GPU box that makes `--gpus 2` with the RCCL gather
    the one multi-rank communicator set-up that can be driven here: both worker processes get the
    unique id through the shared block and reach ncclCommInitRank, which reports the duplicate device --
    the tool must say so, exit 1 within seconds (nobody left waiting for the other) and leave no
    temporary behind.  (With two or more GPUs the same command succeeds and is covered by _check.)"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_int(0)
    assert hip.hipGetDeviceCount(ctypes.byref(n)) == 0
    env = {"GT4HIP_GPUS": "2", "GT4HIP_GATHER": "rccl", "GT4HIP_HBM_LIMIT": "64K", "NCCL_DEBUG": "NONE"}
    if n.value >= 2:
        _check(next(c for c in GPU_CASES if c["id"] == "edge_ragged"), workdir, env)
        return
    before = set(os.listdir(workdir))
    rc, out, err, files = _run_env(["A8.list", "B8.list", "-u", "-o", "dup"], workdir, env)
    assert rc == 1, (rc, err)
    assert "ncclCommInitRank" in err
    assert not files and set(os.listdir(workdir)) == before
