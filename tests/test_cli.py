"""The drop-in `glistcompare` CLI (genometester4_amd/glistcompare) against the reference's
transcripts and output files in tests/golden.

CPU part: everything the CLI decides before it touches the GPU (argv grammar, validation order,
messages, exit codes, -v/-h) and that it fails loudly without a device.  GPU part: every golden
invocation replayed through the real binary -- exit code, stdout, stderr and every output file
byte-identical to what the reference wrote."""
import os
import subprocess
import tempfile

import pytest

import golden_util as G
from genometester4_amd.listio import write_list, write_list_v40

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "genometester4_amd", "glistcompare")
SETOPS = os.path.join(ROOT, "genometester4_amd", "setops_driver")
CASES, INPUTS, OUTPUTS = G.load()
GL_CASES = [c for c in CASES if c["tool"] == "glistcompare"]
SETOPS_CASES = [c for c in CASES if c["tool"] == "ref_setops"]
# decided before any device work
NO_GPU_IDS = {"version", "help", "err_wordlength", "err_one_file", "err_unknown_flag", "multi_diff_rejected",
              "pair_only_u_r_min", "pair_only_u_r_subtract"}
NO_GPU_IDS |= {c["id"] for c in GL_CASES if c["id"].startswith("multi_r") and c["exit"] == 1 and "Invalid rule" not in c["stderr"]}


@pytest.fixture(scope="module")
def workdir():
    d = tempfile.mkdtemp(prefix="gt4cli_")
    for name, (rec, k, flavour) in INPUTS.items():
        (write_list_v40 if flavour == "v40" else write_list)(os.path.join(d, name + ".list"), rec, k)
    yield d
    import shutil
    shutil.rmtree(d, ignore_errors=True)


def _run(binary, argv, cwd):
    before = set(os.listdir(cwd))
    p = subprocess.run([binary] + argv, cwd=cwd, capture_output=True, timeout=300)
    created = sorted(set(os.listdir(cwd)) - before)
    data = {}
    for f in created:
        with open(os.path.join(cwd, f), "rb") as fh:
            data[f] = fh.read()
        os.remove(os.path.join(cwd, f))
    return p.returncode, p.stdout.decode("latin-1"), p.stderr.decode("latin-1"), data


def _check(case, binary, workdir):
    rc, out, err, files = _run(binary, case["argv"], workdir)
    assert rc == case["exit"], (rc, err)
    assert out == case["stdout"]
    assert err == case["stderr"]
    assert sorted(files) == sorted(case["files"])
    for name, data in files.items():
        assert data == bytes(OUTPUTS["%s/%s" % (case["id"], name)]), "%s differs from the reference output" % name


def test_cli_binary_exists():
    assert os.access(CLI, os.X_OK), "build it: make -C genometester4_amd/csrc"


@pytest.mark.parametrize("case", [c for c in GL_CASES if c["id"] in NO_GPU_IDS], ids=lambda c: c["id"])
def test_cli_host_logic_matches_reference(case, workdir):
    _check(case, CLI, workdir)


def test_cli_fails_loudly_without_gpu(workdir):
    from genometester4_amd import capi
    if capi.lib().gt4hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    rc, out, err, files = _run(CLI, ["A8.list", "B8.list", "-u"], workdir)
    assert rc == 1 and not files and "no HIP device" in err


def test_cli_missing_file_is_an_error_not_a_crash(workdir):
    rc, out, err, files = _run(CLI, ["A8.list", "nope.list", "-u"], workdir)
    assert rc == 1 and "Error: Cannot open nope.list" in err and "Stopping..." in err


def test_cli_count_cutoff_alias_parses(workdir):
    # BASELINE.json writes --count_cutoff; the reference flag is -c/--cutoff.  Both must parse.
    rc, out, err, _ = _run(CLI, ["A8.list", "--count_cutoff", "x3"], workdir)
    assert rc == 1 and "Invalid frequency cut-off: x3" in err


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in GL_CASES if c["id"] not in NO_GPU_IDS], ids=lambda c: c["id"])
def test_cli_reproduces_reference_run(case, workdir):
    _check(case, CLI, workdir)


@pytest.mark.gpu
@pytest.mark.parametrize("case", SETOPS_CASES, ids=lambda c: c["id"])
def test_set_operations_entry_points_match_reference(case, workdir):
    """gt4_write_union / gt4_union / gt4_is_union (include/gt4_set_operations.h) through
    examples/setops_driver.c vs the reference's set-operations.c through oracle/ref_setops_driver.c."""
    _check(case, SETOPS, workdir)


@pytest.mark.gpu
@pytest.mark.parametrize("case", SETOPS_CASES, ids=lambda c: c["id"])
def test_set_operations_over_file_backed_handles_match_reference(case, workdir):
    """The same goldens with GT4HIP_HBM_LIMIT = 1 KiB: every list file is larger than the resident share, so the
    handles stay file-backed -- gt4_write_union streams them through the device in key-range chunks (the command-line
    tool's pipeline, csrc/gt4_shard.c) into the caller's descriptor; gt4_union / gt4_is_union upload on first use."""
    before = set(os.listdir(workdir))
    p = subprocess.run([SETOPS] + case["argv"], cwd=workdir, capture_output=True, timeout=300, env=dict(os.environ, GT4HIP_HBM_LIMIT="1K"))
    created = sorted(set(os.listdir(workdir)) - before)
    files = {}
    for f in created:
        with open(os.path.join(workdir, f), "rb") as fh:
            files[f] = fh.read()
        os.remove(os.path.join(workdir, f))
    assert p.returncode == case["exit"], (p.returncode, p.stderr.decode("latin-1"))
    assert p.stdout.decode("latin-1") == case["stdout"]
    assert sorted(files) == sorted(case["files"])
    for name, data in files.items():
        assert data == bytes(OUTPUTS["%s/%s" % (case["id"], name)]), "%s differs from the reference output" % name


@pytest.mark.gpu
@pytest.mark.parametrize("block", ["1", "3", "50"])
@pytest.mark.parametrize("case", [c for c in SETOPS_CASES if c["argv"][0] in ("union", "is_union", "union_stop")], ids=lambda c: c["id"])
def test_walks_in_key_range_blocks_match_reference(case, block, workdir):
    """gt4_union / gt4_is_union walk long lists in key-range blocks (a table per block, so that a callback that stops
    the walk stops the device work too); GT4HIP_WALK_BLOCK forces blocks of 1, 3 and 50 records of the pacing list on
    the goldens: same callbacks in the same order, the exhausted-list quirk and the early stop included."""
    p = subprocess.run([SETOPS] + case["argv"], cwd=workdir, capture_output=True, timeout=300, env=dict(os.environ, GT4HIP_WALK_BLOCK=block))
    assert p.returncode == case["exit"], (p.returncode, p.stderr.decode("latin-1"))
    assert p.stdout.decode("latin-1") == case["stdout"]


@pytest.mark.gpu
def test_write_union_streams_32_lists_beyond_the_resident_share(tmp_path):
    """glistmaker's collation width (reference src/glistmaker.c:787-835: up to 32 temporary lists into gt4_write_union):
    32 lists of 2e5 records with a 1 MiB resident share -- about a hundred chunks -- against the oracle's loop
    (reference src/set-operations.c:77-116), byte for byte, for two cutoffs; and the count-only form (ofile = 0)."""
    import numpy as np
    import oracle_lib as O
    from genometester4_amd.listio import make_records, header_bytes
    rng = np.random.default_rng(5)
    universe = np.unique(rng.integers(0, 1 << 44, size=1_500_000, dtype=np.uint64))
    lists, names = [], []
    for j in range(32):
        m = rng.random(len(universe)) < 0.15
        rec = make_records(universe[m], rng.integers(1, 9, size=int(m.sum()), dtype=np.uint32))
        lists.append(rec)
        names.append("t%02d.list" % j)
        write_list(os.path.join(tmp_path, names[-1]), rec, 22)
    for cutoff in (1, 3):
        rc_o, n_o, t_o, r_o = O.union_multi(lists, cutoff, 1, 1)  # rule ADD, cutoff on the sum
        p = subprocess.run([SETOPS, "write_union", str(cutoff), "out.list"] + names, cwd=tmp_path, capture_output=True, timeout=600,
                           env=dict(os.environ, GT4HIP_HBM_LIMIT="1M"))
        assert p.returncode == 0, p.stderr.decode("latin-1")
        assert p.stdout.decode() == "NUnique\t%d\nNTotal\t%d\nresult\t0\n" % (n_o, t_o)
        with open(os.path.join(tmp_path, "out.list"), "rb") as fh:
            got = fh.read()
        assert got[:48] == header_bytes(22, n_o, t_o) and got[48:] == r_o.tobytes()
        os.remove(os.path.join(tmp_path, "out.list"))


@pytest.mark.gpu
def test_cli_can_check_that_inputs_are_sorted(workdir):
    import numpy as np
    rec, k, _ = INPUTS["A8"]
    bad = rec.copy()
    bad[[3, 4]] = bad[[4, 3]]
    write_list(os.path.join(workdir, "unsorted.list"), bad, k)
    env = dict(os.environ, GT4HIP_CHECK_SORTED="1")
    p = subprocess.run([CLI, "A8.list", "unsorted.list", "-u", "--count_only"], cwd=workdir, capture_output=True, env=env)
    assert p.returncode == 1 and b"unsorted.list is not sorted" in p.stderr
    p = subprocess.run([CLI, "A8.list", "B8.list", "-u", "--count_only"], cwd=workdir, capture_output=True, env=env)
    assert p.returncode == 0
    os.remove(os.path.join(workdir, "unsorted.list"))
