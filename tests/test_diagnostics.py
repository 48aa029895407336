"""The reader's failure diagnostics, --stream runs and -D debug output of the drop-in CLI against
transcripts of the REFERENCE binary (tests/golden/diag_cases.json, made by make_golden_diag.py).

Malformed inputs are refused before any device work, so those cases run without a GPU.  One line of
the reference's stderr is not reproduced: its az_object assertion macro prints the path of its own
source file (`File /.../az/object.c line 115 (?): Assertion obj != NULL failed`); the messages in
front of and behind it are compared exactly.  A malformed FIRST file makes the reference crash
(NULL dereference, src/glistcompare.c:275-279); here it is the same message and exit code as for a
malformed second file.

Reference: src/word-map.c:181-215, src/index-map.c:317-373, src/word-list-stream.c:127-186,
src/glistcompare.c:224-225, :809-812, :914."""
import json
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "genometester4_amd", "glistcompare")
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = json.load(open(os.path.join(GOLDEN, "diag_cases.json")))
FILES = np.load(os.path.join(GOLDEN, "diag_files.npz"))
ASSERT_LINE = re.compile(r"^File .* line \d+ \(\?\): Assertion .* failed\n", re.M)
ERROR_CASES = [c for c in CASES if c["exit"] != 0]
OK_CASES = [c for c in CASES if c["exit"] == 0]


@pytest.fixture(scope="module")
def workdir():
    d = tempfile.mkdtemp(prefix="gt4diag_")
    for k in FILES.files:
        if k.startswith("in/"):
            with open(os.path.join(d, k[3:]), "wb") as f:
                f.write(bytes(FILES[k]))
    yield d
    import shutil
    shutil.rmtree(d, ignore_errors=True)


def _run(argv, cwd, env=None):
    before = set(os.listdir(cwd))
    p = subprocess.run([CLI] + argv, cwd=cwd, capture_output=True, timeout=300, env=env)
    created = sorted(set(os.listdir(cwd)) - before)
    data = {}
    for f in created:
        data[f] = open(os.path.join(cwd, f), "rb").read()
        os.remove(os.path.join(cwd, f))
    return p.returncode, p.stdout.decode("latin-1"), p.stderr.decode("latin-1"), data


def _check(case, cwd, env=None):
    rc, out, err, files = _run(case["argv"], cwd, env)
    assert rc == case["exit"], (rc, err)
    assert out == case["stdout"]
    assert err == ASSERT_LINE.sub("", case["stderr"])
    assert sorted(files) == sorted(case["files"])
    for name, data in files.items():
        assert data == bytes(FILES["out/%s/%s" % (case["id"], name)]), "%s differs from the reference output" % name


@pytest.mark.parametrize("case", ERROR_CASES, ids=lambda c: c["id"])
def test_malformed_inputs_are_refused_with_the_reference_messages(case, workdir):
    _check(case, workdir)


def test_malformed_first_file_is_an_error_not_a_crash(workdir):
    for bad, msg in (("badtag.list", "Error: File badtag.list has unknown format"),
                     ("major5.list", "gt4_word_map_new: incompatible major version 5 (required 4)"),
                     ("trunc.list", "gt4_word_map_new: file size too small (115, should be at least 660)"),
                     ("major5.index", "gt4_index_map_new: incompatible major version 5 (required 4)")):
        rc, out, err, files = _run([bad, "B8.list", "-u"], workdir)
        assert rc == 1 and not files
        assert err.startswith(msg + "\n") or (msg + "\n") in err
        assert err.endswith("Stopping...\n")


def _hostile_cases():
    import hostile_headers
    return hostile_headers.cases()


@pytest.mark.parametrize("position", ["first", "second", "third of three"])
@pytest.mark.parametrize("name,blob,message", _hostile_cases(), ids=[c[0] for c in _hostile_cases()])
def test_hostile_headers_are_refused_without_a_signal(name, blob, message, position, workdir):
    """Headers that pass (or wrap) the reference's size test while the readers stride 12 -- the reference segfaults on
    several (`/root/reference/src/word-map.c:211-215`); here: the size diagnostic, exit code 1, nothing written
    (Missing 4; the same files run under ASan + UBSan in tests/test_host_sanitizers.py)."""
    import hostile_headers
    path = os.path.join(workdir, name)
    with open(path, "wb") as f:
        f.write(blob)
    with open(os.path.join(workdir, "h_good.list"), "wb") as f:
        f.write(hostile_headers.good_list())
    try:
        argv = {"first": [name, "h_good.list", "-u"], "second": ["h_good.list", name, "-u"],
                "third of three": ["h_good.list", "h_good.list", name, "-u"]}[position]
        rc, out, err, files = _run(argv + ["-o", "hostile"], workdir)
        assert rc == 1, (rc, err)  # (a signal would be a negative return code)
        assert message in err and "Error: File %s is invalid or corrupted\n" % name in err, err
        assert err.endswith("Stopping...\n") and not files and out == ""
    finally:
        os.remove(path)
        os.remove(os.path.join(workdir, "h_good.list"))


@pytest.mark.gpu
def test_hostile_headers_are_refused_on_the_gpu_box_too(workdir):
    """the same fifteen files where a device IS there (the driver's suite): refused before any device work, nothing
    written, and a well-formed pair in the same directory still runs"""
    import hostile_headers
    with open(os.path.join(workdir, "h_good.list"), "wb") as f:
        f.write(hostile_headers.good_list())
    try:
        for name, blob, message in _hostile_cases():
            with open(os.path.join(workdir, name), "wb") as f:
                f.write(blob)
            try:
                rc, out, err, files = _run(["h_good.list", name, "-u", "-i", "-o", "hostile"], workdir)
                assert rc == 1 and message in err and not files, (name, rc, err)
            finally:
                os.remove(os.path.join(workdir, name))
        rc, out, err, files = _run(["h_good.list", "h_good.list", "-i", "--count_only"], workdir)
        assert rc == 0 and "NUnique\t8" in out, (rc, out, err)
    finally:
        os.remove(os.path.join(workdir, "h_good.list"))


@pytest.mark.gpu
@pytest.mark.parametrize("case", OK_CASES, ids=lambda c: c["id"])
def test_stream_debug_and_header_variants_match_reference(case, workdir):
    _check(case, workdir)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in OK_CASES if c["id"].startswith(("stream_", "diag_minor"))], ids=lambda c: c["id"])
def test_stream_and_header_variants_through_the_chunked_path(case, workdir):
    _check(case, workdir, dict(os.environ, GT4HIP_HBM_LIMIT="2K"))
