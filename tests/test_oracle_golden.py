"""The CPU oracle (oracle/gt4_oracle.c) against the reference's own outputs (tests/golden).

These are the pins that make the oracle trustworthy: every file the reference binary wrote
for the committed sweep must be reproduced byte for byte.
"""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

CASES, INPUTS, OUTPUTS = G.load()
FILE_CASES = [c for c in CASES if c["tool"] == "glistcompare" and c["exit"] == 0 and c["files"]]
COUNT_CASES = [c for c in CASES if c["tool"] == "glistcompare" and c["exit"] == 0 and "--count_only" in c["argv"]]


def _expected_files(case):
    p = G.parse_argv(case["argv"])
    lists = [G.input_records(INPUTS, f) for f in p["files"]]
    k = lists[0][1]
    recs = [r for r, _ in lists]
    out = {}
    stats = []
    if len(recs) == 2:
        res = O.compare(recs[0], recs[1], p["ops"], p["rule"], p["cutoff"], p["subtract"], p["count_override"])
        for bit, (n, total, r) in sorted(res.items()):
            out["%s_%d_%s.list" % (p["out"], k, G.OP_FILES[bit])] = G.list_file_bytes(k, n, total, r)
            stats.append((n, total))
    else:
        if p["ops"] & 1:
            rc, n, total, r = O.union_multi(recs, p["cutoff"], p["rule"], p["count_override"])
            assert rc == 0
            out["%s_%d_union.list" % (p["out"], k)] = G.list_file_bytes(k, n, total, r)
            stats.append((n, total))
        if p["ops"] & 2:
            rc, n, total, r = O.intersect_multi(recs, p["cutoff"], p["rule"], p["count_override"])
            assert rc == 0
            out["%s_%d_intrsec.list" % (p["out"], k)] = G.list_file_bytes(k, n, total, r)
            stats.append((n, total))
    return out, stats


@pytest.mark.parametrize("case", FILE_CASES, ids=[c["id"] for c in FILE_CASES])
def test_oracle_reproduces_reference_files(case):
    exp, _ = _expected_files(case)
    assert sorted(exp) == sorted(case["files"])
    for name, data in exp.items():
        ref = bytes(OUTPUTS["%s/%s" % (case["id"], name)])
        assert data == ref, "%s differs from the reference output" % name


@pytest.mark.parametrize("case", COUNT_CASES, ids=[c["id"] for c in COUNT_CASES])
def test_oracle_reproduces_count_only_stdout(case):
    _, stats = _expected_files(case)
    text = "".join("NUnique\t%d\nNTotal\t%d\n" % s for s in stats)
    assert text == case["stdout"]


def test_multi_rule_rejection_matches_reference():
    # reference: union_multi/intersect_multi return 1 on rules they do not support -> exit 1, no file
    rej = [c for c in CASES if c["id"].startswith("multi_r") and c["exit"] == 1]
    assert rej
    recs = [INPUTS["M%d" % j][0] for j in range(4)]
    seen_kernel_reject = 0
    for c in rej:
        p = G.parse_argv(c["argv"])
        if "Invalid rule" in c["stderr"]:
            fn = O.union_multi if p["ops"] & 1 else O.intersect_multi
            assert fn(recs, p["cutoff"], p["rule"], p["count_override"])[0] == 1
            seen_kernel_reject += 1
        assert c["files"] == []
    assert seen_kernel_reject > 0


def _rows(stdout):
    lines = stdout.strip().split("\n")
    assert lines[-1].startswith("result\t")
    return int(lines[-1].split("\t")[1]), [tuple(int(x) for x in ln.split("\t")) for ln in lines[:-1]]


@pytest.mark.parametrize("cid", ["setops_union", "setops_union_k32", "setops_union_pair", "setops_union_stop5", "setops_is_union"])
def test_oracle_walks_match_reference(cid):
    case = next(c for c in CASES if c["id"] == cid)
    argv = case["argv"]
    stop = int(argv[1]) if argv[0] == "union_stop" else 0
    files = [a for a in argv if a.endswith(".list")]
    recs = [G.input_records(INPUTS, f)[0] for f in files]
    fn = O.is_union_walk if argv[0] == "is_union" else O.union_walk
    r, rows = fn(recs, stop)
    ref_r, ref_rows = _rows(case["stdout"])
    assert r == ref_r
    assert rows == ref_rows


@pytest.mark.parametrize("cid", ["setops_write_union_c1", "setops_write_union_c3", "setops_write_union_wrap"])
def test_oracle_write_union_matches_reference(cid):
    case = next(c for c in CASES if c["id"] == cid)
    argv = case["argv"]
    files = [a for a in argv[3:]]
    recs = [G.input_records(INPUTS, f)[0] for f in files]
    k = G.input_records(INPUTS, files[0])[1]
    rc, n, total, r = O.write_union(recs, int(argv[1]))
    assert rc == 0
    assert G.list_file_bytes(k, n, total, r) == bytes(OUTPUTS["%s/%s" % (cid, argv[2])])
    assert case["stdout"].startswith("NUnique\t%d\nNTotal\t%d\n" % (n, total))


def test_header_parse_v40_and_v42():
    from genometester4_amd.listio import parse_header
    import struct
    v40 = struct.pack("<IIIIQQQ", 0x47543443, 4, 0, 8, 3, 6, 0) + b"\0" * 36
    h = parse_header(v40)
    assert h["list_start"] == 40 and h["word_bytes"] == 8 and h["count_bytes"] == 4
    v42 = struct.pack("<IIIIQQQII", 0x47543443, 4, 2, 8, 3, 6, 48, 0, 0)
    h = parse_header(v42)
    assert h["list_start"] == 48 and h["word_bytes"] == 8
    with pytest.raises(ValueError):
        parse_header(b"XXXX" + v42[4:])


def test_tree_union_equals_multi_union_at_default_cutoff():
    # a MakeUnion.pl pairwise tree == union_multi at cutoff 1 with counts >= 1
    recs = [INPUTS["M%d" % j][0] for j in range(4)]
    u01 = O.compare(recs[0], recs[1], 1)[1][2]
    u23 = O.compare(recs[2], recs[3], 1)[1][2]
    tree = O.compare(u01, u23, 1)[1][2]
    rc, n, total, multi = O.union_multi(recs)
    assert rc == 0 and np.array_equal(tree, multi)
