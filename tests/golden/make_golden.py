#!/usr/bin/env python3
"""Generate tests/golden/ fixtures by running the REFERENCE binaries built in oracle/_ref.

Run in the build container (needs /root/reference to have been compiled by
`make -C oracle ref`):

    python tests/golden/make_golden.py

Inputs are made by this repo's own seeded generator; expected outputs (files, stdout,
stderr, exit status) are whatever oracle/_ref/glistcompare and oracle/_ref/ref_setops
produce.  Writes:

    tests/golden/inputs.npz     named input record arrays (+ word length, + header flavour)
    tests/golden/cases.json     one entry per invocation: argv, exit code, stdout, stderr,
                                names of the files it created
    tests/golden/outputs.npz    raw bytes of every created file, keyed "<case>/<file>"

The fixtures are data only (inputs + the reference's outputs); no reference source text.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from genometester4_amd.listio import make_records, write_list, write_list_v40  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "glistcompare")
REF_SETOPS = os.path.join(ROOT, "oracle", "_ref", "ref_setops")

RULE_ARGS = ["default", "add", "subtract", "min", "max", "first", "second", "3"]


def gen_pair(rng, k, n_universe, p_a, p_b, max_count=8):
    limit = (1 << (2 * k)) if k < 32 else (1 << 64)
    keys = np.unique(rng.integers(0, limit, size=n_universe, dtype=np.uint64))
    in_a = rng.random(len(keys)) < p_a
    in_b = rng.random(len(keys)) < p_b
    ka, kb = keys[in_a], keys[in_b]
    ca = rng.integers(1, max_count + 1, size=len(ka), dtype=np.uint32)
    cb = rng.integers(1, max_count + 1, size=len(kb), dtype=np.uint32)
    return make_records(ka, ca), make_records(kb, cb)


def build_inputs():
    rng = np.random.default_rng(20240917)
    inp = {}
    a, b = gen_pair(rng, 8, 90, 0.6, 0.6)
    inp["A8"], inp["B8"] = (a, 8, "v42"), (b, 8, "v42")
    # four lists for the N-way sweep, overlapping universe
    keys = np.unique(rng.integers(0, 1 << 16, size=70, dtype=np.uint64))
    for j in range(4):
        m = rng.random(len(keys)) < 0.55
        inp["M%d" % j] = (make_records(keys[m], rng.integers(1, 6, size=int(m.sum()), dtype=np.uint32)), 8, "v42")
    empty = make_records([], [])
    inp["E8"] = (empty, 8, "v42")
    inp["E8b"] = (empty, 8, "v42")
    # disjoint / identical
    inp["D1"] = (make_records([1, 5, 9, 13], [1, 2, 3, 4]), 8, "v42")
    inp["D2"] = (make_records([2, 6, 10, 14, 15], [4, 3, 2, 1, 9]), 8, "v42")
    # k=32: full 64-bit keys incl. >= 2^63 and the all-ones key
    hi = np.array([3, 1 << 62, (1 << 63) + 5, 0xFFFFFFFFFFFFFFF0, 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
    inp["H1"] = (make_records(hi, [1, 2, 3, 4, 5]), 32, "v42")
    inp["H2"] = (make_records(hi[[1, 2, 4]], [7, 1, 2]), 32, "v42")
    inp["H3"] = (make_records(hi[[0, 4]], [2, 2]), 32, "v42")
    # count wrap-around and zero counts
    inp["W1"] = (make_records([10, 20, 30, 40], [0xFFFFFFFF, 0xFFFFFFFF, 5, 0]), 8, "v42")
    inp["W2"] = (make_records([10, 20, 30, 40, 50], [2, 1, 0, 0, 0]), 8, "v42")
    inp["W3"] = (make_records([10, 20, 40, 50], [0xFFFFFFFF, 0xFFFFFFFF, 3, 1]), 8, "v42")
    # version 4.0 file (40-byte header)
    inp["V40"] = (a, 8, "v40")
    # different word length (error transcript)
    inp["K9"] = (make_records([1, 2, 3], [1, 1, 1]), 9, "v42")
    # a larger ragged pair: sizes differ 10x, long runs from one side
    big_keys = np.unique(rng.integers(0, 1 << 24, size=3000, dtype=np.uint64))
    m_a = rng.random(len(big_keys)) < 0.9
    m_b = rng.random(len(big_keys)) < 0.08
    inp["R1"] = (make_records(big_keys[m_a], rng.integers(1, 9, size=int(m_a.sum()), dtype=np.uint32)), 12, "v42")
    inp["R2"] = (make_records(big_keys[m_b], rng.integers(1, 9, size=int(m_b.sum()), dtype=np.uint32)), 12, "v42")
    return inp


def materialise(inp, d):
    for name, (rec, k, flavour) in inp.items():
        p = os.path.join(d, name + ".list")
        (write_list_v40 if flavour == "v40" else write_list)(p, rec, k)


def run_case(cases, outputs, cid, argv, workdir, binary=REF):
    before = set(os.listdir(workdir))
    p = subprocess.run([binary] + argv, cwd=workdir, capture_output=True)
    created = sorted(set(os.listdir(workdir)) - before)
    for f in created:
        with open(os.path.join(workdir, f), "rb") as fh:
            outputs["%s/%s" % (cid, f)] = np.frombuffer(fh.read(), dtype=np.uint8)
        os.remove(os.path.join(workdir, f))
    cases.append(dict(id=cid, tool=os.path.basename(binary), argv=argv, exit=p.returncode,
                      stdout=p.stdout.decode("latin-1"), stderr=p.stderr.decode("latin-1"), files=created))


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    inp = build_inputs()
    work = tempfile.mkdtemp(prefix="gt4golden_")
    materialise(inp, work)
    cases, outputs = [], {}
    # --- two-file sweep: 8 rules x 5 cutoffs x {-d, -du}, all four outputs per run
    for r in RULE_ARGS:
        for c in (0, 1, 2, 3, 4):
            for dflag in ("-d", "-du"):
                cid = "pair_r%s_c%d_%s" % (r, c, dflag.strip("-"))
                run_case(cases, outputs, cid, ["A8.list", "B8.list", "-u", "-i", dflag, "-dd", "-r", r, "-c", str(c), "-o", "g"], work)
    # single-op runs (rule validity checks differ when -i is absent)
    for flags in (["-u"], ["-i"], ["-d"], ["-dd"], ["-du"], ["-u", "-r", "min"], ["-u", "-r", "subtract"], ["-d", "-r", "subtract"], []):
        cid = "pair_only_" + ("_".join(x.strip("-") for x in flags) or "none")
        run_case(cases, outputs, cid, ["A8.list", "B8.list"] + flags + ["-o", "s"], work)
    # count_only / print_operation transcripts
    run_case(cases, outputs, "pair_count_only", ["A8.list", "B8.list", "-u", "-i", "-d", "-dd", "--count_only"], work)
    run_case(cases, outputs, "pair_count_only_c3", ["A8.list", "B8.list", "-u", "-d", "-c", "3", "--count_only"], work)
    run_case(cases, outputs, "pair_print_operation", ["A8.list", "B8.list", "-u", "-i", "--print_operation", "-o", "po"], work)
    run_case(cases, outputs, "pair_stream", ["A8.list", "B8.list", "-u", "-i", "-d", "-dd", "--stream", "-o", "st"], work)
    run_case(cases, outputs, "pair_noscouts", ["A8.list", "B8.list", "-u", "--disable_scouts", "-o", "ns"], work)
    # edge inputs
    for cid, x, y in (("empty_empty", "E8", "E8b"), ("empty_nonempty", "E8", "B8"), ("nonempty_empty", "A8", "E8"),
                      ("disjoint", "D1", "D2"), ("identical", "A8", "A8"), ("k32_allones", "H1", "H2"),
                      ("wrap", "W1", "W2"), ("v40_input", "V40", "B8"), ("ragged", "R1", "R2"), ("ragged_rev", "R2", "R1")):
        run_case(cases, outputs, "edge_" + cid, [x + ".list", y + ".list", "-u", "-i", "-d", "-dd", "-o", "e"], work)
    for c in (0, 2):
        for r in ("add", "min", "max", "subtract"):
            run_case(cases, outputs, "edge_wrap_r%s_c%d" % (r, c), ["W1.list", "W2.list", "-u", "-i", "-d", "-dd", "-r", r, "-c", str(c), "-o", "w"], work)
    run_case(cases, outputs, "edge_ragged_c3_du", ["R1.list", "R2.list", "-u", "-i", "-du", "-dd", "-c", "3", "-o", "e"], work)
    # --- N-way sweep: 8 rules x 5 cutoffs x {union, intersect}, N = 4
    multi = ["M0.list", "M1.list", "M2.list", "M3.list"]
    for r in RULE_ARGS:
        for c in (0, 1, 2, 5, 9):
            for op in ("-u", "-i"):
                run_case(cases, outputs, "multi_r%s_c%d_%s" % (r, c, op.strip("-")), multi + [op, "-r", r, "-c", str(c), "-o", "m"], work)
    run_case(cases, outputs, "multi_ui", multi + ["-u", "-i", "-o", "m"], work)
    run_case(cases, outputs, "multi_count_only", multi + ["-u", "-i", "--count_only"], work)
    run_case(cases, outputs, "multi_with_empty_u", ["M0.list", "E8.list", "M1.list", "-u", "-o", "m"], work)
    run_case(cases, outputs, "multi_with_empty_i", ["M0.list", "E8.list", "M1.list", "-i", "-o", "m"], work)
    run_case(cases, outputs, "multi_wrap", ["W1.list", "W2.list", "W3.list", "-u", "-i", "-o", "m"], work)
    run_case(cases, outputs, "multi_wrap_c0", ["W1.list", "W2.list", "W3.list", "-u", "-i", "-c", "0", "-o", "m"], work)
    run_case(cases, outputs, "multi_wrap_add_i", ["W1.list", "W2.list", "W3.list", "-i", "-r", "add", "-c", "0", "-o", "m"], work)
    run_case(cases, outputs, "multi_k32", ["H1.list", "H2.list", "H3.list", "-u", "-i", "-o", "m"], work)
    run_case(cases, outputs, "multi_diff_rejected", multi + ["-d"], work)
    # pairwise tree (MakeUnion.pl shape, scripts/MakeUnion.pl:31-95) vs one N-way call
    run_case(cases, outputs, "tree_01", ["M0.list", "M1.list", "-u", "-o", "t01"], work)
    # error transcripts
    run_case(cases, outputs, "err_wordlength", ["A8.list", "K9.list", "-u"], work)
    run_case(cases, outputs, "err_one_file", ["A8.list", "-u"], work)
    run_case(cases, outputs, "err_unknown_flag", ["A8.list", "B8.list", "--bogus"], work)
    run_case(cases, outputs, "version", ["-v"], work)
    run_case(cases, outputs, "help", ["-h"], work)
    # --- exported set-operations entry points through this repo's driver (oracle/ref_setops_driver.c)
    run_case(cases, outputs, "setops_write_union_c1", ["write_union", "1", "wu.list"] + multi, work, REF_SETOPS)
    run_case(cases, outputs, "setops_write_union_c3", ["write_union", "3", "wu.list"] + multi, work, REF_SETOPS)
    run_case(cases, outputs, "setops_write_union_wrap", ["write_union", "1", "wu.list", "W1.list", "W2.list", "W3.list"], work, REF_SETOPS)
    run_case(cases, outputs, "setops_union", ["union"] + multi, work, REF_SETOPS)
    run_case(cases, outputs, "setops_union_k32", ["union", "H1.list", "H2.list", "H3.list"], work, REF_SETOPS)
    run_case(cases, outputs, "setops_union_stop5", ["union_stop", "5"] + multi, work, REF_SETOPS)
    run_case(cases, outputs, "setops_is_union", ["is_union", "M0.list", "M1.list", "M2.list"], work, REF_SETOPS)
    run_case(cases, outputs, "setops_union_pair", ["union", "A8.list", "B8.list"], work, REF_SETOPS)

    np.savez_compressed(os.path.join(HERE, "inputs.npz"),
                        **{n: r for n, (r, _, _) in inp.items()},
                        **{"__meta__": np.frombuffer(json.dumps({n: [k, f] for n, (_, k, f) in inp.items()}).encode(), dtype=np.uint8)})
    np.savez_compressed(os.path.join(HERE, "outputs.npz"), **outputs)
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(cases, f, indent=0)
    shutil.rmtree(work)
    print("wrote %d cases, %d output files" % (len(cases), len(outputs)))


if __name__ == "__main__":
    main()
