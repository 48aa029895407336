#!/usr/bin/env python3
"""Generate the GT4I-index fixtures of tests/golden/ with the REFERENCE binaries in oracle/_ref.

Run in the build container after `make -C oracle ref`:

    python tests/golden/make_golden_index.py

A seeded synthetic FASTA pair is indexed by the reference's own `glistmaker --index` (and listed by
plain `glistmaker`); the reference `glistcompare` is then run on index/index, index/list, list/index
and three-input combinations.  Writes:

    tests/golden/index_inputs.npz    raw bytes of every input file (.index and .list)
    tests/golden/index_cases.json    one entry per invocation: argv, exit code, stdout, stderr, created files
    tests/golden/index_outputs.npz   raw bytes of every created file, keyed "<case>/<file>"

Fixtures are data only (files the reference tools produced from synthetic sequences).
"""
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "glistcompare")
MAKER = os.path.join(ROOT, "oracle", "_ref", "glistmaker")


def fasta(path, rnd, seqs):
    with open(path, "w") as f:
        for name, s in seqs:
            f.write(">%s\n" % name)
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")


def main():
    if not (os.path.exists(REF) and os.path.exists(MAKER)):
        sys.exit("build the reference first: make -C oracle ref")
    rnd = random.Random(20241002)
    rs = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))
    work = tempfile.mkdtemp(prefix="gt4gidx_")
    s1, s2, s3 = rs(900), rs(500), rs(700)
    fasta(os.path.join(work, "a.fa"), rnd, [("a1", s1), ("a2", s2)])
    fasta(os.path.join(work, "b.fa"), rnd, [("b1", s3), ("b2", s1[:400] + "N" + s2[100:300])])
    fasta(os.path.join(work, "c.fa"), rnd, [("c1", s2[:250] + s3[:250])])
    k = 6
    for stem in ("a", "b", "c"):
        subprocess.check_call([MAKER, stem + ".fa", "-w", str(k), "--index", "-o", "I" + stem], cwd=work,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.check_call([MAKER, stem + ".fa", "-w", str(k), "-o", "L" + stem], cwd=work,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ia, ib, ic = ("I%s_%d.index" % (x, k) for x in "abc")
    la, lb = ("L%s_%d.list" % (x, k) for x in "ab")
    inputs = {}
    for f in (ia, ib, ic, la, lb):
        inputs[f] = np.frombuffer(open(os.path.join(work, f), "rb").read(), dtype=np.uint8)
    cases, outputs = [], {}

    def run(cid, argv):
        before = set(os.listdir(work))
        p = subprocess.run([REF] + argv, cwd=work, capture_output=True)
        created = sorted(set(os.listdir(work)) - before)
        for f in created:
            outputs["%s/%s" % (cid, f)] = np.frombuffer(open(os.path.join(work, f), "rb").read(), dtype=np.uint8)
            os.remove(os.path.join(work, f))
        cases.append(dict(id=cid, tool="glistcompare", argv=argv, exit=p.returncode, stdout=p.stdout.decode("latin-1"),
                          stderr=p.stderr.decode("latin-1"), files=created))

    run("idx_idx_all", [ia, ib, "-u", "-i", "-d", "-dd", "-o", "x"])
    run("idx_idx_c2_max", [ia, ib, "-u", "-i", "-d", "-dd", "-c", "2", "-r", "max", "-o", "x"])
    run("idx_idx_du", [ia, ib, "-du", "-o", "x"])
    run("idx_list_all", [ia, lb, "-u", "-i", "-d", "-dd", "-o", "x"])
    run("list_idx_all", [la, ib, "-u", "-i", "-d", "-dd", "-o", "x"])
    run("idx_same_list", [ia, la, "-i", "-d", "-dd", "-o", "x"])   # an index and the list of the same sequences
    run("idx_count_only", [ia, ib, "-u", "-i", "-d", "-dd", "--count_only"])
    run("idx3_union", [ia, ib, ic, "-u", "-o", "m"])
    run("idx3_intersect_c2", [ia, lb, ic, "-i", "-c", "2", "-o", "m"])

    np.savez_compressed(os.path.join(HERE, "index_inputs.npz"), **inputs)
    np.savez_compressed(os.path.join(HERE, "index_outputs.npz"), **outputs)
    with open(os.path.join(HERE, "index_cases.json"), "w") as f:
        json.dump(cases, f, indent=0)
    shutil.rmtree(work)
    print("wrote %d cases, %d output files, %d inputs (%d bytes)" % (len(cases), len(outputs), len(inputs),
                                                                        sum(len(v) for v in inputs.values())))


if __name__ == "__main__":
    main()
