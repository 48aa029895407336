#!/usr/bin/env python3
"""Fixtures for glistmaker's table step: small FASTA texts made here, the canonical
k-mer words extracted from them by this script (the text parsing is out of scope of the GPU path),
and the .list the REFERENCE glistmaker writes for each -- the device's sort + fold of the words must
reproduce that list's records.  Needs oracle/_ref/glistmaker; writes tests/golden/maker_fixture.npz.

Reference: src/glistmaker.c:914-924 (wordtable_sort, wordtable_find_frequencies), :333 / :814
(gt4_write_union); canonical words: the smaller of a word and its reverse complement."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref", "glistmaker")
CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def canonical_words(seqs, k):
    mask = (1 << (2 * k)) - 1
    words = []
    for s in seqs:
        w = rc = n = 0
        for ch in s.upper():
            if ch not in CODE:
                w = rc = n = 0
                continue
            c = CODE[ch]
            w = ((w << 2) | c) & mask
            rc = (rc >> 2) | ((3 - c) << (2 * (k - 1)))
            n += 1
            if n >= k:
                words.append(min(w, rc))
    return np.array(words, dtype=np.uint64)


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    rng = np.random.default_rng(4)
    work = tempfile.mkdtemp(prefix="gt4maker_")
    out = {}
    for k, n_seq, length, repeat in ((11, 3, 5000, 0), (25, 4, 4000, 3), (32, 2, 6000, 2), (5, 2, 3000, 0)):
        seqs = []
        with open(os.path.join(work, "t.fa"), "w") as f:
            for i in range(n_seq):
                s = "".join(rng.choice(list("ACGT"), size=length))
                if i == 1:
                    s = s[:1000] + "N" * 5 + s[1000:] + "acgtacgt" + "T" * 80 + "A" * 90
                for _ in range(repeat):  # repeated stretches: counts above 1 for long words too
                    a = int(rng.integers(0, length - 300))
                    s += s[a:a + 300]
                seqs.append(s)
                f.write(">s%d\n" % i)
                for j in range(0, len(s), 70):
                    f.write(s[j:j + 70] + "\n")
        r = subprocess.run([REF, "t.fa", "-w", str(k), "-o", "out"], cwd=work, capture_output=True)
        assert r.returncode == 0, r.stderr.decode()
        data = open(os.path.join(work, "out_%d.list" % k), "rb").read()
        words = canonical_words(seqs, k)
        rng.shuffle(words)  # the device step must not depend on the order the words arrive in
        out["words_%d" % k] = words
        out["list_%d" % k] = np.frombuffer(data, dtype=np.uint8)
        print("k=%d: %d words -> %d list bytes" % (k, len(words), len(data)))
    np.savez_compressed(os.path.join(HERE, "maker_fixture.npz"), **out)


if __name__ == "__main__":
    main()
