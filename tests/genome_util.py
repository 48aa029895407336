"""A seeded synthetic genome with the features real k-mer lists have and uniform keys do not -- tandem repeats,
poly-A runs, a satellite, two diverged copies of one segment -- and eight "samples" of it (SNPs, a deleted
stretch each), plus the canonical k-mer counting glistmaker does (reference src/glistmaker.c:914-924 on the words
of src/sequence.c:116-130: A C G T = 0 1 2 3, first base in the highest bits, the smaller of a word and its
reverse complement), vectorised in numpy.  tests/golden/make_golden_genome.py pins this counting against the
REFERENCE glistmaker's files byte for byte; the GPU tests rebuild the very same lists from the seed alone, so the
fixtures hold digests only."""
import hashlib

import numpy as np

GENOME_SEED = 20251003
N_SAMPLES = 8
BASE_LENGTH = 1_600_000


def make_genome(seed=GENOME_SEED, length=BASE_LENGTH):
    """codes 0..3 of one sequence (uint8)"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=length, dtype=np.uint8)
    # two diverged copies of one 60 kbp segment (3 % substitutions)
    seg = g[100_000:160_000].copy()
    for at in (400_000, 820_000):
        c = seg.copy()
        m = rng.random(len(c)) < 0.03
        c[m] = (c[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) % 4
        g[at:at + len(c)] = c
    # tandem repeats: units of 2 .. 40 bases, 30 .. 400 copies
    for _ in range(60):
        unit = rng.integers(0, 4, size=int(rng.integers(2, 41)), dtype=np.uint8)
        rep = np.tile(unit, int(rng.integers(30, 401)))
        at = int(rng.integers(0, length - len(rep)))
        g[at:at + len(rep)] = rep
    # poly-A / poly-T runs
    for _ in range(80):
        n = int(rng.integers(30, 600))
        at = int(rng.integers(0, length - n))
        g[at:at + n] = 0 if rng.random() < 0.5 else 3
    # a satellite: one 171-base unit, 300 copies, each with a few changes
    unit = rng.integers(0, 4, size=171, dtype=np.uint8)
    sat = np.tile(unit, 300)
    m = rng.random(len(sat)) < 0.01
    sat[m] = (sat[m] + 1) % 4
    g[600_000:600_000 + len(sat)] = sat
    return g


def make_sample(genome, i, seed=GENOME_SEED):
    """sample i: 0.4 % SNPs of its own and one deleted stretch (lists of somewhat different length)"""
    rng = np.random.default_rng(seed + 1000 + i)
    s = genome.copy()
    m = rng.random(len(s)) < 0.004
    s[m] = (s[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) % 4
    cut = int(rng.integers(20_000, 120_000))
    at = int(rng.integers(0, len(s) - cut))
    return np.concatenate([s[:at], s[at + cut:]])


def fasta_text(codes, name):
    s = np.array(list("ACGT"), dtype="U1")[codes]
    txt = "".join(s.tolist())
    lines = [">" + name]
    lines += [txt[j:j + 70] for j in range(0, len(txt), 70)]
    return "\n".join(lines) + "\n"


def canonical_kmers(codes, k):
    """all canonical k-mer words of one sequence, in sequence order (uint64)"""
    n = len(codes) - k + 1
    c = codes.astype(np.uint64)
    w = np.zeros(n, dtype=np.uint64)
    rc = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        w |= c[j:j + n] << np.uint64(2 * (k - 1 - j))
        rc |= (np.uint64(3) - c[j:j + n]) << np.uint64(2 * j)
    return np.minimum(w, rc)


def kmer_list(codes, k):
    """(keys ascending, counts): what glistmaker writes for the sequence"""
    keys, counts = np.unique(canonical_kmers(codes, k), return_counts=True)
    return keys.astype(np.uint64), counts.astype(np.uint32)


def sample_lists(k, n_samples=N_SAMPLES):
    g = make_genome()
    return [kmer_list(make_sample(g, i), k) for i in range(n_samples)]


def sha(data):
    return hashlib.sha256(data).hexdigest()
