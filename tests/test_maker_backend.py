"""glistmaker's table step on the device: gt4hip_words_to_list = radix sort of packed
k-mer words + folding of equal words into (word, occurrences) records, then gt4_write_union-style
collation of several such lists.

Pinned to the REFERENCE glistmaker: tests/golden/maker_fixture.npz holds, for four word lengths,
the canonical words of a small FASTA text (shuffled) and the .list file the reference's glistmaker
wrote for that text (make_golden_maker.py); the device step must give that file's header totals and
records.  Larger random cases are checked against numpy's sort / unique.

Reference: src/word-table.c:217-260, src/utils.c:127-198, src/glistmaker.c:914-924, :333, :814."""
import os

import numpy as np
import pytest

from genometester4_amd.listio import RECORD_DTYPE, parse_header

pytestmark = pytest.mark.gpu

FIX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maker_fixture.npz"))


@pytest.fixture(scope="module")
def ctx():
    from genometester4_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("k", [5, 11, 25, 32])
def test_words_to_list_reproduces_the_reference_glistmaker_list(ctx, k):
    words = FIX["words_%d" % k]
    ref = bytes(FIX["list_%d" % k])
    h = parse_header(ref)
    lst = ctx.words_to_list(words, k)
    assert lst.n_words == h["n_words"] and lst.word_length == k
    assert lst.sum_counts() == h["total_count"] == len(words)
    assert lst.download().tobytes() == ref[h["list_start"]:]
    # the words in four arbitrary parts, each made into a list, then collated by the N-way ADD union
    # with cutoff 1 (gt4_write_union, src/set-operations.c:40-129): the same list again
    parts = np.array_split(words, 4)
    lists = [ctx.words_to_list(p, k) for p in parts]
    rc, n, total, out = ctx.union_multi(lists, 1, 1, 1)
    assert rc == 0 and (n, total) == (h["n_words"], h["total_count"])
    assert out.download().tobytes() == ref[h["list_start"]:]


@pytest.mark.parametrize("n,k", [(1, 16), (2, 16), (255, 3), (2048, 8), (2049, 20), (100003, 13), (3_000_000, 25), (1_500_000, 32),
                                 (50_000, 26), (100_003, 27), (100_003, 29), (200_000, 31), (9000, 5)])  # 9-bit digits: some, all, all but one pass
def test_sort_and_fold_against_numpy(ctx, n, k):
    rng = np.random.default_rng(n + k)
    space = (1 << 64) if k == 32 else (1 << (2 * k))
    # a skewed draw: many repeats of few words beside a uniform bulk
    bulk = rng.integers(0, min(space, 1 << 63), size=n, dtype=np.uint64)
    if k == 32:
        bulk = bulk * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    hot = rng.integers(0, min(space, 1 << 63), size=5, dtype=np.uint64)
    pick = rng.random(n) < 0.3
    words = np.where(pick, hot[rng.integers(0, 5, size=n)], bulk).astype(np.uint64)
    u, c = np.unique(words, return_counts=True)
    lst = ctx.words_to_list(words, k)
    got = lst.download()
    assert len(got) == len(u)
    assert got["key"].tobytes() == u.tobytes()
    assert got["count"].tobytes() == c.astype(np.uint32).tobytes()
    assert lst.is_sorted()


@pytest.mark.parametrize("runs", [[70000], [1, 8191, 8192, 8193, 1, 49157, 1, 1], [8192, 8192, 8192], [8191, 1, 8192, 1, 1, 8190, 3], [63, 1, 64, 65, 1023, 1, 1025, 30000, 2]])
def test_fold_runs_across_wavefronts_and_tiles(ctx, runs):
    """Runs of equal words that end exactly at, one before and one behind the 64-word rounds, the
    1024-word stretches of a wavefront and the 8192-word tiles of the fold kernels, runs that cover whole
    tiles (tiles without a start), a single run (wordtable_find_frequencies, src/word-table.c:233-260)."""
    rng = np.random.default_rng(len(runs))
    keys = np.sort(rng.choice(1 << 40, size=len(runs), replace=False).astype(np.uint64))
    words = np.repeat(keys, runs)
    rng.shuffle(words)
    lst = ctx.words_to_list(words, 20)
    got = lst.download()
    assert got["key"].tolist() == keys.tolist()
    assert got["count"].tolist() == runs


def test_empty_input_gives_an_empty_list(ctx):
    lst = ctx.words_to_list(np.zeros(0, dtype=np.uint64), 16)
    assert lst.n_words == 0
