"""Full-size parity (BASELINE.json configs 1, 2, 3 and 4 at their real sizes) on one MI355X.

No CPU oracle finishes 4e9 records in test time, but every set operation is KEY-LOCAL: the result
for key x depends only on the records with key x.  So besides size-independent identities the GPU
outputs are checked against the CPU oracle on key WINDOWS: a window [lo, hi) is cut out of both
inputs by `lower_bound`, the oracle runs on the two slices, and its output must equal, byte for
byte, the slice of the GPU output between the same two keys.  Windows are placed at the start, around
record index 2^31 (and 2^32 where a list reaches it), in the middle and at the very end.

Reference: compare_wordmaps hot loop, src/glistcompare.c:843-905; predicates :459-489."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

WINDOW = 400_000  # records of A per window (the oracle takes ~10 ms per 10^6 records)


@pytest.fixture()
def ctx():
    from genometester4_amd import capi
    c = capi.Context(0)
    yield c
    c.close()  # returns every list and the pool to the driver before the next full-size case


def build_pair(ctx, n, k, seed_base=0):
    """|A| = |B| = n, |A n B| = n // 2 exactly, from three disjoint residue classes (as bench.py)."""
    from genometester4_amd import capi
    n_s, n_p = n // 2, n - n // 2
    s, p = ctx.alloc(n_s, k), ctx.alloc(n_p, k)
    ctx.generate_ex(s, n_s, seed_base + 11, seed_base + 21, 8, 3, 0)
    ctx.generate_ex(p, n_p, seed_base + 12, seed_base + 23, 8, 3, 1)
    a = ctx.compare(s, p, capi.OP_UNION)[1][capi.OP_UNION]
    ctx.generate_ex(s, n_s, seed_base + 11, seed_base + 22, 8, 3, 0)
    ctx.generate_ex(p, n_p, seed_base + 13, seed_base + 24, 8, 3, 2)
    b = ctx.compare(s, p, capi.OP_UNION)[1][capi.OP_UNION]
    s.free()
    p.free()
    assert a.n_words == n and b.n_words == n
    return a, b


def window_starts(n):
    """Record indices of A at which windows start: both ends, the middle, and the 32-bit edges."""
    pts = [0, n // 2, max(0, n - WINDOW)]
    for edge in (1 << 31, 1 << 32):
        if edge + WINDOW < n:
            pts.append(edge - WINDOW // 2)
    return sorted(set(pts))


def check_windows(a, b, outs, stats, ops, rule=0, cutoff=1, subtract=0, ovr=1):
    n = a.n_words
    checked = 0
    for i0 in window_starts(n):
        i1 = min(n, i0 + WINDOW)
        lo = a.get_word(i0)[0] if i0 else 0
        to_end = i1 == n
        hi = None if to_end else a.get_word(i1)[0]

        def cut(lst):
            first = lst.lower_bound(lo) if lo else 0
            last = lst.n_words if to_end else lst.lower_bound(hi)
            return first, last

        fa, la = cut(a)
        fb, lb = cut(b)
        assert (fa, la) == (i0, i1)
        ha, hb = a.download_range(fa, la - fa), b.download_range(fb, lb - fb)
        exp = O.compare(ha, hb, ops, rule, cutoff, subtract, ovr)
        for bit, (n_exp, _, recs) in exp.items():
            fo, lo_ = cut(outs[bit])
            assert lo_ - fo == n_exp, "op %d window at %d: %d records, oracle %d" % (bit, i0, lo_ - fo, n_exp)
            got = outs[bit].download_range(fo, lo_ - fo)
            assert got.tobytes() == recs.tobytes(), "op %d window at %d differs from the oracle" % (bit, i0)
            checked += n_exp
    assert checked > 0
    # header totals against a device-side recount of what was written
    for bit, (n_words, total) in stats.items():
        assert outs[bit].n_words == n_words
        assert outs[bit].sum_counts() == total
        assert outs[bit].is_sorted()


def test_config1_intersection_alone_beyond_2_pow_31(ctx):
    """The benchmarked specialisation (ops = intersection alone, default rule: the folded 6-position
    kernel, single pass) with input record indices beyond 2^31: 2 x 2.2e9 k=25 records."""
    n = 2_200_000_000
    a, b = build_pair(ctx, n, 25)
    assert a.is_sorted() and b.is_sorted()
    st, out, _ = ctx.compare(a, b, 2)
    assert st[2][0] == n // 2
    check_windows(a, b, out, st, 2)
    # the same call on the dependency-free two-pass path gives the same totals
    ctx.set_option("two_pass", 1)
    st2, out2, _ = ctx.compare(a, b, 2)
    ctx.set_option("two_pass", 0)
    assert st2 == st
    mid = st[2][0] // 2
    assert out2[2].download_range(mid, 100000).tobytes() == out[2].download_range(mid, 100000).tobytes()


def test_config2_union_and_first_complement_with_cutoff(ctx):
    """BASELINE config 2 exactly: `-u -d -c 3` on the 2 x 2e9 k=25 pair (any-combination kernel, two
    simultaneous output streams, cutoff-driven compaction; the union is 3e9 records > 2^31)."""
    n = 2_000_000_000
    a, b = build_pair(ctx, n, 25)
    st, out, _ = ctx.compare(a, b, 1 | 4, cutoff=3)
    assert st[1][0] > (1 << 31)
    check_windows(a, b, out, st, 1 | 4, cutoff=3)
    stc, _, _ = ctx.compare(a, b, 1 | 4, cutoff=3, count_only=True)
    assert stc == st


def test_config4_shape_k32_full_key_range(ctx):
    """BASELINE config 4's shape on one GPU: k=32 keys over the whole 64-bit range (values >= 2^63),
    2 x 4e9 records, intersection; every key shared so the output is 4e9 records too (record
    indices, tile ranges and output offsets beyond 2^31 and merged positions beyond 2^32)."""
    n = 4_000_000_000
    a, b = ctx.alloc(n, 32), ctx.alloc(n, 32)
    ctx.generate_ex(a, n, 7, 50, 8, 1, 0)
    ctx.generate_ex(b, n, 7, 51, 8, 1, 0)
    assert a.get_word(n - 1)[0] >= (1 << 63)
    st, out, _ = ctx.compare(a, b, 2)
    assert st[2][0] == n
    check_windows(a, b, out, st, 2)
    del out
    # a disjoint partner: nothing in common, the union interleaves both lists
    m = 1_000_000_000
    c = ctx.alloc(m, 32)
    ctx.generate_ex(c, m, 9, 52, 8, 2, 1)   # odd keys only ...
    ctx.generate_ex(b, n, 7, 51, 8, 2, 0)   # ... against even keys only
    st, out, _ = ctx.compare(b, c, 1 | 2)
    assert st[2][0] == 0 and st[1][0] == n + m
    check_windows(b, c, out, st, 1 | 2)


def _check_multi_windows(lists, out, n_words, total, cutoff, rule, ovr=1):
    """N-way analogue of check_windows: key windows cut out of every list by lower_bound, the oracle's
    union_multi (reference src/glistcompare.c:545-591) on the slices, byte-compared with the slice of
    the GPU output between the same keys; header totals recounted on the device."""
    longest = max(lists, key=lambda l: l.n_words)
    n = longest.n_words
    w4 = WINDOW // 4
    starts = {0, n // 2, max(0, n - w4)}
    if out.n_words > (1 << 31):  # a window around output record 2^31
        starts.add(min(max(0, n - w4), max(0, int((1 << 31) * n / out.n_words) - w4 // 2)))
    checked = 0
    for i0 in sorted(starts):
        i1 = min(n, i0 + w4)
        lo = longest.get_word(i0)[0] if i0 else 0
        to_end = i1 == n
        hi = None if to_end else longest.get_word(i1)[0]

        def cut(lst):
            first = lst.lower_bound(lo) if lo else 0
            last = lst.n_words if to_end else lst.lower_bound(hi)
            return first, last

        host = []
        for l in lists:
            f, e = cut(l)
            host.append(l.download_range(f, e - f))
        rc, n_exp, _, recs = O.union_multi(host, cutoff, rule, ovr)
        assert rc == 0
        fo, lo_ = cut(out)
        assert lo_ - fo == n_exp, "window at %d: %d records, oracle %d" % (i0, lo_ - fo, n_exp)
        assert out.download_range(fo, lo_ - fo).tobytes() == recs.tobytes(), "window at %d differs from the oracle" % i0
        checked += n_exp
    assert checked > 0
    assert out.n_words == n_words and out.sum_counts() == total and out.is_sorted()


@pytest.mark.parametrize("dist", ["stride", "iid", "clustered", "genomic"])
def test_config3_eight_way_union_full_size(ctx, dist):
    """BASELINE config 3 on one GPU: the union of eight 5e8-entry k=25 lists (the bench's construction:
    even lists share one key set, odd lists own disjoint ones; genometester4_amd/synth.py) -- by the library's
    own choice (the one-pass N-way tile kernel; the pairwise tree for clustered keys), by the tile kernel
    whatever the keys, and by the pairwise tree (whose intermediate levels keep every key and add raw counts,
    src/glistcompare.c:545-591), each against the oracle on key windows, and against each other.  Key
    distributions: one key per stride of the key space (the bench's default), independent uniform draws,
    stretches of adjacent keys between wide gaps (which the default setting hands to the tree), and GENOMIC
    k-mers (round 5: canonical k-mers of mutated copies of one sequence with planted repeats, made by the
    repo's own sort + fold -- the shape glistmaker's lists have, src/glistmaker.c:914-924; which path the
    library chooses for them is printed)."""
    from genometester4_amd import synth
    n = 500_000_000
    lists = synth.make_lists8(ctx, n, 25, dist)
    n_in = sum(l.n_words for l in lists)
    if dist == "stride":
        assert n_in == 8 * n
    calls, declined = ctx.get_counter("kway_calls"), ctx.get_counter("kway_declined")
    rc, nw, tot, out = ctx.union_multi(lists)
    assert rc == 0 and ctx.get_counter("single_pass_fallbacks") == 0
    if dist == "stride":
        assert nw == 5 * n
    if dist == "clustered":
        assert ctx.get_counter("kway_declined") == declined + 1 and ctx.get_counter("nway_one_pass") == 0
    elif dist == "genomic":  # the library's own choice for real-shaped keys: recorded, either is exact
        one = ctx.get_counter("nway_one_pass")
        assert (ctx.get_counter("kway_declined") == declined + 1) != (one == 1)
        print("genomic keys, 8 x %d records: %s" % (n_in // 8, "one pass of the tile kernel" if one else "declined: pairwise tree"))
    else:
        assert ctx.get_counter("kway_calls") == calls + 1 and ctx.get_counter("nway_one_pass") == 1
    _check_multi_windows(lists, out, nw, tot, 1, 0)
    probe = out.download_range(nw // 3, 200000).tobytes()
    out.free()
    for kway in (0, 3):  # the tree; the tile kernel whatever the keys
        ctx.set_option("kway", kway)
        try:
            rc, nw_t, tot_t, out_t = ctx.union_multi(lists)
        finally:
            ctx.set_option("kway", 1)
        assert (rc, nw_t, tot_t) == (0, nw, tot)
        assert ctx.get_counter("nway_one_pass") == (1 if kway else 0)
        if kway == 0 or dist in ("clustered", "genomic"):
            _check_multi_windows(lists, out_t, nw, tot, 1, 0)
        assert out_t.download_range(nw // 3, 200000).tobytes() == probe
        out_t.free()
    # rule MAX with a cutoff on the result (union_multi :574): both paths again
    cut = 2 if dist == "genomic" else 5  # (k-mer occurrences are 1 almost everywhere: the planted repeats reach 2 and more)
    ctx.set_option("kway", 3)
    try:
        rc, nw2, tot2, out2 = ctx.union_multi(lists, cut, 4)
    finally:
        ctx.set_option("kway", 1)
    assert rc == 0
    _check_multi_windows(lists, out2, nw2, tot2, cut, 4)
    ctx.set_option("kway", 0)
    try:
        rc, nw3, tot3, _ = ctx.union_multi(lists, cut, 4, 1, True)
    finally:
        ctx.set_option("kway", 1)
    assert (nw3, tot3) == (nw2, tot2)


def test_sort_and_fold_beyond_2_pow_32_words(ctx):
    """N2 at a size the 32-bit version refused: 2^32 + 10^6 random k=16 words (32 significant bits, four
    radix passes) -> sorted (word, occurrences) list.  Checked: ascending, the occurrences add up to the
    number of words, and every record with a word below 2^17 against numpy on exactly those words
    (wordtable_sort + wordtable_find_frequencies, reference src/word-table.c:217-260)."""
    import torch
    n, k = (1 << 32) + 1_000_000, 16
    g = torch.Generator(device="cuda")
    g.manual_seed(99)
    words = torch.randint(0, 1 << 32, (n,), dtype=torch.int64, device="cuda", generator=g)
    step = 1 << 30  # (torch's masked select does not take 2^32 elements at once)
    low = np.concatenate([(lambda c: c[c < (1 << 17)].cpu().numpy())(words[i:i + step]) for i in range(0, n, step)]).astype(np.uint64)
    lst = ctx.device_words_to_list(words.data_ptr(), n, k)
    del words
    torch.cuda.empty_cache()
    assert lst.is_sorted()
    assert lst.sum_counts() == n
    exp_keys, exp_counts = np.unique(low, return_counts=True)
    m = lst.lower_bound(1 << 17)
    assert m == len(exp_keys)
    got = lst.download_range(0, m)
    assert (got["key"] == exp_keys).all() and (got["count"] == exp_counts.astype(np.uint32)).all()
    lst.free()


@pytest.mark.parametrize("dist", ["stride", "genomic", "stride x 32"])
def test_count_tables_at_bench_size(ctx, dist):
    """N3 at the size bench.py --workload table runs: six 1e8-entry k=25 lists (even lists share one key
    set).  The union table (4e8 rows x 6) and the table restricted to the keys of list 0, by the N-way tile
    kernel: row count and key column against the N-way union's own output, every column against
    searchsorted on key windows of the lists (gt4_union / gt4_is_union rows, reference
    src/set-operations.c:131-183, :185-228); the merge path (option kway = 0) gives the same bytes on the windows."""
    from genometester4_amd import capi
    import ctypes as C
    n, nl = 100_000_000, 6
    wide = dist == "stride x 32"  # (round 5) 32 lists of 2e7: the 32-list instance of the kernel, one launch as well
    if wide:
        n, nl, dist = 20_000_000, 32, "stride"
    lists = []
    if dist == "genomic":  # (round 5) k-mer lists of mutated copies of one sequence: most keys in all six lists
        from genometester4_amd import synth
        lists = synth.make_lists8(ctx, n, 25, "genomic", nl)
        n = lists[0].n_words
    for j in range(nl if dist == "stride" else 0):
        lst = ctx.alloc(n, 25)
        shared = j % 2 == 0
        ctx.generate_ex(lst, n, 7 if shared else 100 + j, 50 + j, 8, 64 if wide else 16, 0 if shared else 1 + j)
        lists.append(lst)
    rc, nw, tot, uni = ctx.union_multi(lists, 0, 4, 1)  # every key kept
    assert rc == 0
    arr = (C.c_void_p * nl)(*[l.h for l in lists])

    def rows(table, first, count):
        keys = np.empty(count, dtype=np.uint64)
        counts = np.empty((count, nl), dtype=np.uint32)
        assert capi.lib().gt4hip_table_download(ctx.h, C.byref(table), first, count, keys.ctypes.data, counts.ctypes.data) == 0
        return keys, counts

    def column(lst, keys):
        """list `lst`'s counts of `keys` (ascending), 0 where absent, and membership"""
        f, e = lst.lower_bound(int(keys[0])), lst.lower_bound(int(keys[-1]) + 1) if int(keys[-1]) + 1 < (1 << 64) else lst.n_words
        recs = lst.download_range(f, e - f)
        if not len(recs):
            return np.zeros(len(keys), dtype=np.uint32), np.zeros(len(keys), dtype=bool)
        idx = np.searchsorted(recs["key"], keys)
        idx[idx == len(recs)] = 0
        hit = recs["key"][idx] == keys
        return np.where(hit, recs["count"][idx], 0).astype(np.uint32), hit

    W = 200_000
    for kway in (1, 0):
        ctx.set_option("kway", kway)
        try:
            t = capi.CountTable()
            calls = ctx.get_counter("kway_calls")
            assert capi.lib().gt4hip_union_table(ctx.h, arr, nl, C.byref(t)) == 0
            assert t.n_keys == nw
            assert ctx.get_counter("kway_calls") == calls + kway, "one launch of the tile kernel with kway = 1, merges without"
            for first in (0, nw // 2, nw - W):
                keys, counts = rows(t, first, W)
                assert keys.tobytes() == uni.download_range(first, W)["key"].tobytes()
                for j, lst in enumerate(lists):
                    assert counts[:, j].tobytes() == column(lst, keys)[0].tobytes(), "union table, list %d, rows from %d" % (j, first)
            capi.lib().gt4hip_table_free(C.byref(t))
            for presence in (0, 1):
                t = capi.CountTable()
                assert capi.lib().gt4hip_probe_table_ex(ctx.h, arr, nl, presence, C.byref(t)) == 0
                assert t.n_keys == n
                for first in (0, n // 3, n - W):
                    keys, counts = rows(t, first, W)
                    assert keys.tobytes() == lists[0].download_range(first, W)["key"].tobytes()
                    for j, lst in enumerate(lists):
                        c, hit = column(lst, keys)
                        exp = hit.astype(np.uint32) if presence else c
                        assert counts[:, j].tobytes() == exp.tobytes(), "probe table (presence %d), list %d, rows from %d" % (presence, j, first)
                capi.lib().gt4hip_table_free(C.byref(t))
        finally:
            ctx.set_option("kway", 1)
