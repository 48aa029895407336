"""GPU parity: the HIP path (through the C ABI, libgt4hip.so) against the reference's golden
outputs and against the CPU oracle on seeded inputs.  Bit-exact: integer/byte work only."""
import numpy as np
import pytest

import golden_util as G
import gpu_util as U
import oracle_lib as O

pytestmark = pytest.mark.gpu

CASES, INPUTS, OUTPUTS = G.load()
FILE_CASES = [c for c in CASES if c["tool"] == "glistcompare" and c["exit"] == 0 and c["files"]]
COUNT_CASES = [c for c in CASES if c["tool"] == "glistcompare" and c["exit"] == 0 and "--count_only" in c["argv"]]


@pytest.fixture(scope="module")
def ctx():
    from genometester4_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _gpu_case(ctx, case, count_only=False):
    """Runs what the CLI would run for this argv; returns ({file: bytes}, [(n, total)...])."""
    p = G.parse_argv(case["argv"])
    lists = [G.input_records(INPUTS, f) for f in p["files"]]
    k = lists[0][1]
    dev = [ctx.upload(r, k) for r, _ in lists]
    files, stats = {}, []
    if len(dev) == 2:
        st, out, _ = ctx.compare(dev[0], dev[1], p["ops"], p["rule"], p["cutoff"], p["subtract"], p["count_override"], count_only)
        for bit in sorted(st):
            n, total = st[bit]
            stats.append((n, total))
            if not count_only:
                recs = out[bit].download()
                assert len(recs) == n
                files["%s_%d_%s.list" % (p["out"], k, G.OP_FILES[bit])] = G.list_file_bytes(k, n, total, recs)
    else:
        for bit, fn, name in ((1, ctx.union_multi, "union"), (2, ctx.intersect_multi, "intrsec")):
            if not p["ops"] & bit:
                continue
            rc, n, total, out = fn(dev, p["cutoff"], p["rule"], p["count_override"], count_only)
            assert rc == 0
            stats.append((n, total))
            if not count_only:
                files["%s_%d_%s.list" % (p["out"], k, name)] = G.list_file_bytes(k, n, total, out.download())
    return files, stats


@pytest.mark.parametrize("case", FILE_CASES, ids=[c["id"] for c in FILE_CASES])
def test_gpu_reproduces_reference_files(ctx, case):
    files, _ = _gpu_case(ctx, case)
    assert sorted(files) == sorted(case["files"])
    for name, data in files.items():
        assert data == bytes(OUTPUTS["%s/%s" % (case["id"], name)]), name


@pytest.mark.parametrize("case", COUNT_CASES, ids=[c["id"] for c in COUNT_CASES])
def test_gpu_count_only_matches_reference_stdout(ctx, case):
    _, stats = _gpu_case(ctx, case, count_only=True)
    assert "".join("NUnique\t%d\nNTotal\t%d\n" % s for s in stats) == case["stdout"]


def test_gpu_multi_rule_rejection(ctx):
    from genometester4_amd import capi
    recs = [ctx.upload(INPUTS["M%d" % j][0], 8) for j in range(4)]
    for c in CASES:
        if c["id"].startswith("multi_r") and c["exit"] == 1 and "Invalid rule" in c["stderr"]:
            p = G.parse_argv(c["argv"])
            fn = ctx.union_multi if p["ops"] & 1 else ctx.intersect_multi
            assert fn(recs, p["cutoff"], p["rule"], p["count_override"])[0] == capi.ERULE


def _check_pair(ctx, a, b, ops, rule=0, cutoff=1, subtract=0, ovr=1, k=16):
    da, db = ctx.upload(a, k), ctx.upload(b, k)
    exp = O.compare(a, b, ops, rule, cutoff, subtract, ovr)
    st, out, _ = ctx.compare(da, db, ops, rule, cutoff, subtract, ovr)
    for bit, (n, total, recs) in exp.items():
        assert st[bit] == (n, total), "op %d stats" % bit
        got = out[bit].download()
        assert got.tobytes() == recs.tobytes(), "op %d records" % bit
    st2, _, _ = ctx.compare(da, db, ops, rule, cutoff, subtract, ovr, count_only=True)
    assert st2 == st


@pytest.mark.parametrize("rule", range(8))
@pytest.mark.parametrize("cutoff", [0, 1, 3])
def test_random_pairs_all_rules(ctx, rule, cutoff):
    a, b = U.random_pair(100 + rule * 7 + cutoff, 60000, 0.6, 0.5)
    for sub in (0, 1):
        _check_pair(ctx, a, b, 15, rule, cutoff, sub, ovr=5)


@pytest.mark.parametrize("ops", [1, 2, 4, 8, 3, 5, 12, 10])
def test_op_subsets(ctx, ops):
    a, b = U.random_pair(7, 40000, 0.5, 0.7)
    _check_pair(ctx, a, b, ops, cutoff=2)


def test_tile_boundary_sizes(ctx):
    """Sizes straddling the merge tile; identical lists put a matched pair on every boundary."""
    T = U.merge_tile()
    rng = np.random.default_rng(5)
    T2 = U.merge_tile(1)  # the large geometry's tile (1024 threads x 4 records - slack)
    T3 = T2 + 2 * 1024    # ... and the intersection's (6 positions per thread)
    for n in (1, 2, T // 2 - 1, T // 2, T // 2 + 1, T - 1, T, T + 1, 2 * T, 3 * T + 1, 10 * T - 1,
              T2 // 2, T2 // 2 + 1, T2 - 1, T2, T2 + 1, 3 * T2 + 1,
              T3 // 2, T3 // 2 + 1, T3 - 1, T3, T3 + 1, 3 * T3 + 1):
        keys = np.unique(rng.integers(0, 1 << 40, size=n, dtype=np.uint64))
        a = U.make_records(keys, rng.integers(1, 9, size=len(keys), dtype=np.uint32))
        b = U.make_records(keys, rng.integers(1, 9, size=len(keys), dtype=np.uint32))
        _check_pair(ctx, a, b, 15, k=20)          # identical key sets
        _check_pair(ctx, a, b[::2], 15, k=20)     # every other key shared
        _check_pair(ctx, a[: n // 3], b, 15, k=20)


def test_empty_and_ragged(ctx):
    a, b = U.random_pair(11, 50000, 0.9, 0.02)
    empty = a[:0]
    _check_pair(ctx, empty, empty, 15)
    _check_pair(ctx, a, empty, 15)
    _check_pair(ctx, empty, b, 15)
    _check_pair(ctx, a, b, 15)
    _check_pair(ctx, b, a, 15)
    # disjoint ranges: all of A below all of B, and interleaved-by-parity
    lo = U.make_records(np.arange(1000, dtype=np.uint64), np.ones(1000, np.uint32))
    hi = U.make_records(np.arange(5000, 9000, dtype=np.uint64), np.full(4000, 2, np.uint32))
    _check_pair(ctx, lo, hi, 15)
    _check_pair(ctx, hi, lo, 15)
    ev = U.make_records(np.arange(0, 20000, 2, dtype=np.uint64), np.ones(10000, np.uint32))
    od = U.make_records(np.arange(1, 20000, 2, dtype=np.uint64), np.ones(10000, np.uint32))
    _check_pair(ctx, ev, od, 15)


@pytest.mark.parametrize("ops", [1, 2, 4, 8, 15])
def test_skewed_tiles(ctx, ops):
    """Tiles that are nearly all A or nearly all B (every chunk slot of a wavefront holds one list),
    runs shorter than 256 records inside a tile (the search's short-run path), and a dense cluster
    of shared keys: each output alone (specialised kernels) and all four together."""
    rng = np.random.default_rng(77)
    big = np.unique(rng.integers(0, 1 << 44, size=150000, dtype=np.uint64))
    few = np.sort(rng.choice(big, size=90, replace=False))               # all shared, sparse
    own = np.unique(rng.integers(0, 1 << 44, size=70, dtype=np.uint64))   # mostly private, sparse
    cluster = big[60000:60400]                                            # 400 consecutive shared keys
    small_keys = np.unique(np.concatenate([few, own, cluster]))
    a = U.make_records(big, rng.integers(1, 9, size=len(big), dtype=np.uint32))
    b = U.make_records(small_keys, rng.integers(1, 9, size=len(small_keys), dtype=np.uint32))
    _check_pair(ctx, a, b, ops, k=22, cutoff=2)
    _check_pair(ctx, b, a, ops, k=22, cutoff=2)


def test_k32_full_range_keys(ctx):
    a, b = U.random_pair(3, 30000, 0.6, 0.6, k=32)
    top = U.make_records([0xFFFFFFFFFFFFFFFE, 0xFFFFFFFFFFFFFFFF], [3, 4])
    a = np.concatenate([a[a["key"] < 0xFFFFFFFFFFFFFFFE], top])
    b = np.concatenate([b[b["key"] < 0xFFFFFFFFFFFFFFFE], top[1:]])
    _check_pair(ctx, a, b, 15, k=32)
    _check_pair(ctx, a, b, 15, rule=1, cutoff=0, k=32)


def test_two_pass_path_equals_lookback(ctx):
    a, b = U.random_pair(21, 80000, 0.6, 0.6)
    ctx.set_option("two_pass", 1)
    try:
        _check_pair(ctx, a, b, 15, cutoff=2)
    finally:
        ctx.set_option("two_pass", 0)


@pytest.mark.parametrize("rule", [0, 1, 3, 4, 7])
@pytest.mark.parametrize("cutoff", [0, 1, 4])
def test_multi_random(ctx, rule, cutoff):
    rng = np.random.default_rng(rule * 10 + cutoff)
    keys = np.unique(rng.integers(0, 1 << 30, size=50000, dtype=np.uint64))
    lists = []
    for j in range(5):
        m = rng.random(len(keys)) < (0.2 + 0.15 * j)
        c = rng.integers(0, 6, size=int(m.sum()), dtype=np.uint32)   # zero counts included (MINZ fold)
        lists.append(U.make_records(keys[m], c))
    lists.insert(2, lists[0][:0])  # an empty member
    dev = [ctx.upload(x, 16) for x in lists]
    for fn_g, fn_o, ls_g, ls_o in ((ctx.union_multi, O.union_multi, dev, lists),
                                   (ctx.intersect_multi, O.intersect_multi, dev, lists),
                                   (ctx.intersect_multi, O.intersect_multi, dev[:2] + dev[3:], lists[:2] + lists[3:])):
        rc_o, n_o, t_o, r_o = fn_o(ls_o, cutoff, rule, 9)
        rc_g, n_g, t_g, out = fn_g(ls_g, cutoff, rule, 9)
        assert (rc_g != 0) == (rc_o != 0)
        if rc_o:
            continue
        assert (n_g, t_g) == (n_o, t_o)
        assert out.download().tobytes() == r_o.tobytes()
        rc_c, n_c, t_c, _ = fn_g(ls_g, cutoff, rule, 9, True)
        assert (n_c, t_c) == (n_o, t_o)


def test_union_table_matches_oracle_walk(ctx):
    lists = [INPUTS["M%d" % j][0] for j in range(4)]
    keys, counts = ctx.union_table([ctx.upload(x, 8) for x in lists])
    _, rows = O.union_walk(lists)
    rows = [r for r in rows if any(r[1:])]  # the reference's duplicate all-zero visits are a host-walk quirk
    assert [int(k) for k in keys] == [r[0] for r in rows]
    assert counts.tolist() == [list(r[1:]) for r in rows]


def test_list_utilities(ctx):
    a, _ = U.random_pair(2, 20000, 0.7, 0.1, special=False)
    d = ctx.upload(a, 16)
    assert d.n_words == len(a) and d.word_length == 16
    assert d.sum_counts() == int(a["count"].astype(np.uint64).sum())
    assert d.is_sorted()
    bad = a.copy()
    bad[100], bad[101] = a[101], a[100]
    assert not ctx.upload(bad, 16).is_sorted()
    for key in (0, int(a["key"][17]), int(a["key"][17]) + 1, 1 << 40):
        assert d.lower_bound(key) == int(np.searchsorted(a["key"], np.uint64(key), "left"))
    assert d.get_word(123) == (int(a["key"][123]), int(a["count"][123]))
    assert d.slice(10, 50).download().tobytes() == a[10:60].tobytes()


def test_generator_matches_cpu_restatement(ctx):
    for k, n, seed in ((25, 100003, 1), (32, 50000, 9), (16, 1 << 16, 3)):
        d = ctx.alloc(n, k)
        ctx.generate(d, n, seed, 8)
        assert d.download().tobytes() == U.generate_cpu(n, seed, k).tobytes()
        assert d.is_sorted()
    d = ctx.alloc(70000, 25)
    ctx.generate_ex(d, 70000, 5, 77, 8, 3, 2)
    assert d.download().tobytes() == U.generate_cpu(70000, 5, 25, 8, 77, 3, 2).tobytes()


def test_medium_generated_pair_properties(ctx):
    """2 x 2e7 records generated in HBM: oracle parity + size-independent identities."""
    n = 20_000_000
    a, b = ctx.alloc(n, 25), ctx.alloc(n, 25)
    ctx.generate(a, n, 1, 8)
    ctx.generate(b, n, 1, 8)          # same keys ...
    st, out, _ = ctx.compare(a, b, 3)
    assert st[1][0] == n and st[2][0] == n          # A == B: union = intersection = A
    sa = a.sum_counts()
    assert st[1][1] == 2 * sa and st[2][1] == sa    # ADD doubles, MIN keeps
    assert out[2].download_range(0, 1000).tobytes() == a.download_range(0, 1000).tobytes()
    del out
    ctx.generate(b, n, 2, 8)          # ... then an independent list
    st, out, _ = ctx.compare(a, b, 15, cutoff=1)
    nu, ni, nd1, nd2 = (st[x][0] for x in (1, 2, 4, 8))
    assert nu == 2 * n - ni and nd1 == n - ni and nd2 == n - ni
    assert all(out[x].is_sorted() for x in (1, 2, 4, 8))
    ha, hb = a.download(), b.download()
    exp = O.compare(ha, hb, 15)
    for bit in (1, 2, 4, 8):
        assert st[bit] == exp[bit][:2]
        assert out[bit].download().tobytes() == exp[bit][2].tobytes()


def test_offsets_beyond_2_pow_31_records(ctx):
    """Lists and outputs longer than 2^31 records (26 GB each): every record index, tile range and
    output offset past the 32-bit boundary.  Checked through size-independent identities and
    rank-addressed spot checks, since no CPU oracle finishes at this size."""
    n = 2_200_000_000
    a, b = ctx.alloc(n, 25), ctx.alloc(n, 25)
    ctx.generate_ex(a, n, 7, 50, 8, 16, 1)
    ctx.generate_ex(b, n, 7, 50, 8, 16, 1)          # B == A
    sa = a.sum_counts()
    st, out, _ = ctx.compare(a, b, 3)
    assert st[1] == (n, 2 * sa) and st[2] == (n, sa)
    assert out[1].is_sorted() and out[2].is_sorted()
    for i in (0, (1 << 31) - 1, 1 << 31, (1 << 31) + 4093, n - 1):
        k, c = a.get_word(i)
        assert out[2].get_word(i) == (k, c) and out[1].get_word(i) == (k, 2 * c)
    del out
    m = 300_000_000
    b = ctx.alloc(m, 25)
    ctx.generate_ex(b, m, 8, 51, 8, 16, 2)           # disjoint residue class: A n B is empty
    st, out, _ = ctx.compare(a, b, 1 | 2 | 8)
    assert st[2][0] == 0 and st[8] == (m, b.sum_counts())
    assert st[1] == (n + m, sa + b.sum_counts()) and out[1].is_sorted()
    for i in (1 << 30, (1 << 31) - 7, (1 << 31) + 12345, n - 1):
        k, c = a.get_word(i)
        assert out[1].get_word(i + b.lower_bound(k)) == (k, c)
    for j in (0, m // 2, m - 1):
        k, c = b.get_word(j)
        assert out[1].get_word(j + a.lower_bound(k)) == (k, c)


def test_single_pass_gives_up_and_falls_back(ctx):
    """Twice as many workgroups as the device holds, and waits bounded to a few polls: the workers
    that are resident wait for tiles of workers that are not, give up, and the call is rerun on the
    dependency-free two-pass path -- same bytes, and the context counts the fallback."""
    a, b = U.random_pair(5, 3_000_000, 0.5, 0.5, k=22)
    exp = O.compare(a, b, 15)
    da, db = ctx.upload(a, 22), ctx.upload(b, 22)
    before = ctx.get_counter("single_pass_fallbacks")
    ctx.set_option("grid", 2048)
    ctx.set_option("spin_limit", 3)
    ctx.set_option("dynamic", -1)  # round-robin dealing: tiles of workgroups that are not resident
    try:
        for ops in (2, 1, 15):
            st, out, _ = ctx.compare(da, db, ops)
            for bit in (1, 2, 4, 8):
                if ops & bit:
                    assert st[bit] == exp[bit][:2]
                    assert out[bit].download().tobytes() == exp[bit][2].tobytes()
    finally:
        ctx.set_option("grid", 0)
        ctx.set_option("spin_limit", 0)
        ctx.set_option("dynamic", 0)
    assert ctx.get_counter("single_pass_fallbacks") > before


def test_fallback_with_default_spin_limit_is_fast(ctx):
    """An oversubscribed grid with the DEFAULT wait bound: the first wait that gives up raises the
    launch's error flag, every other wait then ends at its next look (the kernel peeks at the flag
    inside its spin loops), so the launch finishes at its normal pace and the two-pass rerun
    happens within seconds -- not after one full bound per remaining tile."""
    import time
    a, b = U.random_pair(15, 6_000_000, 0.5, 0.5, k=24)
    exp = O.compare(a, b, 3)
    da, db = ctx.upload(a, 24), ctx.upload(b, 24)
    before = ctx.get_counter("single_pass_fallbacks")
    ctx.set_option("grid", 1024)
    ctx.set_option("dynamic", -1)
    try:
        t0 = time.perf_counter()
        st, out, _ = ctx.compare(da, db, 3)
        elapsed = time.perf_counter() - t0
    finally:
        ctx.set_option("grid", 0)
        ctx.set_option("dynamic", 0)
    assert ctx.get_counter("single_pass_fallbacks") > before
    assert elapsed < 60.0, "fallback took %.1f s" % elapsed
    for bit in (1, 2):
        assert st[bit] == exp[bit][:2]
        assert out[bit].download().tobytes() == exp[bit][2].tobytes()


@pytest.mark.parametrize("ops", [1, 2, 3, 5, 15])
def test_tiles_by_ticket_need_no_resident_grid(ctx, ops):
    """Dynamic dealing: tiles go to the workgroups that are running, in arrival order, so four times
    the resident grid finishes on the single-pass path (no fallback) with the oracle's bytes; the
    same sizes round-robin and by ticket give identical outputs."""
    a, b = U.random_pair(21 + ops, 2_500_000, 0.55, 0.45, k=23)
    exp = O.compare(a, b, ops, cutoff=2)
    da, db = ctx.upload(a, 23), ctx.upload(b, 23)
    before = ctx.get_counter("single_pass_fallbacks")
    ctx.set_option("grid", 1024)
    ctx.set_option("dynamic", 1)
    try:
        st, out, _ = ctx.compare(da, db, ops, cutoff=2)
    finally:
        ctx.set_option("grid", 0)
        ctx.set_option("dynamic", 0)
    assert ctx.get_counter("single_pass_fallbacks") == before
    for bit, (n, total, recs) in exp.items():
        assert st[bit] == (n, total)
        assert out[bit].download().tobytes() == recs.tobytes()
    for dyn in (-1, 1):
        ctx.set_option("dynamic", dyn)
        try:
            st2, out2, _ = ctx.compare(da, db, ops, cutoff=2)
        finally:
            ctx.set_option("dynamic", 0)
        assert st2 == st
        for bit in exp:
            assert out2[bit].download().tobytes() == exp[bit][2].tobytes()


@pytest.mark.parametrize("rule", range(8))
@pytest.mark.parametrize("cutoff", [0, 1, 3])
def test_intersection_alone_long_and_short_first_list(ctx, rule, cutoff):
    """ops = 2 alone swaps its inputs when the first list is the longer one (FIRST and SECOND trade
    places, SUBTRACT must not swap): every rule, both orders, against the oracle."""
    rng = np.random.default_rng(900 + rule * 5 + cutoff)
    big = np.unique(rng.integers(0, 1 << 36, size=120000, dtype=np.uint64))
    small = np.sort(rng.choice(big, size=7000, replace=False))
    extra = np.unique(rng.integers(0, 1 << 36, size=2000, dtype=np.uint64))
    small = np.unique(np.concatenate([small, extra]))
    a = U.make_records(big, rng.integers(0, 9, size=len(big), dtype=np.uint32))
    b = U.make_records(small, rng.integers(0, 9, size=len(small), dtype=np.uint32))
    _check_pair(ctx, a, b, 2, rule, cutoff, 0, ovr=5, k=18)
    _check_pair(ctx, b, a, 2, rule, cutoff, 0, ovr=5, k=18)


@pytest.mark.parametrize("rule", [0, 1, 3, 4, 7])
def test_intersect_multi_long_first_list(ctx, rule):
    """intersect_multi with a long first list and short later ones (the chain's pair calls swap)."""
    rng = np.random.default_rng(40 + rule)
    big = np.unique(rng.integers(0, 1 << 34, size=90000, dtype=np.uint64))
    lists = [U.make_records(big, rng.integers(0, 7, size=len(big), dtype=np.uint32))]
    for frac in (0.3, 0.05, 0.5):
        m = rng.random(len(big)) < frac
        lists.append(U.make_records(big[m], rng.integers(0, 7, size=int(m.sum()), dtype=np.uint32)))
    dev = [ctx.upload(x, 17) for x in lists]
    for cutoff in (0, 1, 3):
        rc_o, n_o, t_o, r_o = O.intersect_multi(lists, cutoff, rule, 4)
        rc_g, n_g, t_g, out = ctx.intersect_multi(dev, cutoff, rule, 4)
        assert rc_g == rc_o == 0
        assert (n_g, t_g) == (n_o, t_o)
        assert out.download().tobytes() == r_o.tobytes()


@pytest.mark.parametrize("ops", [2, 4, 8, 1, 15])
@pytest.mark.parametrize("shape", ["wrap", "wide", "mixed"])
def test_key_spans_around_the_32_bit_search(ctx, ops, shape):
    """The intersection / complement kernels compare low dwords relative to the tile's smallest key when
    the tile spans less than 2^32 (rank_group, 32-bit probes).  `wrap`: dense keys around multiples of
    2^32, so the low dwords of one tile wrap; `wide`: every tile spans far more than 2^32 (64-bit
    probes); `mixed`: dense stretches separated by gaps of 2^40, so tiles of both kinds alternate and
    some tiles span just under / just over 2^32."""
    rng = np.random.default_rng(77 + ops)
    if shape == "wrap":
        keys = np.concatenate([(np.uint64(j) << np.uint64(32)) + rng.integers(-(1 << 21), 1 << 21, size=9000).astype(np.int64).astype(np.uint64)
                               for j in (2, 3, 1 << 20)])
    elif shape == "wide":
        keys = rng.integers(0, 1 << 62, size=30000, dtype=np.uint64)
    else:
        parts = []
        for j in range(12):
            base = np.uint64(j + 1) << np.uint64(40)
            width = (1 << 31) + (j - 6) * (1 << 28)  # a stretch of about one tile: spans from 2^31 - 1.5 * 2^30 to 2^31 + 1.25 * 2^30 ... and 2x that with the neighbour
            parts.append(base + rng.integers(0, max(width, 1 << 20), size=2500, dtype=np.uint64))
            parts.append(base + (np.uint64(1) << np.uint64(32)) + rng.integers(0, 1 << 20, size=700, dtype=np.uint64))
        keys = np.concatenate(parts)
    keys = np.unique(keys)
    in_a, in_b = rng.random(len(keys)) < 0.6, rng.random(len(keys)) < 0.6
    a = U.make_records(keys[in_a], rng.integers(1, 9, size=int(in_a.sum()), dtype=np.uint32))
    b = U.make_records(keys[in_b], rng.integers(1, 9, size=int(in_b.sum()), dtype=np.uint32))
    exp = O.compare(a, b, ops, cutoff=2)
    st, out, _ = ctx.compare(ctx.upload(a, 31), ctx.upload(b, 31), ops, cutoff=2)
    for bit, (n, total, recs) in exp.items():
        assert st[bit] == (n, total)
        assert out[bit].download().tobytes() == recs.tobytes()
