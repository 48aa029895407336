"""The one-pass N-way union (k_nway_merge, genometester4_amd/csrc/gt4hip_nway.hip) against the CPU
oracle's union_multi (reference src/glistcompare.c:500-603): rules ADD / MAX / NUMBER, cutoff on the
resulting count, zero counts, u32 wrap, empty members, more than eight lists (levels of eight-way
merges), tiles that hold one list only, identical lists (every key eight times), clustered keys (the
tiles' search path), the retry with fewer samples per tile when a tile would not fit LDS, and the
count-only form."""
import numpy as np
import pytest

import gpu_util as U
import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from genometester4_amd import capi
    c = capi.Context(0)
    c.set_option("kway", 3)  # the tile kernel whatever the keys (1, the default, hands clustered keys to the tree)
    yield c
    c.close()


def _check(ctx, lists, k=20, rule=0, cutoff=1, ovr=5, expect_kway=True):
    dev = [ctx.upload(x, k) for x in lists]
    before = ctx.get_counter("kway_calls")
    rc_o, n_o, t_o, r_o = O.union_multi(lists, cutoff, rule, ovr)
    rc_g, n_g, t_g, out = ctx.union_multi(dev, cutoff, rule, ovr)
    assert rc_g == rc_o == 0
    assert (n_g, t_g) == (n_o, t_o)
    assert out.download().tobytes() == r_o.tobytes()
    rc_c, n_c, t_c, _ = ctx.union_multi(dev, cutoff, rule, ovr, True)
    assert (n_c, t_c) == (n_o, t_o)
    if expect_kway:
        assert ctx.get_counter("kway_calls") >= before + 2
    for d in dev:
        d.free()


def _random_lists(rng, n_lists, universe, k_bits=40, zero_counts=True):
    keys = np.unique(rng.integers(0, 1 << k_bits, size=universe, dtype=np.uint64))
    lists = []
    for j in range(n_lists):
        m = rng.random(len(keys)) < rng.uniform(0.05, 0.9)
        c = rng.integers(0 if zero_counts else 1, 7, size=int(m.sum()), dtype=np.uint32)
        lists.append(U.make_records(keys[m], c))
    return lists


@pytest.mark.parametrize("n_lists", [3, 4, 5, 7, 8])
@pytest.mark.parametrize("rule,cutoff", [(0, 1), (1, 0), (4, 3), (7, 2), (1, 9)])
def test_random_lists(ctx, n_lists, rule, cutoff):
    rng = np.random.default_rng(1000 * n_lists + 10 * rule + cutoff)
    _check(ctx, _random_lists(rng, n_lists, 60000), rule=rule, cutoff=cutoff)


@pytest.mark.parametrize("universe", [1, 5, 700, 6143, 6144, 6145, 12289, 200001])
def test_sizes_around_the_tile_capacity(ctx, universe):
    rng = np.random.default_rng(universe)
    keys = np.unique(rng.integers(0, 1 << 44, size=universe, dtype=np.uint64))
    lists = [U.make_records(keys, rng.integers(1, 9, size=len(keys), dtype=np.uint32)) for _ in range(3)]
    lists += [U.make_records(keys[::3], rng.integers(1, 9, size=len(keys[::3]), dtype=np.uint32))]
    _check(ctx, lists, k=22, expect_kway=len(keys) > 0)


def test_eight_identical_lists_and_u32_wrap(ctx):
    rng = np.random.default_rng(8)
    keys = np.unique(rng.integers(0, 1 << 50, size=150000, dtype=np.uint64))
    cnt = rng.integers(1, 5, size=len(keys), dtype=np.uint32)
    cnt[::1000] = 0xFFFFFFFF  # eight times 0xFFFFFFFF wraps to 0xFFFFFFF8; with 0x20000000 x 8 -> 0
    cnt[1::1000] = 0x20000000
    lists = [U.make_records(keys, cnt) for _ in range(8)]
    _check(ctx, lists, k=25)
    _check(ctx, lists, k=25, rule=4, cutoff=2)


def test_disjoint_key_ranges_and_skewed_sizes(ctx):
    rng = np.random.default_rng(9)
    lists = []
    for j in range(6):  # list j owns keys [j * 2^30, (j + 1) * 2^30): every tile holds one list only
        k = np.unique(rng.integers(j << 30, (j + 1) << 30, size=30000 + 7000 * j, dtype=np.uint64))
        lists.append(U.make_records(k, rng.integers(1, 9, size=len(k), dtype=np.uint32)))
    _check(ctx, lists, k=20)
    _check(ctx, lists[::-1], k=20)
    big = np.unique(rng.integers(0, 1 << 40, size=900000, dtype=np.uint64))
    lists = [U.make_records(big, rng.integers(1, 9, size=len(big), dtype=np.uint32))]
    for n in (1, 17, 300, 5000):
        s = np.sort(rng.choice(big, size=n, replace=False))
        lists.append(U.make_records(s, rng.integers(1, 9, size=n, dtype=np.uint32)))
    own = np.unique(rng.integers(0, 1 << 40, size=2000, dtype=np.uint64))
    lists.append(U.make_records(own, np.ones(len(own), np.uint32)))
    _check(ctx, lists, k=20, cutoff=2)


def test_k32_keys_up_to_all_ones(ctx):
    rng = np.random.default_rng(10)
    base = np.unique(rng.integers(0, 1 << 63, size=40000, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=40000, dtype=np.uint64))
    top = np.array([0xFFFFFFFFFFFFFFFE, 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
    lists = []
    for j in range(5):
        m = rng.random(len(base)) < 0.5
        k = np.concatenate([base[m], top[j % 2:]])
        lists.append(U.make_records(k, rng.integers(1, 9, size=len(k), dtype=np.uint32)))
    _check(ctx, lists, k=32)


@pytest.mark.parametrize("n_lists", [9, 12, 17, 26])
def test_more_than_eight_lists(ctx, n_lists):
    rng = np.random.default_rng(n_lists)
    lists = _random_lists(rng, n_lists, 40000)
    lists.insert(3, lists[0][:0])  # an empty member is skipped (:525-532)
    # one pass over up to 32 lists (round 5), levels of eight-way merges, and the library's choice between the two
    # (a probe of how many lists share a key; option "kway_max": 33 / 8 / 32)
    for kmax in (33, 8, 32):
        ctx.set_option("kway_max", kmax)
        try:
            _check(ctx, lists, rule=1, cutoff=2)
            _check(ctx, lists, rule=4, cutoff=0)
            assert kmax == 32 or ctx.get_counter("kway_width") == (32 if kmax == 33 else 8)
        finally:
            ctx.set_option("kway_max", 32)


def test_thirty_two_lists_in_one_pass(ctx):
    """glistmaker's collation width (reference src/glistmaker.c:787-835) as ONE launch of the tile kernel: lists of very
    different lengths that share little, lists of which sixteen are the same (the library then takes levels of eight),
    thirty-three lists (a level of 32, then the pair kernel)"""
    rng = np.random.default_rng(32)
    keys = np.unique(rng.integers(0, 1 << 44, size=400000, dtype=np.uint64))
    lists = []
    for j in range(32):
        m = rng.random(len(keys)) < (0.02 if j % 3 else 0.2)
        lists.append(U.make_records(keys[m], rng.integers(0, 9, size=int(m.sum()), dtype=np.uint32)))
    _check(ctx, lists, k=22, rule=1, cutoff=1)
    assert ctx.get_counter("kway_width") == 32 and ctx.get_counter("kway_shared_x100") < 500
    same = U.make_records(keys[::2], rng.integers(1, 9, size=len(keys[::2]), dtype=np.uint32))
    shared = [same if j % 2 == 0 else lists[j] for j in range(32)]
    _check(ctx, shared, k=22, rule=1, cutoff=3)
    assert ctx.get_counter("kway_width") == 8 and ctx.get_counter("kway_shared_x100") > 500
    ctx.set_option("kway_max", 33)
    try:
        _check(ctx, shared, k=22, rule=4, cutoff=2)
        _check(ctx, lists + [same], k=22, rule=1, cutoff=2)
    finally:
        ctx.set_option("kway_max", 32)


def test_every_tile_through_the_search_path(ctx):
    """Option kway_vt = 99 sends every tile through the path clustered keys take (records back to LDS as
    sorted runs, positions by lower bounds in all the runs)."""
    ctx.set_option("kway_vt", 99)
    try:
        for n_lists, rule, cutoff in ((3, 0, 1), (8, 4, 2), (5, 7, 3)):
            rng = np.random.default_rng(4242 + n_lists)
            _check(ctx, _random_lists(rng, n_lists, 50000), rule=rule, cutoff=cutoff)
    finally:
        ctx.set_option("kway_vt", 0)


def test_every_tile_bucketed_by_its_pivot_run(ctx):
    """Option kway_vt = 98: every tile skips the interpolation and is bucketed by rank in its longest run
    (what tiles with clustered keys do on their own)."""
    ctx.set_option("kway_vt", 98)
    try:
        for n_lists, rule, cutoff in ((3, 0, 1), (8, 4, 2), (5, 7, 3), (8, 1, 0)):
            rng = np.random.default_rng(989 + n_lists)
            _check(ctx, _random_lists(rng, n_lists, 50000), rule=rule, cutoff=cutoff)
        # identical lists: every key eight times, the pivot's buckets hold eight keys each
        keys = np.unique(rng.integers(0, 1 << 50, size=60000, dtype=np.uint64))
        _check(ctx, [U.make_records(keys, rng.integers(1, 9, size=len(keys), dtype=np.uint32)) for _ in range(8)], k=25)
    finally:
        ctx.set_option("kway_vt", 0)


def test_clustered_keys(ctx):
    """Clusters of adjacent keys 2^40 apart: the interpolation puts a whole cluster into one bucket, the
    tiles take the search path on their own."""
    rng = np.random.default_rng(31)
    centres = np.sort(rng.choice(1 << 22, size=900, replace=False).astype(np.uint64)) << np.uint64(40)
    keys = np.unique((centres[:, None] + rng.integers(0, 400, size=(900, 150), dtype=np.uint64)).ravel())
    lists = []
    for j in range(6):
        m = rng.random(len(keys)) < 0.5
        lists.append(U.make_records(keys[m], rng.integers(0, 7, size=int(m.sum()), dtype=np.uint32)))
    _check(ctx, lists, k=32)
    _check(ctx, lists, k=32, rule=4, cutoff=3)
    # one dense cluster inside an otherwise uniform list: only the tiles around it search
    uni = np.unique(rng.integers(0, 1 << 50, size=200000, dtype=np.uint64))
    dense = np.arange(1 << 49, (1 << 49) + 30000, dtype=np.uint64)
    lists = []
    for j in range(4):
        k = np.unique(np.concatenate([uni[rng.random(len(uni)) < 0.6], dense[rng.random(len(dense)) < 0.7]]))
        lists.append(U.make_records(k, rng.integers(1, 9, size=len(k), dtype=np.uint32)))
    _check(ctx, lists, k=25)


@pytest.mark.parametrize("g", [28, 30, 34])
def test_tiles_that_do_not_fit_are_cut_in_two(ctx, g):
    """More samples per tile than the capacity takes on average (option kway_g): about half of the tiles exceed the 64 wave
    slots and are cut at the middle key of their longest run (k_nway_need / _emit; counter kway_splits) -- same bytes
    as the oracle's, no retry with fewer samples."""
    rng = np.random.default_rng(g)
    lists = _random_lists(rng, 8, 400000, zero_counts=False)
    ctx.set_option("kway_g", g)
    try:
        before = ctx.get_counter("kway_overflows")
        _check(ctx, lists, k=20)
        assert ctx.get_counter("kway_splits") > 0
        if g <= 30:
            assert ctx.get_counter("kway_overflows") == before
        _check(ctx, lists, k=20, rule=4, cutoff=2)
    finally:
        ctx.set_option("kway_g", 0)


def test_default_setting_hands_clustered_keys_to_the_tree(ctx):
    """Option kway = 1 (the default): a probe of the longest list's keys sees stretches of adjacent keys between
    wide gaps and the call takes the pairwise tree -- same bytes, counter kway_declined; evenly spread keys of the
    same size stay on the tile kernel."""
    rng = np.random.default_rng(77)
    n_cl = 400
    centres = np.sort(rng.choice(1 << 20, size=n_cl, replace=False).astype(np.uint64)) << np.uint64(30)
    clustered = np.unique((centres[:, None] + rng.integers(0, 5000, size=(n_cl, 900), dtype=np.uint64)).ravel())
    uniform = np.unique(rng.integers(0, 1 << 50, size=len(clustered), dtype=np.uint64))
    ctx.set_option("kway", 1)
    try:
        for keys, declined in ((clustered, True), (uniform, False)):
            lists = []
            for j in range(5):
                m = rng.random(len(keys)) < 0.6
                lists.append(U.make_records(keys[m], rng.integers(1, 7, size=int(m.sum()), dtype=np.uint32)))
            d0, c0 = ctx.get_counter("kway_declined"), ctx.get_counter("kway_calls")
            _check(ctx, lists, k=25, expect_kway=False)
            assert (ctx.get_counter("kway_declined") > d0) == declined
            assert (ctx.get_counter("kway_calls") > c0) == (not declined)
            assert ctx.get_counter("nway_one_pass") == (0 if declined else 1)
    finally:
        ctx.set_option("kway", 3)


@pytest.mark.parametrize("span_bits", [14, 22, 31])
def test_dense_keys_take_the_narrow_path(ctx, span_bits):
    """Tiles whose keys span less than 2^32 group and walk 4-byte keys relative to the tile's smallest
    possible key (the bench's lists do; random 40-bit universes do not)."""
    rng = np.random.default_rng(span_bits)
    base = np.uint64(0x0123456789000000)
    keys = base + np.unique(rng.integers(0, 1 << span_bits, size=min(90000, 1 << (span_bits - 1)), dtype=np.uint64))
    lists = []
    for j in range(5):
        m = rng.random(len(keys)) < (0.9 if j < 3 else 0.3)
        lists.append(U.make_records(keys[m], rng.integers(0, 7, size=int(m.sum()), dtype=np.uint32)))
    _check(ctx, lists, k=32, expect_kway=len(keys) > 10)
    _check(ctx, lists, k=32, rule=4, cutoff=2, expect_kway=len(keys) > 10)


@pytest.mark.parametrize("span", [(1 << 32) - 1, 1 << 32, (1 << 32) + 1])
def test_one_tile_spanning_2_pow_32(ctx, span):
    """One tile from key x to key x + span: the last narrow span, the first wide ones; the low dwords
    of the keys wrap inside the tile."""
    rng = np.random.default_rng(span & 0xff)
    x = 0x00000007fffffff0
    inner = x + np.unique(rng.integers(1, span, size=1500, dtype=np.uint64))
    keys = np.unique(np.concatenate([np.array([x, x + span], dtype=np.uint64), inner.astype(np.uint64)]))
    lists = []
    for j in range(3):
        m = rng.random(len(keys)) < 0.7
        m[0] = m[-1] = True
        lists.append(U.make_records(keys[m], rng.integers(1, 9, size=int(m.sum()), dtype=np.uint32)))
    _check(ctx, lists, k=32)


def test_tile_overflow_is_retried_with_fewer_samples_per_tile(ctx):
    """64 samples per tile = 8192 records expected per tile > the LDS capacity: the partition check
    refuses, the cut is redone with fewer samples per tile; still the reference's bytes."""
    rng = np.random.default_rng(11)
    lists = _random_lists(rng, 5, 300000)
    before = ctx.get_counter("kway_overflows")
    ctx.set_option("kway_g", 64)
    try:
        _check(ctx, lists, expect_kway=False)
    finally:
        ctx.set_option("kway_g", 0)
    assert ctx.get_counter("kway_overflows") > before
    ctx.set_option("kway", 0)
    try:
        _check(ctx, lists, expect_kway=False)
    finally:
        ctx.set_option("kway", 3)


def test_generated_eight_lists_against_the_oracle(ctx):
    """Eight 2.5e6-record k=25 lists generated in HBM (the bench's construction: even lists share
    one key set, odd lists own disjoint residue classes), three sample levels deep."""
    n = 2_500_000
    dev, host = [], []
    for j in range(8):
        lst = ctx.alloc(n, 25)
        shared = j % 2 == 0
        ctx.generate_ex(lst, n, 7 if shared else 100 + j, 50 + j, 8, 16, 0 if shared else 1 + j)
        dev.append(lst)
        host.append(lst.download())
    rc_o, n_o, t_o, r_o = O.union_multi(host, 1, 0, 1)
    before = ctx.get_counter("kway_calls")
    rc_g, n_g, t_g, out = ctx.union_multi(dev)
    assert ctx.get_counter("kway_calls") == before + 1
    assert (n_g, t_g) == (n_o, t_o) and n_g == 5 * n
    assert out.download().tobytes() == r_o.tobytes()
    rc_o, n_o, t_o, r_o = O.union_multi(host, 12, 0, 1)
    rc_g, n_g, t_g, out = ctx.union_multi(dev, 12)
    assert (n_g, t_g) == (n_o, t_o)
    assert out.download().tobytes() == r_o.tobytes()


def test_device_shards_world_of_one_with_rccl_gather(ctx):
    """genometester4_amd.distributed.DeviceShards on one GPU: shards cut on the device, the union
    through the C ABI, totals, and the RCCL gatherv of the C ABI with a communicator of one rank --
    the payload never leaves HBM."""
    from genometester4_amd import capi
    from genometester4_amd import distributed as D
    rng = np.random.default_rng(77)
    lists = _random_lists(rng, 6, 120000, zero_counts=False)
    dev = [ctx.upload(x, 20) for x in lists]
    sh = D.DeviceShards(ctx, 0, 1, capi.comm_unique_id())
    try:
        shards = [sh.shard_of(d, 20) for d in dev]
        assert [s.n_words for s in shards] == [len(x) for x in lists]
        for cutoff, rule in ((1, 0), (3, 4)):
            n, total, res, totals = sh.run(shards, D.gpu_union_multi_op(ctx, cutoff, rule), lambda n, t: [(n, t)])
            rc_o, n_o, t_o, r_o = O.union_multi(lists, cutoff, rule, 1)
            assert (n, total) == (n_o, t_o) and totals == [(n_o, t_o)]
            assert res.n_words == n_o and res.download().tobytes() == r_o.tobytes()
        n, total, res, _ = sh.run(shards, D.gpu_intersect_multi_op(ctx), lambda n, t: [(n, t)])
        rc_o, n_o, t_o, r_o = O.intersect_multi(lists, 1, 0, 1)
        assert (n, total) == (n_o, t_o) and res.download().tobytes() == r_o.tobytes()
    finally:
        sh.close()


def _check_table(ctx, lists, k=20, expect_kway=True):
    """gt4hip_union_table (glistquery's multi-list dump, reference src/set-operations.c:131-183) by the
    tile kernel (one launch: every tile writes keys and counts where its records start -- a ragged table, gathered by
    gt4hip_table_download or made contiguous by gt4hip_table_compact) against numpy -- keys = sorted union of all
    lists, column j = list j's count of the key, 0 where absent -- and against the table built by merges (option
    kway = 0)."""
    dev = [ctx.upload(x, k) for x in lists]
    before = ctx.get_counter("kway_calls")
    tk, tc = ctx.union_table(dev)
    if expect_kway and 2 <= sum(len(x) > 0 for x in lists) <= 32:
        assert ctx.get_counter("kway_calls") == before + 1
        assert ctx.last_table_was_ragged
        ck, cc = ctx.union_table(dev, compact=True)
        assert ck.tobytes() == tk.tobytes() and cc.tobytes() == tc.tobytes()
    uni = np.unique(np.concatenate([x["key"] for x in lists])) if lists else np.zeros(0, dtype=np.uint64)
    assert tk.tobytes() == uni.tobytes()
    for j, x in enumerate(lists):
        col = np.zeros(len(uni), dtype=np.uint32)
        col[np.searchsorted(uni, x["key"])] = x["count"]
        assert tc[:, j].tobytes() == col.tobytes(), "column %d" % j
    ctx.set_option("kway", 0)
    try:
        mk, mc = ctx.union_table(dev)
    finally:
        ctx.set_option("kway", 3)
    assert mk.tobytes() == tk.tobytes() and mc.tobytes() == tc.tobytes()
    # the tables restricted to the keys of list 0 (gt4_is_union / search_lists_multi; reference
    # src/set-operations.c:185-228, src/glistquery.c:776-812): counts, and membership (a list may hold a key with count 0)
    if len(lists) and len(lists[0]):
        before = ctx.get_counter("kway_calls")
        pk, pc = ctx.union_table(dev, probe=True)
        pk2, pp = ctx.union_table(dev, probe=True, presence=True)
        if expect_kway and 2 <= 1 + sum(len(x) > 0 for x in lists[1:]) <= 32:
            assert ctx.get_counter("kway_calls") == before + 2
        assert pk.tobytes() == lists[0]["key"].tobytes() == pk2.tobytes()
        for j, x in enumerate(lists):
            if len(x):
                idx = np.searchsorted(x["key"], lists[0]["key"])
                idx[idx == len(x)] = 0
                hit = x["key"][idx] == lists[0]["key"]
                cnt = np.where(hit, x["count"][idx], 0).astype(np.uint32)
            else:
                hit = np.zeros(len(lists[0]), dtype=bool)
                cnt = np.zeros(len(lists[0]), dtype=np.uint32)
            assert pp[:, j].tobytes() == hit.astype(np.uint32).tobytes(), "membership column %d" % j
            assert pc[:, j].tobytes() == cnt.tobytes(), "probe column %d" % j
        ctx.set_option("kway", 0)
        try:
            qk, qc = ctx.union_table(dev, probe=True)
            _, qp = ctx.union_table(dev, probe=True, presence=True)
        finally:
            ctx.set_option("kway", 3)
        assert qk.tobytes() == pk.tobytes() and qc.tobytes() == pc.tobytes() and qp.tobytes() == pp.tobytes()
    for d in dev:
        d.free()


@pytest.mark.parametrize("n_lists", [2, 3, 6, 8])
@pytest.mark.parametrize("universe", [1, 700, 6145, 200001, 3_000_000])
def test_count_table_by_the_tile_kernel(ctx, n_lists, universe):
    rng = np.random.default_rng(31 * n_lists + universe)
    _check_table(ctx, _random_lists(rng, n_lists, universe))


@pytest.mark.parametrize("n_lists", [9, 12, 17, 32])
@pytest.mark.parametrize("universe", [1, 700, 40000, 1_500_000])
def test_count_table_of_nine_to_thirty_two_lists_in_one_launch(ctx, n_lists, universe):
    """glistquery's table over up to 32 lists (reference src/set-operations.c:131-183, :185-228): the 32-list instance of the
    tile kernel, one launch, against numpy and against the table built by merges"""
    rng = np.random.default_rng(131 * n_lists + universe)
    lists = _random_lists(rng, n_lists, universe)
    _check_table(ctx, lists, k=25)
    if universe == 40000:  # empty members, a list twice, list 0 short
        empty = U.make_records(np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
        mixed = [lists[0][: len(lists[0]) // 7]] + [empty] + lists[1:-1] + [lists[1].copy()]
        _check_table(ctx, mixed, k=25)


def test_count_table_with_empty_members_identical_lists_and_more_than_eight(ctx):
    rng = np.random.default_rng(77)
    lists = _random_lists(rng, 5, 50000)
    empty = U.make_records(np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
    _check_table(ctx, [lists[0], empty, lists[1], lists[2], empty, lists[3]])          # columns of empty lists stay 0
    _check_table(ctx, [lists[0], lists[0].copy(), lists[0].copy(), lists[1]])          # equal keys in several lists
    _check_table(ctx, [lists[0], empty], expect_kway=False)                            # one non-empty list: by merges
    _check_table(ctx, _random_lists(rng, 11, 30000))                                   # more than eight: the 32-list instance of the kernel
    ctx.set_option("kway_max", 8)
    try:
        _check_table(ctx, _random_lists(rng, 11, 30000), expect_kway=False)            # ... by merges when that is switched off
    finally:
        ctx.set_option("kway_max", 32)
    _check_table(ctx, _random_lists(rng, 34, 9000), expect_kway=False)                 # more than 32: by merges
    # 390 columns, three of them with a list: wider than a batch of rows in LDS is worth (384): rows zeroed in global
    # memory and written by scattered stores, as before round 5 (the probe tables: zeroed by the host)
    _check_table(ctx, [lists[0]] + [empty] * 200 + [lists[1]] + [empty] * 187 + [lists[2]])
    _check_table(ctx, [lists[0]] + [empty] * 381 + [lists[1], lists[2]])              # 384 columns: the last width through LDS
    # keys 0 and 2^64 - 1 (the all-ones filler of the bucket walks is a legal k = 32 key)
    edge = [U.make_records(np.array([0, 5, (1 << 64) - 1], dtype=np.uint64), np.array([3, 0, 7], dtype=np.uint32)),
            U.make_records(np.array([5, (1 << 63), (1 << 64) - 1], dtype=np.uint64), np.array([1, 2, 0], dtype=np.uint32)),
            U.make_records(np.array([(1 << 64) - 1], dtype=np.uint64), np.array([9], dtype=np.uint32))]
    _check_table(ctx, edge, k=32)


def test_count_table_of_clustered_keys(ctx):
    """tiles that are bucketed by their pivot run or take the search path (see test_clustered_keys)"""
    rng = np.random.default_rng(5)
    base = np.sort(rng.choice(1 << 20, size=3000, replace=False).astype(np.uint64)) << np.uint64(40)
    keys = np.unique((base[:, None] + np.arange(40, dtype=np.uint64)[None, :]).ravel())
    lists = []
    for j in range(6):
        m = rng.random(len(keys)) < 0.5
        lists.append(U.make_records(keys[m], rng.integers(1, 9, size=int(m.sum()), dtype=np.uint32)))
    _check_table(ctx, lists, k=32)
    # twelve lists of the same clustered keys: the 32-list instance of the kernel (round 5)
    wide = []
    for j in range(12):
        m = rng.random(len(keys)) < 0.3
        wide.append(U.make_records(keys[m], rng.integers(1, 9, size=int(m.sum()), dtype=np.uint32)))
    _check_table(ctx, wide, k=32)
    for vt in (98, 99):
        ctx.set_option("kway_vt", vt)
        try:
            _check_table(ctx, lists[:4], k=32)
            _check_table(ctx, wide[:10], k=32)
        finally:
            ctx.set_option("kway_vt", 0)


@pytest.mark.parametrize("n_lists", [3, 8])
def test_partition_by_bracket_searches_gives_the_same_tiles_result(ctx, n_lists):
    """The tile boundaries come from the merged samples' list numbers (prefix counts + a search inside
    one sample stretch); option kway_vt = 97 keeps the older partition (binary searches over brackets of
    64 tiles).  Both against the oracle, on lists with many equal keys (ties at the boundaries: a sample
    equal to the boundary key merged behind it)."""
    rng = np.random.default_rng(97 + n_lists)
    lists = _random_lists(rng, n_lists, 400000, k_bits=19)  # dense: most keys are in several lists
    _check(ctx, lists, rule=1, cutoff=2)
    ctx.set_option("kway_vt", 97)
    try:
        _check(ctx, lists, rule=1, cutoff=2)
    finally:
        ctx.set_option("kway_vt", 0)
