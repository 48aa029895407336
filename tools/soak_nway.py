"""Randomised parity soak of the N-way tile kernel (union, count-only, count tables) against the CPU oracle
(GPU box; not part of the test-suite).   python tools/soak_nway.py [seconds] [seed]
Shapes the fixed-seed tests of tests/test_kway.py do not enumerate: 3 - 32 lists (round 5: more than eight through the
32-list instance of the kernel in ONE pass, option kway_max = 33) of very different lengths, their count tables (all keys;
the keys of list 0, counts and membership) against numpy, uniform /
clustered / heavily shared keys, zero counts and counts near 2^32, every rule and cutoff, the tile kernel forced
(option kway = 3), every fallback forced now and then (kway_vt 98 / 99), fuller tiles (kway_g)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from genometester4_amd import capi
import oracle_lib as O
import gpu_util as U

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = capi.Context(0)
ctx.set_option("kway", 3)
t0 = time.time(); n_cases = 0; n_tables = 0
while time.time() - t0 < budget:
    n_lists = int(rng.integers(3, 9)) if rng.random() < 0.5 else int(rng.integers(9, 33))
    ctx.set_option("kway_max", 33 if rng.random() < 0.8 else 32)  # (33: one pass whatever the keys; 32: the library probes and may take levels)
    universe = int(rng.choice([40, 5000, 70000, 400000, 1500000]))
    shape = str(rng.choice(["uniform", "clustered", "shared"]))
    if shape == "clustered":
        centres = np.sort(rng.choice(1 << 20, size=max(1, universe // 900), replace=False).astype(np.uint64)) << np.uint64(28)
        keys = np.unique((centres[:, None] + rng.integers(0, 4000, size=(len(centres), 900), dtype=np.uint64)).ravel())
    else:
        keys = np.unique(rng.integers(0, 1 << int(rng.choice([24, 40, 50])), size=universe, dtype=np.uint64))
    lists = []
    for j in range(n_lists):
        p = 0.97 if shape == "shared" else float(rng.choice([0.01, 0.1, 0.5, 0.9]))
        m = rng.random(len(keys)) < p
        top = int(rng.choice([2, 7, 0xffffffff]))
        c = rng.integers(0, top, size=int(m.sum()), dtype=np.uint64).astype(np.uint32)
        lists.append(U.make_records(keys[m], c))
    rule = int(rng.choice([0, 1, 4, 7])); cutoff = int(rng.choice([0, 1, 2, 3, 9])); ovr = int(rng.integers(0, 5))
    vt = int(rng.choice([0, 0, 0, 98, 99])); g = int(rng.choice([0, 0, 0, 18, 27, 30]))
    ctx.set_option("kway_vt", vt); ctx.set_option("kway_g", g)
    dev = [ctx.upload(x, 25) for x in lists]
    tag = (n_lists, universe, shape, rule, cutoff, ovr, vt, g, [len(x) for x in lists])
    rc_o, n_o, t_o, r_o = O.union_multi(lists, cutoff, rule, ovr)
    rc_g, n_g, t_g, out = ctx.union_multi(dev, cutoff, rule, ovr)
    assert rc_g == rc_o, ("rc", tag)
    if rc_o == 0:
        assert (n_g, t_g) == (n_o, t_o), ("totals", tag, (n_g, t_g), (n_o, t_o))
        assert out.download().tobytes() == r_o.tobytes(), ("records", tag)
        rc_c, n_c, t_c, _ = ctx.union_multi(dev, cutoff, rule, ovr, True)
        assert (n_c, t_c) == (n_o, t_o), ("count only", tag)
        out.free()
    if sum(len(x) for x in lists) <= 3_000_000:  # the count tables (one launch up to 32 lists)
        uni = np.unique(np.concatenate([x["key"] for x in lists]))
        tk, tc = ctx.union_table(dev)
        assert tk.tobytes() == uni.tobytes(), ("table keys", tag)
        for j, x in enumerate(lists):
            col = np.zeros(len(uni), dtype=np.uint32)
            col[np.searchsorted(uni, x["key"])] = x["count"]
            assert tc[:, j].tobytes() == col.tobytes(), ("table column", j, tag)
        if len(lists[0]):
            pk, pc = ctx.union_table(dev, probe=True)
            _, pp = ctx.union_table(dev, probe=True, presence=True)
            assert pk.tobytes() == lists[0]["key"].tobytes(), ("probe keys", tag)
            for j, x in enumerate(lists):
                if len(x):
                    idx = np.searchsorted(x["key"], lists[0]["key"])
                    idx[idx == len(x)] = 0
                    hit = x["key"][idx] == lists[0]["key"]
                    cnt = np.where(hit, x["count"][idx], 0).astype(np.uint32)
                else:
                    hit = np.zeros(len(lists[0]), dtype=bool)
                    cnt = np.zeros(len(lists[0]), dtype=np.uint32)
                assert pc[:, j].tobytes() == cnt.tobytes(), ("probe column", j, tag)
                assert pp[:, j].tobytes() == hit.astype(np.uint32).tobytes(), ("membership column", j, tag)
        n_tables += 1
    for d in dev:
        d.free()
    n_cases += 1
ctx.set_option("kway_vt", 0); ctx.set_option("kway_g", 0); ctx.set_option("kway_max", 32)
print("soak_nway: %d cases (%d with their three count tables) in %.0f s, seed %d: all equal to the oracle (kway_calls %d, splits %d, overflows %d)" % (
    n_cases, n_tables, time.time() - t0, seed, ctx.get_counter("kway_calls"), ctx.get_counter("kway_splits"), ctx.get_counter("kway_overflows")))
ctx.close()
