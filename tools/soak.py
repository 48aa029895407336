"""Randomised parity soak against the CPU oracle (GPU box; not part of the test-suite).
python tools/soak.py [seconds] [seed]"""
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
t0 = time.time(); n_cases = 0
while time.time() - t0 < budget:
    k = int(rng.choice([8, 12, 16, 20, 25, 31, 32]))
    n_u = int(rng.choice([50, 3000, 20000, 150000, 700000, 2500000]))
    if k == 8: n_u = min(n_u, 40000)
    if k == 12: n_u = min(n_u, 3000000)
    p_a, p_b = float(rng.choice([0.02, 0.3, 0.6, 0.95])), float(rng.choice([0.02, 0.3, 0.6, 0.95]))
    a, b = U.random_pair(int(rng.integers(1 << 30)), n_u, p_a, p_b, k=k, max_count=int(rng.choice([1, 3, 8, 0xffffffff])))
    ops = int(rng.choice([1, 2, 4, 8, 3, 5, 6, 9, 12, 15]))
    rule = int(rng.integers(0, 8)); cutoff = int(rng.choice([0, 1, 2, 3, 7])); sub = int(rng.integers(0, 2)); ovr = int(rng.integers(0, 5))
    two = int(rng.integers(0, 5) == 0)
    ctx.set_option("two_pass", two)
    da, db = ctx.upload(a, k), ctx.upload(b, k)
    exp = O.compare(a, b, ops, rule, cutoff, sub, ovr)
    st, out, _ = ctx.compare(da, db, ops, rule, cutoff, sub, ovr)
    for bit, (n, total, recs) in exp.items():
        assert st[bit] == (n, total), ("stats", k, n_u, p_a, p_b, ops, rule, cutoff, sub, ovr, two, bit, st[bit], (n, total))
        assert out[bit].download().tobytes() == recs.tobytes(), ("records", k, n_u, p_a, p_b, ops, rule, cutoff, sub, ovr, two, bit)
    st2, _, _ = ctx.compare(da, db, ops, rule, cutoff, sub, ovr, count_only=True)
    assert st2 == st
    if n_cases % 7 == 0 and len(a) and len(b):
        c, _ = U.random_pair(int(rng.integers(1 << 30)), n_u, 0.5, 0.5, k=k)
        lists = [a, b, c, a[::3]]
        dl = [ctx.upload(x, k) for x in lists]
        for fn_o, fn_g, rules in ((O.union_multi, ctx.union_multi, (0, 1, 4, 7)), (O.intersect_multi, ctx.intersect_multi, (0, 1, 3, 4, 7))):
            r = int(rng.choice(rules))
            rc, n, total, recs = fn_o(lists, cutoff, r, ovr)
            rc2, n2, total2, outl = fn_g(dl, cutoff, r, ovr)
            assert (rc, n, total) == (rc2, n2, total2), ("multi", fn_o.__name__, r, cutoff)
            if rc == 0:
                assert outl.download().tobytes() == recs.tobytes(), ("multi records", fn_o.__name__, r, cutoff)
    n_cases += 1
ctx.close()
print("soak ok: %d random cases in %.0f s (seed %d)" % (n_cases, time.time() - t0, seed))
