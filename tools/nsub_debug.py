"""debug helper (not product): one N-way union against the oracle, details of the first difference"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gpu_util as U
import oracle_lib as O
from genometester4_amd import capi

ctx = capi.Context(0)
ctx.set_option("kway", 3)
n_lists, universe = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(7)
keys = np.unique(rng.integers(0, 1 << 40, size=universe, dtype=np.uint64))
lists = []
for j in range(n_lists):
    m = rng.random(len(keys)) < rng.uniform(0.05, 0.9)
    lists.append(U.make_records(keys[m], rng.integers(0, 7, size=int(m.sum()), dtype=np.uint32)))
dev = [ctx.upload(x, 20) for x in lists]
rc_o, n_o, t_o, r_o = O.union_multi(lists, 1, 0, 5)
for rep in range(3):
    rc_g, n_g, t_g, out = ctx.union_multi(dev, 1, 0, 5)
    got = out.download()
    print("rep", rep, "oracle", n_o, t_o, "gpu", n_g, t_g, "tiles", ctx.get_counter("nway_tiles"), "one_pass", ctx.get_counter("nway_one_pass"), "fallbacks", ctx.get_counter("single_pass_fallbacks"))
    if got.tobytes() != r_o.tobytes():
        m = min(len(got), len(r_o))
        d = np.nonzero((got["key"][:m] != r_o["key"][:m]) | (got["count"][:m] != r_o["count"][:m]))[0]
        print("  first diff at", d[:5], "of", m, "len got", len(got), "len exp", len(r_o))
        if len(d):
            i = int(d[0])
            print("  got", got[max(0, i - 2):i + 3], "\n  exp", r_o[max(0, i - 2):i + 3])
    rc_c, n_c, t_c, _ = ctx.union_multi(dev, 1, 0, 5, True)
    print("  count-only", n_c, t_c)
