"""Experiment (not product): lists of very different length (2e9 vs 2e7), both orders, -i / -u / -d."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
ctx = capi.Context(0)
na, nb = 2_000_000_000, 20_000_000
a = ctx.alloc(na, 25); ctx.generate_ex(a, na, 5, 50, 8, 1, 0)
b = ctx.alloc(nb, 25); ctx.generate_ex(b, nb, 6, 51, 8, 1, 0)
for x, y, tag in ((a, b, "big,small"), (b, a, "small,big")):
    for ops, name in ((2, "intersect"), (1, "union"), (4, "diff1")):
        out = {ops: ctx.alloc(x.n_words + y.n_words if ops == 1 else x.n_words, 25)}
        for rep in range(2):
            st, o, t = ctx.compare(x, y, ops, out=out)
        n_in = x.n_words + y.n_words
        alg = 12 * (n_in + st[ops][0])
        print("%-10s %-9s merge %.2f ms  %.1f G k-mers/s  %.2f TB/s  n_out %d" % (tag, name, t["merge_kernel_ms"], n_in / t["merge_kernel_ms"] / 1e6, alg / t["merge_kernel_ms"] / 1e9, st[ops][0]), flush=True)
        assert o[ops].is_sorted()
        del out, o
