"""Experiment (not product): BASELINE config 4 shape on ONE GPU -- k=32 full-range keys, 4e9 entries per list."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000_000
ctx = capi.Context(0)
a = ctx.alloc(n, 32); ctx.generate_ex(a, n, 5, 50, 8, 1, 0)
b = ctx.alloc(n, 32); ctx.generate_ex(b, n, 5, 51, 8, 1, 0)   # same keys, other counts
print("generated", a.get_word(n - 1), flush=True)
out = {2: ctx.alloc(n, 32)}
for rep in range(2):
    st, o, t = ctx.compare(a, b, 2, out=out)
    print("k=32 intersection 2 x %d: merge %.2f ms, %s -> %.1f G k-mers/s, %.2f TB/s algorithmic" % (n, t["merge_kernel_ms"], st, 2 * n / t["merge_kernel_ms"] / 1e6, 36 * n / t["merge_kernel_ms"] / 1e9), flush=True)
assert st[2][0] == n and o[2].is_sorted()
sa, sb = a.sum_counts(), b.sum_counts()
print("identities ok: n_out = n, sorted; total", st[2][1], "<= min-sum bound", min(sa, sb))
for i in (0, (1 << 32) - 1 if n > (1 << 32) else n // 2, n - 1):
    ka, ca = a.get_word(i); kb, cb = b.get_word(i); ko, co = o[2].get_word(i)
    assert ka == kb == ko and co == min(ca, cb), (i, ka, kb, ko, ca, cb, co)
print("spot checks ok")
