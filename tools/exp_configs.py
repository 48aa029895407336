"""Experiment driver (not product): BASELINE configs 2 and 4 on one GPU (timing + identities)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
from bench import build_lists

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000_000
ctx = capi.Context(0)
a, b = build_lists(ctx, capi, n, 25, 0)
out = {1: ctx.alloc(2 * n, 25), 4: ctx.alloc(n, 25)}
for rep in range(3):
    st, _, t = ctx.compare(a, b, 1 | 4, cutoff=2, out=out)
    print("config 2 (union + diff1, cutoff 2): merge %.2f ms device %.2f ms" % (t["merge_kernel_ms"], t["device_ms"]), st, flush=True)
nu, nd = st[1][0], st[4][0]
alg = 12 * (2 * n) + 12 * (nu + nd)
print("  algorithmic %.1f GB -> %.2f TB/s; %.1f G k-mers/s" % (alg / 1e9, alg / t["merge_kernel_ms"] / 1e9, 2 * n / t["merge_kernel_ms"] / 1e6))
del out, a, b
# config 4 per-GPU share: k=32 full 64-bit keys
m = n // 4
a = ctx.alloc(m, 32); ctx.generate(a, m, 11, 8)
b = ctx.alloc(m, 32); ctx.generate(b, m, 12, 8)
oi = {2: ctx.alloc(m, 32)}
for rep in range(3):
    st, _, t = ctx.compare(a, b, 2, out=oi)
    print("config 4 share (k=32 intersection, 2 x %d): merge %.2f ms" % (m, t["merge_kernel_ms"]), st, flush=True)
print("  %.1f G k-mers/s" % (2 * m / t["merge_kernel_ms"] / 1e6))
