"""Experiment driver (not product): the three bench workloads' kernels in one process (kernel ms)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
from bench import build_lists
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000_000
ctx = capi.Context(0)
a, b = build_lists(ctx, capi, n, 25, 0)
oi = {2: ctx.alloc(n, 25)}
for rep in range(4):
    st, _, t = ctx.compare(a, b, 2, out=oi)
print("intersect: merge %.2f ms device %.2f ms" % (t["merge_kernel_ms"], t["device_ms"]), st[2], flush=True)
oi[2].free()
out = {1: ctx.alloc(2 * n, 25), 4: ctx.alloc(n, 25)}
for rep in range(4):
    st, _, t = ctx.compare(a, b, 1 | 4, cutoff=3, out=out)
print("c2 (-u -d -c 3): merge %.2f ms device %.2f ms" % (t["merge_kernel_ms"], t["device_ms"]), st, flush=True)
out[4].free()
for rep in range(3):
    st, _, t = ctx.compare(a, b, 1, out={1: out[1]})
print("union: merge %.2f ms device %.2f ms" % (t["merge_kernel_ms"], t["device_ms"]), st[1], flush=True)
out[1].free(); a.free(); b.free()
lists = []
for j in range(8):
    l = ctx.alloc(n // 4, 25)
    shared = j % 2 == 0
    ctx.generate_ex(l, n // 4, 7 if shared else 100 + j, 50 + j, 8, 16, 0 if shared else 1 + j)
    lists.append(l)
for rep in range(3):
    rc, nw, tot, o = ctx.union_multi(lists)
    o.free()
print("union8: device %.2f ms kernel %.2f ms" % (ctx.last_multi_device_ms, ctx.get_counter("nway_kernel_us") / 1000.0), (nw, tot), "fallbacks", ctx.get_counter("single_pass_fallbacks"), flush=True)
