"""Experiment driver (not product): the union of the bench's two lists by the pair kernel and by the
one-pass N-way tile kernel (option kway = 2 sends two lists there too)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
from bench import build_lists
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000_000
ctx = capi.Context(0)
a, b = build_lists(ctx, capi, n, 25, 0)
out = ctx.alloc(2 * n, 25)
for rep in range(3):
    st, _, t = ctx.compare(a, b, 1, out={1: out})
    print("pair kernel union: merge %.2f ms device %.2f ms" % (t["merge_kernel_ms"], t["device_ms"]), st[1], flush=True)
ctx.set_option("kway", 2)
for rep in range(3):
    rc, nw, tot, o = ctx.union_multi([a, b], out=out)
    print("N-way kernel, two lists: device %.2f ms kernel %.2f ms" % (ctx.last_multi_device_ms, ctx.get_counter("nway_kernel_us") / 1000.0), (nw, tot), flush=True)
