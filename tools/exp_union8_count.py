"""Experiment driver (not product): the one-pass N-way union count-only (no staging, no chain) on the bench's lists."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
ctx = capi.Context(0)
lists = []
for j in range(8):
    l = ctx.alloc(n, 25)
    shared = j % 2 == 0
    ctx.generate_ex(l, n, 7 if shared else 100 + j, 50 + j, 8, 16, 0 if shared else 1 + j)
    lists.append(l)
ctx.set_option("kway", 1)
for rep in range(3):
    rc, nw, tot, out = ctx.union_multi(lists, count_only=True)
    print("count-only rep", rep, rc, nw, tot, "device ms %.2f" % ctx.last_multi_device_ms, "nway kernel ms %.2f" % (ctx.get_counter("nway_kernel_us") / 1000.0), flush=True)
