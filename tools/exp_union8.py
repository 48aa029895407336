"""Experiment driver (not product): 8-way union through the C ABI only -- the pairwise tree against the
one-pass N-way kernel on the bench's lists.  usage: exp_union8.py [n per list] [lists]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = capi.Context(0)
for opt in ("scan_group", "dynamic", "spin_limit"):
    if os.environ.get(opt.upper()):
        ctx.set_option(opt, int(os.environ[opt.upper()]))
lists = []
for j in range(nl):
    l = ctx.alloc(n, 25)
    shared = j % 2 == 0
    ctx.generate_ex(l, n, 7 if shared else 100 + j, 50 + j, 8, 16, 0 if shared else 1 + j)
    lists.append(l)
print("generated", nl, "x", n, flush=True)
ref = None
for kway in [int(x) for x in os.environ.get("KWAY", "0,1").split(",")]:
    ctx.set_option("kway", kway)
    for rep in range(int(os.environ.get("REPS", "3"))):
        rc, nw, tot, out = ctx.union_multi(lists)
        head, tail = out.download_range(0, min(nw, 200000)).tobytes(), out.download_range(max(0, nw - 200000), min(nw, 200000)).tobytes()
        sig = (rc, nw, tot, hash(head), hash(tail))
        if ref is None:
            ref = sig
        print("kway", kway, "rep", rep, rc, nw, tot, "device ms %.2f" % ctx.last_multi_device_ms,
              "nway kernel ms %.2f tiles %d" % (ctx.get_counter("nway_kernel_us") / 1000.0, ctx.get_counter("nway_tiles")) if kway else "",
              "sorted", out.is_sorted(), "same as first" if sig == ref else "DIFFERS from the first run", flush=True)
        out.free()
print("fallbacks", ctx.get_counter("single_pass_fallbacks"), "overflow retries", ctx.get_counter("kway_overflows"))
