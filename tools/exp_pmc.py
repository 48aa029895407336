"""Experiment driver (not product): merge variants with distinct kernel names, for rocprofv3 --pmc."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
from bench import build_lists

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
ctx = capi.Context(0)
a, b = build_lists(ctx, capi, n, 25, 0)
out_i = ctx.alloc(n, 25)
out_u = ctx.alloc(2 * n, 25)


def run(tag, ops, out=None, count_only=False, **opts):
    for k, v in opts.items():
        ctx.set_option(k, v)
    st, _, t = ctx.compare(a, b, ops, out=out, count_only=count_only)
    print("%-28s merge %.2f ms" % (tag, t["merge_kernel_ms"]), {k: v[0] for k, v in st.items()}, flush=True)
    for k in opts:
        ctx.set_option(k, 0)


run("intersect g1 single", 2, {2: out_i})
run("intersect g1 two_pass", 2, {2: out_i}, two_pass=1)
run("intersect g1 count", 2, count_only=True, geom1=1)
run("union g1 single", 1, {1: out_u})
run("diff1 g1 single", 4, {4: out_i})
run("union+intersect g1 single", 3, {1: out_u, 2: out_i})
