"""Experiment driver (not product): time the merge kernel's variants on the bench workload."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
from bench import build_lists

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
ctx = capi.Context(0)
for opt in ("geom0", "geom1", "grid", "scan_group", "dynamic"):
    if os.environ.get("GT4_" + opt.upper()):
        ctx.set_option(opt, int(os.environ["GT4_" + opt.upper()]))
a, b = build_lists(ctx, capi, n, 25, 0)
out_i = ctx.alloc(n, 25)
out_u = ctx.alloc(2 * n, 25)


def run(tag, ops, out=None, count_only=False, two_pass=0, reps=3):
    if two_pass and os.environ.get("GT4_SKIP_TWO_PASS"):
        return
    ctx.set_option("two_pass", two_pass)
    ms = []
    for _ in range(reps):
        st, _, t = ctx.compare(a, b, ops, out=out, count_only=count_only)
        ms.append(t["merge_kernel_ms"])
    recs = 2 * n
    print("%-28s merge %.2f ms  -> %.1f G rec/s  (n_out %s)" % (tag, min(ms), recs / min(ms) / 1e6, {k: v[0] for k, v in st.items()}), flush=True)


run("intersect lookback", 2, {2: out_i})
run("intersect two_pass", 2, {2: out_i}, two_pass=1)
run("intersect count_only", 2, count_only=True)
run("union lookback", 1, {1: out_u})
run("union two_pass", 1, {1: out_u}, two_pass=1)
run("union count_only", 1, count_only=True)
run("all4 count_only", 15, count_only=True)
run("diff1 lookback", 4, {4: out_i})
run("diff2 lookback", 8, {8: out_i})
run("union+intersect lookback", 3, {1: out_u, 2: out_i})
