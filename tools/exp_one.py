"""Experiment driver (not product): run one merge configuration a few times (for rocprofv3 --pmc)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
from bench import build_lists

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
ops = int(sys.argv[2]) if len(sys.argv) > 2 else 2
count_only = len(sys.argv) > 3 and sys.argv[3] == "count"
ctx = capi.Context(0)
for opt in ("geom0", "geom1", "grid", "two_pass"):
    if os.environ.get("GT4_" + opt.upper()):
        ctx.set_option(opt, int(os.environ["GT4_" + opt.upper()]))
a, b = build_lists(ctx, capi, n, 25, 0)
if count_only:
    out = None
elif ops in (1, 2, 4, 8):
    out = {ops: ctx.alloc(2 * n if ops == 1 else n, 25)}
else:
    out = {bit: ctx.alloc(2 * n if bit == 1 else n, 25) for bit in (1, 2, 4, 8) if ops & bit}
for _ in range(3):
    st, _, t = ctx.compare(a, b, ops, out=out, count_only=count_only, cutoff=int(os.environ.get("GT4_CUTOFF", "1")))
    print("merge %.3f ms  device %.3f ms  tiles %d" % (t["merge_kernel_ms"], t["device_ms"], t["merge_tiles"]), st, flush=True)
