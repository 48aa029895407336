"""Does the alignment of the two input lists against each other matter to the headline intersection?  (Both streams advance
at about the same rate; if their bases map to the same HBM channels the channels would see pairs of requests.)  List B is
generated X records longer and its view [X, X + n) is merged: the view's base lies 12 X bytes behind an allocation boundary.
GPU box; not part of the test-suite.   python tools/align_probe.py [entries per list]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genometester4_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000_000
ctx = capi.Context(0)
a = ctx.alloc(n, 25)
ctx.generate_ex(a, n, 7, 50, 8, 2, 0)
out = None
for x in (0, 1, 16, 1024, 5461, 21845, 87381, 349525, 1398101, 44739243):
    b_all = ctx.alloc(n + x, 25)
    ctx.generate_ex(b_all, n + x, 9, 51, 8, 2, 1)
    b = b_all.slice(x, n)
    ts = []
    for it in range(8):
        ctx.synchronize()
        t0 = time.perf_counter()
        stats, lists, timing = ctx.compare(a, b, capi.OP_INTRSEC, 0, 1, out=out)
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        if out is None:
            out = {capi.OP_INTRSEC: lists[capi.OP_INTRSEC]}
    print("B starts %11d bytes behind its allocation: %.3f ms per intersection (min %.3f), %d records out" % (12 * x, sum(ts[3:]) / len(ts[3:]), min(ts), stats[capi.OP_INTRSEC][0]))
    b.free()
    b_all.free()
ctx.close()
