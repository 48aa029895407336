"""diagnostic (not product): the 8-way union's tile kernel in its count-only form (no chained scan, no stores) and
in its union form, at the bench's size -- what the kernel costs without the output side."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genometester4_amd import capi, synth

n8 = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
dist = sys.argv[2] if len(sys.argv) > 2 else "stride"
ctx = capi.Context(0)
ctx.set_option("kway", 3)
full = synth.make_lists8(ctx, n8, 25, dist, 8)
out = ctx.alloc(sum(l.n_words for l in full), 25)
for name, co in (("count-only", True), ("union", False)):
    ms = []
    for i in range(5):
        rc, n, t, res = ctx.union_multi(full, 1, 0, 1, co, out=None if co else out)
        ms.append(ctx.get_counter("nway_kernel_us") / 1000.0)
    print("%s: kernel ms %s  n %d total %d  tiles %d" % (name, " ".join("%.2f" % x for x in ms[1:]), n, t, ctx.get_counter("nway_tiles")))
