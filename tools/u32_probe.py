"""diagnostic: 32 lists with different amounts of sharing, one pass (kway_max 32) against levels of eight (kway_max 8)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genometester4_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
ctx = capi.Context(0)
ctx.set_option("kway", 3)
for every in (2, 8, 32, 0):  # every `every`-th list holds the shared key set (0: none does)
    lists = []
    for j in range(32):
        lst = ctx.alloc(n, 25)
        shared = every and j % every == 0
        ctx.generate_ex(lst, n, 7 if shared else 100 + j, 50 + j, 8, 64, 0 if shared else 1 + j)
        lists.append(lst)
    out = ctx.alloc(32 * n, 25)
    res = {}
    for kmax in (33, 8, 32):
        ctx.set_option("kway_max", kmax)
        ms = []
        for i in range(4):
            rc, nw, t, _ = ctx.union_multi(lists, 1, 0, 1, out=out)
            ms.append(ctx.last_multi_device_ms)
        res[kmax] = (min(ms[1:]), nw, t, ctx.get_counter("kway_splits"), ctx.get_counter("nway_tiles"))
    assert res[33][1:3] == res[8][1:3] == res[32][1:3]
    print("shared by every %d-th list: one pass %.1f ms (tiles %d, cut in two %d), levels of eight %.1f ms, the library's choice %.1f ms (a key lies in %.2f lists); %d records out" % (every, res[33][0], res[33][4], res[33][3], res[8][0], res[32][0], ctx.get_counter("kway_shared_x100") / 100.0, res[32][1]))
    for l in lists + [out]:
        l.free()
