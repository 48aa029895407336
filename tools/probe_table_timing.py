"""Timing of gt4hip_probe_table_ex (the count table restricted to the keys of list 0: gt4_is_union / search_lists_multi,
reference src/set-operations.c:185-228, src/glistquery.c:776-812) at the sizes of bench.py --workload table (GPU box; not
part of the test-suite).   python tools/probe_table_timing.py [lists entries]..."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genometester4_amd import capi

args = [int(a) for a in sys.argv[1:]] or [6, 100_000_000, 32, 20_000_000]
ctx = capi.Context(0)
for nl, n in zip(args[0::2], args[1::2]):
    lists = []
    for j in range(nl):
        lst = ctx.alloc(n, 25)
        shared = j % 2 == 0
        ctx.generate_ex(lst, n, 7 if shared else 100 + j, 50 + j, 8, 16 if nl < 16 else 64, 0 if shared else 1 + j)
        lists.append(lst)
    arr = (C.c_void_p * nl)(*[l.h for l in lists])
    for km in (32, 8):
        ctx.set_option("kway_max", km)
        if km == 8 and nl <= 8:
            continue
        for presence in (0, 1):
            ts = []
            for it in range(6):
                t = capi.CountTable()
                ctx.synchronize()
                t0 = time.perf_counter()
                assert capi.lib().gt4hip_probe_table_ex(ctx.h, arr, nl, presence, C.byref(t)) == 0
                ctx.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
                rows = t.n_keys
                capi.lib().gt4hip_table_free(C.byref(t))
            ms = sum(ts[2:]) / len(ts[2:])
            alg = 12 * nl * n + (8 + 4 * nl) * rows
            print("probe table, %2d lists x %d, presence %d, kway_max %d: %.2f ms, %d rows, %.3f of 8 TB/s" % (nl, n, presence, km, ms, rows, alg / ms / 1e6 / 8000))
    ctx.set_option("kway_max", 32)
    for l in lists:
        l.free()
ctx.close()
