"""Experiment driver (not product): count tables of six 1e8-entry lists -- union table and the table
restricted to the keys of list 0 -- by the N-way tile kernel (kway 1) and by merges (kway 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from genometester4_amd import capi
from genometester4_amd.capi import lib, CountTable
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = capi.Context(0)
lists = []
for j in range(6):
    l = ctx.alloc(n, 25)
    shared = j % 2 == 0
    ctx.generate_ex(l, n, 7 if shared else 100 + j, 50 + j, 8, 16, 0 if shared else 1 + j)
    lists.append(l)
arr = (C.c_void_p * len(lists))(*[l.h for l in lists])
for kway in (1, 0):
    ctx.set_option("kway", kway)
    for what in ("union", "probe", "membership"):
        for rep in range(3):
            t = CountTable()
            ctx.synchronize()
            t0 = time.perf_counter()
            if what == "union":
                rc = lib().gt4hip_union_table(ctx.h, arr, len(lists), C.byref(t))
            else:
                rc = lib().gt4hip_probe_table_ex(ctx.h, arr, len(lists), 1 if what == "membership" else 0, C.byref(t))
            ctx.synchronize()
            dt = time.perf_counter() - t0
            nk = t.n_keys
            lib().gt4hip_table_free(C.byref(t))
        print("kway", kway, what, "rc", rc, "keys", nk, "%.2f ms" % (dt * 1e3), flush=True)
