"""Experiment driver (not product): 8-way union through the C ABI only (no torch buffer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
ctx = capi.Context(0)
ctx.set_option("kway", int(os.environ.get("KWAY", "1")))
lists = []
for j in range(8):
    l = ctx.alloc(n, 25)
    shared = j % 2 == 0
    ctx.generate_ex(l, n, 7 if shared else 100 + j, 50 + j, 8, 16, 0 if shared else 1 + j)
    lists.append(l)
print("generated", flush=True)
for vt in [int(x) for x in os.environ.get("KWAY_VT", "0").split(",")]:
  ctx.set_option("kway_vt", vt)
  print("kway_vt", vt, flush=True)
  for rep in range(2):
    rc, nw, tot, out = ctx.union_multi(lists)
    print("  rep", rep, rc, nw, tot, "device ms %.2f" % ctx.last_multi_device_ms, "sorted", out.is_sorted(), flush=True)
    out.free()
