"""Experiment driver (not product): one big pair operation through the C ABI."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
na, nb = int(sys.argv[1]), int(sys.argv[2])
ops = int(sys.argv[3]) if len(sys.argv) > 3 else 1
two = len(sys.argv) > 4 and sys.argv[4] == "two"
count = len(sys.argv) > 4 and sys.argv[4] == "count"
ctx = capi.Context(0)
if two: ctx.set_option("two_pass", 1)
a = ctx.alloc(na, 25); ctx.generate_ex(a, na, 7, 50, 8, 16, 1)
b = ctx.alloc(nb, 25); ctx.generate_ex(b, nb, 8, 51, 8, 16, 2)
print("generated", a.is_sorted(), b.is_sorted(), flush=True)
st, out, t = ctx.compare(a, b, ops, count_only=count)
print("compared", st, t, flush=True)
if not count: print("sorted", out[ops].is_sorted(), flush=True)
