"""Experiment (not product): host<->device transfer rates through the C ABI with pageable buffers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t0 = time.perf_counter()
from genometester4_amd import capi
from genometester4_amd.listio import RECORD_DTYPE
t1 = time.perf_counter()
ctx = capi.Context(0)
t2 = time.perf_counter()
print("import %.3f s, context %.3f s" % (t1 - t0, t2 - t1))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
d = ctx.alloc(n, 25); ctx.generate(d, n, 1, 8)
host = d.download()
for rep in range(3):
    t = time.perf_counter(); u = ctx.upload(host, 25); ctx.synchronize(); dt = time.perf_counter() - t
    print("upload   %.2f GB in %.3f s = %.1f GB/s" % (12 * n / 1e9, dt, 12 * n / dt / 1e9))
    t = time.perf_counter(); h2 = u.download(); dt = time.perf_counter() - t
    print("download %.2f GB in %.3f s = %.1f GB/s" % (12 * n / 1e9, dt, 12 * n / dt / 1e9))
    u.free()
assert h2.tobytes() == host.tobytes()
