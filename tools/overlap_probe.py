"""Do two N-way unions in flight hide each other's sampling and partition?  (the call's 3.3 ms in
front of the tile kernel.)  Two contexts on ONE device, each with its own eight lists and its own stream, each running
gt4hip_union_multi in a loop from its own host thread (ctypes releases the GIL): the steps in front of one call's tile
kernel can run beside the other call's tile kernel only if the device lets them (the tile kernel's workgroups hold all of a
CU's wavefront slots' registers but 96 per SIMD and all LDS but 36 KB).  Prints ms per call, one thread and two.
GPU box; not part of the test-suite.   python tools/overlap_probe.py [entries per list] [calls per thread]"""
import os
import sys
import threading
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genometester4_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctxs = [capi.Context(0), capi.Context(0)]
lists = [synth.make_lists8(c, n, 25, "stride", 8) for c in ctxs]
outs = []
for c, ls in zip(ctxs, lists):  # the output list of every context, allocated once
    rc, nw, tot, out = c.union_multi(ls, 1, 1, 1)
    assert rc == 0
    outs.append((out, nw, tot))


def loop(i, m, res):
    c, ls = ctxs[i], lists[i]
    out, nw, tot = outs[i]
    for _ in range(m):
        rc, n2, t2, _o = c.union_multi(ls, 1, 1, 1, False, out)
        assert rc == 0 and (n2, t2) == (nw, tot)
    c.synchronize()
    res[i] = True


def run(threads):
    res = {}
    ts = [threading.Thread(target=loop, args=(i, calls, res)) for i in range(threads)]
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    assert len(res) == threads
    return dt * 1e3 / (calls * threads)


run(1)
a = run(1)
b = run(2)
a2 = run(1)
b2 = run(2)
print("8 x %d-entry lists, %d calls per thread: one call at a time %.2f / %.2f ms per call; two contexts in flight %.2f / %.2f ms per call" % (n, calls, a, a2, b, b2))
for c in ctxs:
    c.close()
