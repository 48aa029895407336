"""Experiment driver (not product): throughput of gt4hip_words_to_list (sort + fold) on random words."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rng = np.random.default_rng(1)
words = rng.integers(0, 1 << (2 * k if k < 32 else 63), size=n, dtype=np.uint64)
words[::7] = words[::7][0]  # a heavy hitter
ctx = capi.Context(0)
for rep in range(3):
    t0 = time.perf_counter()
    lst = ctx.words_to_list(words, k)
    dt = time.perf_counter() - t0
    print("rep %d: %d words k=%d -> %d distinct in %.3f s (%.1f M words/s, upload included)" % (rep, n, k, lst.n_words, dt, n / dt / 1e6), flush=True)
    lst.free()
