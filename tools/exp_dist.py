"""Experiment driver (not product): the pair kernels and the one-pass N-way kernel on key distributions
other than the bench's near-uniform strides (Next 6):
  uniform    the bench's lists (reference point)
  skew       2e9 against 2e7 records (rank_group's general branch: runs of very different length per tile)
  clustered  stretches of ~3000 adjacent keys 2^40 apart (tiles that span a gap; the N-way kernel's
             interpolation puts a whole stretch into one bucket: its tiles take the search path)
usage: exp_dist.py [records of the long lists]   (run it under rocprofv3 --kernel-trace --stats for the summary)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genometester4_amd import capi
from bench import build_lists

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ctx = capi.Context(0)


def time_pair(tag, a, b, ops):
    out = {bit: ctx.alloc({1: a.n_words + b.n_words, 2: min(a.n_words, b.n_words), 4: a.n_words}[bit], a.word_length) for bit in (1, 2, 4) if ops & bit}
    for rep in range(3):
        st, o, t = ctx.compare(a, b, ops, out=out)
    n_in = a.n_words + b.n_words
    n_out = sum(st[bit][0] for bit in st)
    alg = 12 * (n_in + n_out)
    print("%-34s merge %7.2f ms  %6.1f G k-mers/s  %5.2f TB/s algorithmic = %.3f of 8 TB/s" % (tag, t["merge_kernel_ms"], n_in / t["merge_kernel_ms"] / 1e6,
                                                                                              alg / t["merge_kernel_ms"] / 1e9, alg / t["merge_kernel_ms"] / 1e9 / 8), flush=True)
    for o_ in out.values():
        o_.free()


def time_nway(tag, lists):
    for rep in range(3):
        rc, nw, tot, o = ctx.union_multi(lists)
        o.free()
    n_in = sum(l.n_words for l in lists)
    ms = ctx.get_counter("nway_kernel_us") / 1000.0
    alg = 12 * (n_in + nw)
    print("%-34s tile kernel %7.2f ms (call %.2f ms)  %6.1f G k-mers/s  %.3f of 8 TB/s" % (tag, ms, ctx.last_multi_device_ms, n_in / ms / 1e6, alg / ms / 1e9 / 8), flush=True)


def clustered(m, seed, keep_mod):
    """records of a clustered universe (stretches of 3000 keys, 2^40 apart) whose hash mod 3 != keep_mod"""
    i = torch.arange(m, dtype=torch.int64, device="cuda")
    h = (i * 0x9E3779B97F4A7C15 + seed) ^ ((i * 0x9E3779B97F4A7C15 + seed) >> 29)
    keys = ((i // 3000) << 40) + (i % 3000) * 1000 + (h & 511)
    keep = (h >> 20) % 3 != keep_mod
    keys = keys[keep]
    rec = torch.empty((keys.numel(), 3), dtype=torch.int32, device="cuda")
    rec[:, 0] = (keys & 0xFFFFFFFF).to(torch.int32) if False else ((keys << 32) >> 32).to(torch.int32)
    rec[:, 1] = (keys >> 32).to(torch.int32)
    rec[:, 2] = ((h[keep] >> 7) & 7).to(torch.int32) + 1
    return rec


# ---- uniform (reference point)
a, b = build_lists(ctx, capi, n, 25, 0)
time_pair("uniform   intersect %d x %d" % (n, n), a, b, 2)
time_pair("uniform   union", a, b, 1)
time_pair("uniform   -u -d -c 3 (cutoff 1 here)", a, b, 5)
# ---- skew: the long list against one a hundredth as long (both orders)
s = ctx.alloc(n // 100, 25)
ctx.generate_ex(s, n // 100, 6, 51, 8, 1, 0)
time_pair("skew 100  intersect long, short", a, s, 2)
time_pair("skew 100  intersect short, long", s, a, 2)
time_pair("skew 100  union long, short", a, s, 1)
time_pair("skew 100  diff1 long, short", a, s, 4)
time_pair("skew 100  diff1 short, long", s, a, 4)
a.free(); b.free(); s.free()
# ---- clustered
torch.cuda.synchronize()
ra, rb = clustered(n + n // 2, 1, 0), clustered(n + n // 2, 1, 1)
torch.cuda.synchronize()
ca, cb = ctx.wrap(ra.data_ptr(), ra.shape[0], 31), ctx.wrap(rb.data_ptr(), rb.shape[0], 31)
assert ca.is_sorted() and cb.is_sorted()
time_pair("clustered intersect %d x %d" % (ca.n_words, cb.n_words), ca, cb, 2)
time_pair("clustered union", ca, cb, 1)
time_pair("clustered -u -d", ca, cb, 5)
# ---- N-way: uniform against clustered (eight lists of n / 4)
m = n // 4
lists = []
for j in range(8):
    l = ctx.alloc(m, 25)
    shared = j % 2 == 0
    ctx.generate_ex(l, m, 7 if shared else 100 + j, 50 + j, 8, 16, 0 if shared else 1 + j)
    lists.append(l)
time_nway("uniform   8-way union 8 x %d" % m, lists)
for l in lists:
    l.free()
recs = [clustered(m + m // 2, 7, j % 3) for j in range(8)]
torch.cuda.synchronize()
cl = [ctx.wrap(r.data_ptr(), r.shape[0], 31) for r in recs]
time_nway("clustered 8-way union (search path)", cl)
ctx.set_option("kway", 0)
for rep in range(2):
    rc, nw, tot, o = ctx.union_multi(cl)
    o.free()
print("%-34s pairwise tree: call %.2f ms" % ("clustered 8-way union", ctx.last_multi_device_ms), flush=True)
print("fallbacks", ctx.get_counter("single_pass_fallbacks"))
