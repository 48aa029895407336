#!/usr/bin/env python3
"""Experiment driver (not product): one script, sub-commands.  Everything goes through the C ABI (capi).

  exp.py union8 [--n N] [--lists L] [--dist D] [--kway 0,1] [--reps R] [--g G]
        N-way union of the bench's lists: pairwise tree (kway 0) against the library's choice (1) / the one-pass
        tile kernel forced (3); totals and a hash of head and tail must agree between the runs
  exp.py dists [--n N] [--n8 N8] [--dists a,b,..]
        the pair kernels (intersection, union, -u -d) and the 8-way union (tree and one pass) on every key
        distribution of genometester4_amd/synth.py
  exp.py pair [--n N] [--dist D] [--ops 2] [--cutoff 1] [--reps R]
        one pair operation, merge kernel time per repetition
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi, synth  # noqa: E402


def _opts(ctx):
    for opt in ("scan_group", "dynamic", "spin_limit", "kway_g", "kway_vt", "grid"):
        if os.environ.get(opt.upper()):
            ctx.set_option(opt, int(os.environ[opt.upper()]))


def time_pair(ctx, tag, a, b, ops, cutoff=1, reps=3):
    out = {bit: ctx.alloc(max(1, {1: a.n_words + b.n_words, 2: min(a.n_words, b.n_words), 4: a.n_words}[bit]), a.word_length) for bit in (1, 2, 4) if ops & bit}
    for _ in range(reps):
        st, _, t = ctx.compare(a, b, ops, cutoff=cutoff, out=out)
    n_in = a.n_words + b.n_words
    n_out = sum(st[bit][0] for bit in st)
    alg = 12 * (n_in + n_out)
    ms = t["merge_kernel_ms"]
    print("%-44s merge %7.2f ms  %6.1f G k-mers/s  %5.2f TB/s algorithmic = %.3f of 8 TB/s  (out %d)" % (tag, ms, n_in / ms / 1e6, alg / ms / 1e9, alg / ms / 1e9 / 8, n_out), flush=True)
    for o in out.values():
        o.free()
    return ms


def time_nway(ctx, tag, lists, kway, reps=3):
    ctx.set_option("kway", kway)
    sig = None
    for _ in range(reps):
        rc, nw, tot, o = ctx.union_multi(lists)
        m = min(nw, 100000)
        sig = (rc, nw, tot, hash(o.download_range(0, m).tobytes()), hash(o.download_range(nw - m, m).tobytes()))
        o.free()
    n_in = sum(l.n_words for l in lists)
    one = ctx.get_counter("nway_one_pass")
    kms = ctx.get_counter("nway_kernel_us") / 1000.0 if one else 0.0
    ms = ctx.last_multi_device_ms
    alg = 12 * (n_in + nw)
    print("%-44s kway %d -> %-8s call %7.2f ms (tile kernel %6.2f)  %6.1f G k-mers/s  whole call %.3f of 8 TB/s  in %d out %d  fallbacks %d overflows %d" % (
        tag, kway, "one pass" if one else "tree", ms, kms, n_in / ms / 1e6, alg / ms / 1e9 / 8, n_in, nw, ctx.get_counter("single_pass_fallbacks"), ctx.get_counter("kway_overflows")), flush=True)
    return ms, sig


def cmd_union8(a):
    ctx = capi.Context(0)
    _opts(ctx)
    if a.g:
        ctx.set_option("kway_g", a.g)
    lists = synth.make_lists8(ctx, a.n, 25, a.dist, a.lists)
    print("generated", [l.n_words for l in lists], flush=True)
    ref = None
    for kway in [int(x) for x in a.kway.split(",")]:
        _, sig = time_nway(ctx, "%s %d x %d" % (a.dist, a.lists, a.n), lists, kway, a.reps)
        if ref is None:
            ref = sig
        print("   ", "same as the first run" if sig == ref else "DIFFERS from the first run: %s vs %s" % (sig, ref), flush=True)
    ctx.close()


def cmd_pair(a):
    ctx = capi.Context(0)
    _opts(ctx)
    x, y = synth.make_pair(ctx, a.n, 25, a.dist)
    for _ in range(a.reps):
        time_pair(ctx, "%s ops %d cutoff %d  %d x %d" % (a.dist, a.ops, a.cutoff, x.n_words, y.n_words), x, y, a.ops, a.cutoff, 1)
    ctx.close()


def cmd_dists(a):
    import time
    ctx = capi.Context(0)
    _opts(ctx)
    for d in a.dists.split(","):
        t0 = time.time()
        x, y = synth.make_pair(ctx, a.n, 25, d)
        ctx.synchronize()
        print("# %s pair generated in %.1f s: %d + %d records, sorted %s %s" % (d, time.time() - t0, x.n_words, y.n_words, x.is_sorted(), y.is_sorted()), flush=True)
        time_pair(ctx, "%-9s intersect" % d, x, y, 2)
        time_pair(ctx, "%-9s union" % d, x, y, 1)
        time_pair(ctx, "%-9s -u -d -c 3" % d, x, y, 5, 3)
        x.free()
        y.free()
        t0 = time.time()
        lists = synth.make_lists8(ctx, a.n8, 25, d)
        ctx.synchronize()
        print("# %s lists generated in %.1f s: %s" % (d, time.time() - t0, [l.n_words for l in lists]), flush=True)
        ref = None
        for kway in (0, 3, 1):
            _, sig = time_nway(ctx, "%-9s 8-way union" % d, lists, kway)
            ref = ref or sig
            if sig != ref:
                print("    DIFFERS from the tree: %s vs %s" % (sig, ref), flush=True)
        for l in lists:
            l.free()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("union8")
    p.add_argument("--n", type=int, default=500_000_000)
    p.add_argument("--lists", type=int, default=8)
    p.add_argument("--dist", default="stride")
    p.add_argument("--kway", default="0,1")
    p.add_argument("--reps", type=int, default=3)
    p.add_argument("--g", type=int, default=0)
    p = sub.add_parser("pair")
    p.add_argument("--n", type=int, default=2_000_000_000)
    p.add_argument("--dist", default="stride")
    p.add_argument("--ops", type=int, default=2)
    p.add_argument("--cutoff", type=int, default=1)
    p.add_argument("--reps", type=int, default=3)
    p = sub.add_parser("dists")
    p.add_argument("--n", type=int, default=1_000_000_000)
    p.add_argument("--n8", type=int, default=250_000_000)
    p.add_argument("--dists", default=",".join(synth.DISTS))
    a = ap.parse_args()
    {"union8": cmd_union8, "pair": cmd_pair, "dists": cmd_dists}[a.cmd](a)


if __name__ == "__main__":
    main()
