#!/usr/bin/env python3
"""Experiment driver (not product): one script, sub-commands.  Everything goes through the C ABI (capi).
(Rounds 1-3 kept one small script per experiment -- exp_union8.py, exp_dist.py, exp_modes.py, exp_one.py, exp_c5.py,
exp_skew.py, exp_sort.py, exp_table.py ...: their profiles' READMEs name them; the sub-commands below are the same
drivers.)

  exp.py union8 [--n N] [--lists L] [--dist D] [--kway 0,1] [--reps R] [--g G] [--count-only]
        N-way union of the bench's lists: pairwise tree (kway 0) against the library's choice (1) / the one-pass
        tile kernel forced (3); totals and a hash of head and tail must agree between the runs
  exp.py dists [--n N] [--n8 N8] [--dists a,b,..]
        the pair kernels (intersection, union, -u -d) and the 8-way union (tree and one pass) on every key
        distribution of genometester4_amd/synth.py
  exp.py pair [--n N] [--dist D] [--ops 2] [--cutoff 1] [--reps R] [--count-only]
        one pair operation, merge kernel time per repetition (for rocprofv3 --pmc passes)
  exp.py modes [--n N]
        the merge kernel's variants on the bench pair: single pass / two pass / count only, per output set
  exp.py skew [--na N] [--nb N]
        lists of very different length, both orders, -i / -u / -d
  exp.py c5 [--n N]
        BASELINE config 4 shape on ONE GPU: k = 32 keys over the whole 64-bit range, 2 x 4e9 entries, identities
  exp.py sort [--n N] [--k K] [--reps R]
        gt4hip_device_words_to_list on random words: sort and fold times
  exp.py table [--n N]
        count tables of six lists by the N-way tile kernel (kway 1) and by merges (kway 0)

Environment: SCAN_GROUP, DYNAMIC, SPIN_LIMIT, KWAY_G, KWAY_VT, GRID, GEOM0, GEOM1, TWO_PASS set the library's options;
GT4HIP_LIB selects another build of the library (e.g. `make -C genometester4_amd/csrc prof`: libgt4hip_prof.so).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi, synth  # noqa: E402


def _opts(ctx):
    for opt in ("scan_group", "dynamic", "spin_limit", "kway_g", "kway_vt", "grid", "geom0", "geom1", "two_pass"):
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


def time_nway(ctx, tag, lists, kway, reps=3, count_only=False):
    ctx.set_option("kway", kway)
    sig = None
    if count_only:
        for _ in range(reps):
            r = ctx.union_multi(lists, count_only=True)
        one = ctx.get_counter("nway_one_pass")
        print("%-44s kway %d -> %-8s count only: call %7.2f ms (tile kernel %6.2f)  %s" % (tag, kway, "one pass" if one else "tree", ctx.last_multi_device_ms,
              ctx.get_counter("nway_kernel_us") / 1000.0 if one else 0.0, r[:3]), flush=True)
        return ctx.last_multi_device_ms, tuple(r[:3])
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
        _, sig = time_nway(ctx, "%s %d x %d" % (a.dist, a.lists, a.n), lists, kway, a.reps, a.count_only)
        if ref is None:
            ref = sig
        print("   ", "same as the first run" if sig == ref else "DIFFERS from the first run: %s vs %s" % (sig, ref), flush=True)
    ctx.close()


def cmd_pair(a):
    ctx = capi.Context(0)
    _opts(ctx)
    x, y = synth.make_pair(ctx, a.n, 25, a.dist)
    for _ in range(a.reps):
        if a.count_only:
            st, _, t = ctx.compare(x, y, a.ops, cutoff=a.cutoff, count_only=True)
            print("count only ops %d: merge %.3f ms  device %.3f ms  tiles %d" % (a.ops, t["merge_kernel_ms"], t["device_ms"], t["merge_tiles"]), st, flush=True)
        else:
            time_pair(ctx, "%s ops %d cutoff %d  %d x %d" % (a.dist, a.ops, a.cutoff, x.n_words, y.n_words), x, y, a.ops, a.cutoff, 1)
    ctx.close()


def cmd_modes(a):
    ctx = capi.Context(0)
    _opts(ctx)
    x, y = synth.make_pair(ctx, a.n, 25, "stride")
    out_i, out_u = ctx.alloc(a.n, 25), ctx.alloc(2 * a.n, 25)

    def run(tag, ops, out=None, count_only=False, two_pass=0, reps=3):
        if two_pass and os.environ.get("GT4_SKIP_TWO_PASS"):
            return
        ctx.set_option("two_pass", two_pass)
        ms = []
        for _ in range(reps):
            st, _, t = ctx.compare(x, y, ops, out=out, count_only=count_only)
            ms.append(t["merge_kernel_ms"])
        print("%-28s merge %.2f ms  -> %.1f G rec/s  (n_out %s)" % (tag, min(ms), 2 * a.n / min(ms) / 1e6, {k: v[0] for k, v in st.items()}), flush=True)

    run("intersect lookback", 2, {2: out_i})
    run("intersect two_pass", 2, {2: out_i}, two_pass=1)
    run("intersect count_only", 2, count_only=True)
    run("union lookback", 1, {1: out_u})
    run("union two_pass", 1, {1: out_u}, two_pass=1)
    run("union count_only", 1, count_only=True)
    run("all4 count_only", 15, count_only=True)
    run("diff1 lookback", 4, {4: out_i})
    run("diff2 lookback", 8, {8: out_i})
    run("union+intersect lookback", 3, {1: out_u, 2: out_i})
    ctx.close()


def cmd_skew(a):
    ctx = capi.Context(0)
    x = ctx.alloc(a.na, 25)
    ctx.generate_ex(x, a.na, 5, 50, 8, 1, 0)
    y = ctx.alloc(a.nb, 25)
    ctx.generate_ex(y, a.nb, 6, 51, 8, 1, 0)
    for p, q, tag in ((x, y, "big,small"), (y, x, "small,big")):
        for ops, name in ((2, "intersect"), (1, "union"), (4, "diff1")):
            time_pair(ctx, "%-10s %-9s" % (tag, name), p, q, ops, reps=2)
    ctx.close()


def cmd_c5(a):
    ctx = capi.Context(0)
    n = a.n
    x = ctx.alloc(n, 32)
    ctx.generate_ex(x, n, 5, 50, 8, 1, 0)
    y = ctx.alloc(n, 32)
    ctx.generate_ex(y, n, 5, 51, 8, 1, 0)  # same keys, other counts
    print("generated", x.get_word(n - 1), flush=True)
    out = {2: ctx.alloc(n, 32)}
    for _ in range(2):
        st, o, t = ctx.compare(x, y, 2, out=out)
        print("k=32 intersection 2 x %d: merge %.2f ms, %s -> %.1f G k-mers/s, %.2f TB/s algorithmic" % (n, t["merge_kernel_ms"], st, 2 * n / t["merge_kernel_ms"] / 1e6, 36 * n / t["merge_kernel_ms"] / 1e9), flush=True)
    assert st[2][0] == n and o[2].is_sorted()
    print("identities ok: n_out = n, sorted; total", st[2][1], "<= min-sum bound", min(x.sum_counts(), y.sum_counts()))
    for i in (0, (1 << 32) - 1 if n > (1 << 32) else n // 2, n - 1):
        ka, ca = x.get_word(i)
        kb, cb = y.get_word(i)
        ko, co = o[2].get_word(i)
        assert ka == kb == ko and co == min(ca, cb), (i, ka, kb, ko, ca, cb, co)
    print("spot checks ok")
    ctx.close()


def cmd_sort(a):
    import torch
    ctx = capi.Context(0)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    pristine = torch.randint(0, 1 << min(2 * a.k, 62), (a.n,), dtype=torch.int64, device="cuda", generator=g)
    work = torch.empty_like(pristine)
    for _ in range(a.reps):
        work.copy_(pristine)
        torch.cuda.synchronize()
        lst = ctx.device_words_to_list(work.data_ptr(), a.n, a.k)
        print("n", a.n, "k", a.k, "sort ms %.2f fold ms %.2f" % (ctx.get_counter("sort_us") / 1000.0, ctx.get_counter("fold_us") / 1000.0), "records", lst.n_words,
              "sorted", lst.is_sorted(), "sum", lst.sum_counts() == a.n, flush=True)
        lst.free()
    ctx.close()


def cmd_table(a):
    import ctypes as C
    import time
    ctx = capi.Context(0)
    lists = synth.make_lists8(ctx, a.n, 25, "stride", 6)
    arr = (C.c_void_p * len(lists))(*[l.h for l in lists])
    for kway in (1, 0):
        ctx.set_option("kway", kway)
        for what in ("union", "probe", "membership"):
            for _ in range(3):
                t = capi.CountTable()
                ctx.synchronize()
                t0 = time.perf_counter()
                if what == "union":
                    rc = capi.lib().gt4hip_union_table(ctx.h, arr, len(lists), C.byref(t))
                else:
                    rc = capi.lib().gt4hip_probe_table_ex(ctx.h, arr, len(lists), 1 if what == "membership" else 0, C.byref(t))
                ctx.synchronize()
                dt = time.perf_counter() - t0
                nk = t.n_keys
                capi.lib().gt4hip_table_free(C.byref(t))
            print("kway", kway, what, "rc", rc, "keys", nk, "%.2f ms" % (dt * 1e3), flush=True)
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
    p.add_argument("--count-only", action="store_true")
    p = sub.add_parser("pair")
    p.add_argument("--n", type=int, default=2_000_000_000)
    p.add_argument("--dist", default="stride")
    p.add_argument("--ops", type=int, default=2)
    p.add_argument("--cutoff", type=int, default=1)
    p.add_argument("--reps", type=int, default=3)
    p.add_argument("--count-only", action="store_true")
    p = sub.add_parser("modes")
    p.add_argument("--n", type=int, default=500_000_000)
    p = sub.add_parser("skew")
    p.add_argument("--na", type=int, default=2_000_000_000)
    p.add_argument("--nb", type=int, default=20_000_000)
    p = sub.add_parser("c5")
    p.add_argument("--n", type=int, default=4_000_000_000)
    p = sub.add_parser("sort")
    p.add_argument("--n", type=int, default=1_000_000_000)
    p.add_argument("--k", type=int, default=25)
    p.add_argument("--reps", type=int, default=3)
    p = sub.add_parser("table")
    p.add_argument("--n", type=int, default=100_000_000)
    p = sub.add_parser("dists")
    p.add_argument("--n", type=int, default=1_000_000_000)
    p.add_argument("--n8", type=int, default=250_000_000)
    p.add_argument("--dists", default=",".join(synth.DISTS))
    a = ap.parse_args()
    {"union8": cmd_union8, "pair": cmd_pair, "dists": cmd_dists, "modes": cmd_modes, "skew": cmd_skew, "c5": cmd_c5, "sort": cmd_sort, "table": cmd_table}[a.cmd](a)


if __name__ == "__main__":
    main()
