"""Synthetic sorted k-mer lists in HBM for the bench and the full-size tests (plumbing, not product).

Four key distributions (`DISTS`); every list is strictly ascending, 12-byte packed records, resident on the device:

  stride     gt4hip_generate_ex: exactly one key in every stride of the key space (the friendliest case
             for anything that interpolates keys; exact sizes, closed-form totals: |A n B| = n / 2)
  iid        independent uniform draws from [0, 4^k): exponential gaps, ties folded (sorted by the
             library's own radix sort, gt4hip_sort_words)
  clustered  stretches of CLUSTER_LEN nearly adjacent keys, the stretches far apart (a tile of a merge
             spans a few stretches and the gaps between them: interpolation fails)
  genomic    canonical k-mers of random 4-letter sequences with planted repeats; the lists of one job
             are the k-mer lists of mutated copies of one ancestor (shared k-mers = k-mers no
             substitution touched), counts = occurrences -- the repo's own sort + fold
             (gt4hip_device_words_to_list) makes the lists, as glistmaker's table step would
             (reference src/sequence.c:116-130 packs 2 bits per base the same way)

`make_pair` mirrors the bench's pair (|A| ~ |B| ~ n, about half of each shared); `make_lists8` mirrors its
eight lists (the four even lists hold the same keys, the four odd lists keys of their own: 5 n distinct
keys from 8 n records).  For `stride` the sizes are exact, for the others within a fraction of a percent.
"""
from __future__ import annotations

import torch

DISTS = ("stride", "iid", "clustered", "genomic")
CHUNK = 1 << 26
CLUSTER_LEN = 3000
_M31 = (1 << 31) - 1


def _hash(x):
    """64-bit mix on int64 tensors (wrapping multiplies, logical shifts by masking)."""
    x = x * -7046029254386353131  # 0x9E3779B97F4A7C15
    x = x ^ ((x >> 29) & ((1 << 35) - 1))
    x = x * -4658895280553007687  # 0xBF58476D1CE4E5B9
    x = x ^ ((x >> 32) & 0xFFFFFFFF)
    return x


def _records(keys, counts):
    rec = torch.empty((keys.numel(), 3), dtype=torch.int32, device=keys.device)
    rec[:, 0] = ((keys << 32) >> 32).to(torch.int32)
    rec[:, 1] = (keys >> 32).to(torch.int32)
    rec[:, 2] = counts.to(torch.int32)
    return rec


class _Builder:
    """Lists cut out of one ascending key universe: key -> class = hash % n_classes, list j holds the
    classes `member[j]`; counts 1..8 from other bits of the hash (different per list)."""

    def __init__(self, ctx, k, n_classes, member, seed):
        self.ctx, self.k, self.n_classes, self.member, self.seed = ctx, k, n_classes, member, seed
        self.sizes = [0] * len(member)
        self.recs = None
        self.offs = None

    def _classes(self, keys):
        h = _hash(keys ^ self.seed)
        return h, (h & _M31) % self.n_classes

    def count(self, keys):
        _, cls = self._classes(keys)
        bc = torch.bincount(cls, minlength=self.n_classes).tolist()
        for j, m in enumerate(self.member):
            self.sizes[j] += sum(bc[c] for c in m)

    def allocate(self):
        self.recs = [torch.empty((max(1, s), 3), dtype=torch.int32, device="cuda") for s in self.sizes]
        self.offs = [0] * len(self.member)

    def fill(self, keys):
        h, cls = self._classes(keys)
        for j, m in enumerate(self.member):
            mask = cls == m[0]
            for c in m[1:]:
                mask |= cls == c
            kk = keys[mask]
            cnt = 1 + (((h[mask] >> (33 + 3 * (j % 8))) ^ (j // 8)) & 7)
            self.recs[j][self.offs[j]: self.offs[j] + kk.numel()] = _records(kk, cnt)
            self.offs[j] += kk.numel()

    def lists(self):
        out = []
        for j, r in enumerate(self.recs):
            assert self.offs[j] == self.sizes[j]
            lst = self.ctx.wrap(r.data_ptr(), self.sizes[j], self.k)
            lst._storage = r  # the tensor owns the memory
            out.append(lst)
        return out


def _universe_chunks(ctx, dist, m, k, seed):
    """Yields ascending, duplicate-free int64 key chunks of a universe of about m keys, in key order; calling
    it twice yields the same chunks."""
    space = 1 << (2 * k if k < 32 else 62)
    if dist == "iid":
        g = torch.Generator(device="cuda")
        g.manual_seed(seed)
        w = torch.empty(m, dtype=torch.int64, device="cuda")
        for s in range(0, m, CHUNK):
            e = min(m, s + CHUNK)
            w[s:e] = torch.randint(0, space, (e - s,), dtype=torch.int64, device="cuda", generator=g)
        torch.cuda.synchronize()
        ctx.sort_words(w.data_ptr(), m, k)
        ctx.synchronize()

        def chunks():
            prev = None
            for s in range(0, m, CHUNK):
                c = w[s: min(m, s + CHUNK)]
                keep = torch.ones(c.numel(), dtype=torch.bool, device="cuda")
                keep[1:] = c[1:] != c[:-1]
                if prev is not None:
                    keep[0] = bool(c[0] != prev)
                prev = c[-1].clone()
                yield c[keep]
        return chunks
    if dist == "clustered":
        n_cl = (m + CLUSTER_LEN - 1) // CLUSTER_LEN
        gap = space // (n_cl + 1)
        step = max(2, min(1024, gap // (64 * CLUSTER_LEN)))  # a stretch spans 1/64 of the distance to the next at most
        assert gap > CLUSTER_LEN * step

        def chunks():
            for s in range(0, m, CHUNK):
                i = torch.arange(s, min(m, s + CHUNK), dtype=torch.int64, device="cuda")
                j = (_hash(i ^ (seed * 7919)) & _M31) % step
                yield (i // CLUSTER_LEN) * gap + (i % CLUSTER_LEN) * step + j
        return chunks
    raise ValueError(dist)


def _from_universe(ctx, dist, m, k, n_classes, member, seed):
    chunks = _universe_chunks(ctx, dist, m, k, seed)
    b = _Builder(ctx, k, n_classes, member, seed)
    for c in chunks():
        b.count(c)
    b.allocate()
    for c in chunks():
        b.fill(c)
    torch.cuda.synchronize()
    return b.lists()


# ---------------------------------------------------------------- genomic

def _kmer_words(seq, k, first, count):
    """canonical k-mer words of positions [first, first + count) of the base tensor (uint8 values 0..3):
    the smaller of the word and its reverse complement, 2 bits per base, first base most significant"""
    fw = torch.zeros(count, dtype=torch.int64, device=seq.device)
    rc = torch.zeros(count, dtype=torch.int64, device=seq.device)
    for j in range(k):
        b = seq[first + j: first + j + count].to(torch.int64)
        fw |= b << (2 * (k - 1 - j))
        rc |= (3 - b) << (2 * j)
    return torch.minimum(fw, rc)


def _genome_list(ctx, seq, k):
    n = seq.numel() - k + 1
    w = torch.empty(n, dtype=torch.int64, device="cuda")
    for s in range(0, n, CHUNK):
        e = min(n, s + CHUNK)
        w[s:e] = _kmer_words(seq, k, s, e - s)
    torch.cuda.synchronize()
    lst = ctx.device_words_to_list(w.data_ptr(), n, k)
    ctx.synchronize()
    del w
    return lst


def _ancestor(length, seed, repeat_frac=0.05, repeat_len=1000):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    seq = torch.empty(length, dtype=torch.uint8, device="cuda")
    for s in range(0, length, CHUNK):
        e = min(length, s + CHUNK)
        seq[s:e] = torch.randint(0, 4, (e - s,), dtype=torch.uint8, device="cuda", generator=g)
    # planted repeats: segments copied to other places (k-mers with more than one occurrence)
    n_rep = int(length * repeat_frac / repeat_len)
    if n_rep and length > 4 * repeat_len:
        # (a handful of gathers, not one copy per repeat: tens of thousands of tiny launches are slow, and rocprofv3's
        # counter passes do not survive them)
        off = torch.arange(repeat_len, dtype=torch.int64, device="cuda")
        step = max(1, (1 << 24) // repeat_len)
        for i in range(0, n_rep, step):
            m = min(step, n_rep - i)
            src = torch.randint(0, length - repeat_len, (m,), generator=g, device="cuda")
            dst = torch.randint(0, length - repeat_len, (m,), generator=g, device="cuda")
            seq[(dst[:, None] + off).reshape(-1)] = seq[(src[:, None] + off).reshape(-1)]
    return seq


def _mutate(seq, rate_inv, seed):
    """substitutions at about one position in rate_inv (every base replaced by a different one)"""
    out = seq.clone()
    for s in range(0, seq.numel(), CHUNK):
        e = min(seq.numel(), s + CHUNK)
        i = torch.arange(s, e, dtype=torch.int64, device="cuda")
        h = _hash(i ^ (seed * 1000003))
        hit = (h & _M31) % rate_inv == 0
        out[s:e] = torch.where(hit, ((seq[s:e].to(torch.int64) + 1 + ((h >> 40) & 1) + ((h >> 41) & 1)) & 3).to(torch.uint8), seq[s:e])
    return out


# ---------------------------------------------------------------- the bench's shapes

def make_pair(ctx, n, k, dist="stride", seed=0):
    """(A, B): about n records each, about half of each shared."""
    from . import capi
    if dist == "stride":
        n_s, n_p = n // 2, n - n // 2
        s = ctx.alloc(n_s, k)
        ctx.generate_ex(s, n_s, seed + 11, seed + 21, 8, 3, 0)
        p = ctx.alloc(n_p, k)
        ctx.generate_ex(p, n_p, seed + 12, seed + 23, 8, 3, 1)
        _, out, _ = ctx.compare(s, p, capi.OP_UNION)
        a = out[capi.OP_UNION]
        ctx.generate_ex(s, n_s, seed + 11, seed + 22, 8, 3, 0)
        ctx.generate_ex(p, n_p, seed + 13, seed + 24, 8, 3, 2)
        _, out, _ = ctx.compare(s, p, capi.OP_UNION)
        b = out[capi.OP_UNION]
        s.free()
        p.free()
        assert a.n_words == n and b.n_words == n, (a.n_words, b.n_words, n)
        return a, b
    if dist == "genomic":
        anc = _ancestor(n + n // 50 + k, seed + 5)
        # substitutions at 1 / 72 of the positions of each copy: (1 - 1/72)^(2k) ~ 0.5 of the k-mers shared for k = 25
        a = _genome_list(ctx, _mutate(anc, 72 * 25 // k, seed + 1), k)
        b = _genome_list(ctx, _mutate(anc, 72 * 25 // k, seed + 2), k)
        return a, b
    a, b = _from_universe(ctx, dist, n + n // 2, k, 3, [(0, 1), (1, 2)], seed + 17)
    return a, b


def make_lists8(ctx, n8, k, dist="stride", n_lists=8):
    """The union bench's lists: even lists share their keys, odd lists have keys of their own."""
    if dist in ("stride", "disjoint"):
        # "disjoint" (round 5): every list has keys of its own -- what glistmaker collates (temporary lists of different
        # stretches of the input share few k-mers, reference src/glistmaker.c:787-835); "stride": even lists share theirs
        lists = []
        for j in range(n_lists):
            lst = ctx.alloc(n8, k)
            shared = j % 2 == 0 and dist == "stride"
            ctx.generate_ex(lst, n8, 7 if shared else 100 + j, 50 + j, 8, 2 * n_lists, 0 if shared else 1 + j)
            lists.append(lst)
        return lists
    if dist == "genomic":
        anc = _ancestor(n8 + n8 // 50 + k, 77)
        return [_genome_list(ctx, _mutate(anc, 40 * 25 // k, 100 + j), k) for j in range(n_lists)]
    n_odd = n_lists // 2
    member = [(0,) if j % 2 == 0 else ((j + 1) // 2,) for j in range(n_lists)]
    return _from_universe(ctx, dist, n8 * (n_odd + 1), k, n_odd + 1, member, 4242)


def make_lists_shared(ctx, n, k, dist="stride", n_lists=8):
    """The N-way intersection bench's lists: every list holds the shared key set (half of its records) and keys
    of its own; the intersection of all of them is the shared set (n // 2 records for `stride`)."""
    from . import capi
    if dist == "stride":
        n_s, n_p = n // 2, n - n // 2
        s, p = ctx.alloc(n_s, k), ctx.alloc(n_p, k)
        lists = []
        for j in range(n_lists):
            ctx.generate_ex(s, n_s, 7, 300 + j, 8, n_lists + 1, 0)
            ctx.generate_ex(p, n_p, 100 + j, 400 + j, 8, n_lists + 1, 1 + j)
            _, out, _ = ctx.compare(s, p, capi.OP_UNION)
            lists.append(out[capi.OP_UNION])
            assert lists[-1].n_words == n
        s.free()
        p.free()
        return lists
    if dist == "genomic":
        anc = _ancestor(n + n // 50 + k, 78)
        # (1 - r)^(k * n_lists) ~ 0.5 of the ancestor's k-mers survive in every copy
        r_inv = max(2, int(k * n_lists / 0.69))
        return [_genome_list(ctx, _mutate(anc, r_inv, 200 + j), k) for j in range(n_lists)]
    member = [(0, j + 1) for j in range(n_lists)]
    return _from_universe(ctx, dist, (n // 2) * (n_lists + 1), k, n_lists + 1, member, 4343)
