/*
 * gt4hip_kernels.hip -- hand-written gfx950 (MI355X, CDNA4) kernels for sorted k-mer list
 * set operations.  wave64 everywhere; no MFMA (integer streaming merge, HBM-bound).
 *
 * The hot path restates, per merged key, what the reference's single-threaded loop does
 * (compare_wordmaps, reference src/glistcompare.c:843-905; predicates :433-489):
 *
 *   K1 k_partition*   merge-path co-ranking (coarse, then fine inside the coarse bracket): tile t
 *                     starts at (a_t, b_t) with a_t + b_t = t*TILE, "A first on ties"; a matching
 *                     A/B pair is never split between tiles.
 *   K2 k_pair_merge   persistent workgroups, tiles dealt statically; per tile: coalesced 16-byte
 *                     loads of the two record ranges (one tile ahead, in registers) -> LDS as they
 *                     lie in HBM (packed 12-byte records), one record per lane: rank in the other
 *                     list by a scalar-planned LDS search, classification {A only, B only, both},
 *                     up to four output predicates, wavefront ballots + popcount prefix for output
 *                     slots; a scanner wavefront chains the tile totals into global output
 *                     offsets; compaction into LDS staging, written out one or two tiles later
 *                     with 16-byte stores.
 *   K3 k_scan_*       tile-count scan for the two-pass fallback.
 *   K0 k_generate     synthetic ascending lists written straight into HBM (bench only).
 */
#define GT4_RESOLVE_LOOKBACK 3
#include "gt4hip_device.h"

#define GT4_IPT_UNION 4 /* 6 (one staging slot written out late) measured 3 % slower than 4 with two slots */
#define GT4_IPT_INTERSECT 6
#define GT4_IPT_INTERSECT_SMALL 4 /* positions per thread of the 512-thread intersection (experiments: 6) */

namespace gt4 {

namespace {

/* ------------------------------------------------------------------ count rules */

enum : u32 { KIND_SKIP = 0, KIND_A = 1, KIND_B = 2, KIND_BOTH = 3 };

/* Wave-uniform description of one output stream: calculate_freq (reference
 * src/glistcompare.c:433-455, + RULE_MINZ :669) as a mask-and-add over {f1, f2, min, max} so that
 * the per-record code is straight-line VALU with scalar operands -- no per-record branching on the
 * rule -- and the keep-predicate of include_in_{union,intersection,complement} (:459-489) as flags. */
struct StreamCoef {
  u32 m_f1, m_f2, m_min, m_max, m_sub2; /* count = (m_f1&f1)+(m_f2&f2)+(m_min&min)+(m_max&max)-(m_sub2&f2)+konst */
  u32 konst;
  u32 minz;     /* RULE_MINZ: min := f2 when f1 == 0                                       */
  u32 check_in; /* apply the cutoff test on the INPUT counts (FILTER_REFERENCE)           */
  u32 lo;       /* keep iff count >= lo: 1 (reference "!= 0"), 0 (raw) or cutoff (result) */
  u32 cutoff;
  u32 subtract; /* -du on diff1: keep iff f1 == f2 && f1 >= cutoff, count = f1            */
};

template <int S>
__device__ __forceinline__ StreamCoef make_coef (const PairParams &p)
{
  StreamCoef c;
  const u32 rule = p.rule[S];
  const u32 all = 0xffffffffu;
  c.m_f1 = (rule == 1 || rule == 5) ? all : 0u;                   /* ADD, FIRST            */
  c.m_f2 = (rule == 1 || rule == 6) ? all : 0u;                   /* ADD, SECOND           */
  c.m_min = (rule == 3 || rule == RULE_MINZ) ? all : 0u;          /* MIN                   */
  c.m_max = (rule == 4 || rule == 2) ? all : 0u;                  /* MAX, SUBTRACT         */
  c.m_sub2 = (rule == 2) ? all : 0u;                              /* SUBTRACT = max(f1,f2) - f2 */
  c.konst = (rule == 7) ? p.count_override : 0u;                  /* NUMBER                */
  c.minz = (rule == RULE_MINZ) ? 1u : 0u;
  const bool ref = (S >= 2) || p.filter == FILTER_REFERENCE;      /* complements are reference-only */
  c.check_in = ref ? 1u : 0u;
  c.lo = ref ? 1u : (p.filter == FILTER_RESULT ? p.cutoff : 0u);
  c.cutoff = p.cutoff;
  c.subtract = (S == 2 && p.subtract) ? 1u : 0u;
  return c;
}

/* Does stream S keep this merged record, and with which count?  Branch-free.
 * S = 0 include_in_union (:459-466), 1 include_in_intersection (:468-475),
 * 2 / 3 include_in_complement (:477-489) called as (f1,f2,subtract) / (f2,f1,0). */
template <int S>
__device__ __forceinline__ bool eval_stream (u32 kind, u32 fa, u32 fb, const StreamCoef &c, u32 &freq)
{
  /* fa / fb are already 0 when the key is absent from that list (:875, :891) */
  const u32 f1 = (S == 3) ? fb : fa, f2 = (S == 3) ? fa : fb;
  bool domain, in_ok;
  if (S == 0) {
    domain = kind != KIND_SKIP;
    in_ok = f1 >= c.cutoff || f2 >= c.cutoff;
  } else if (S == 1) {
    domain = kind == KIND_BOTH;
    in_ok = f1 >= c.cutoff && f2 >= c.cutoff;
  } else {
    domain = (kind & (S == 2 ? KIND_A : KIND_B)) != 0;
    in_ok = f1 >= c.cutoff && f2 < c.cutoff;
  }
  const u32 mx = f1 > f2 ? f1 : f2;
  u32 mn = f1 < f2 ? f1 : f2;
  if (c.minz && f1 == 0) mn = f2;
  u32 f = (c.m_f1 & f1) + (c.m_f2 & f2) + (c.m_min & mn) + (c.m_max & mx) - (c.m_sub2 & f2) + c.konst;
  bool keep = domain && (in_ok || !c.check_in) && f >= c.lo;
  if (S == 2) {
    const bool keep_sub = domain && f1 == f2 && f1 >= c.cutoff;
    keep = c.subtract ? keep_sub : keep;
    f = c.subtract ? f1 : f;
  }
  freq = f;
  return keep;
}

/* The reference predicates with every stream's DEFAULT rule (union ADD, intersection MIN,
 * complements SUBTRACT; no -du) and any cutoff, written out: what eval_stream computes through its
 * coefficient form when glistcompare is run without -r (src/glistcompare.c:459-489). */
template <int S>
__device__ __forceinline__ bool eval_default (u32 kind, u32 fa, u32 fb, u32 cutoff, u32 &freq)
{
  if (S == 0) {
    freq = fa + fb;
    return kind != KIND_SKIP && (fa >= cutoff || fb >= cutoff) && freq != 0u;
  }
  if (S == 1) {
    freq = fa < fb ? fa : fb;
    return kind == KIND_BOTH && freq >= cutoff && freq != 0u; /* both >= cutoff <=> min >= cutoff */
  }
  const u32 f1 = S == 3 ? fb : fa, f2 = S == 3 ? fa : fb;
  freq = f1 - f2; /* kept only when f1 >= cutoff > f2 */
  return (kind & (S == 2 ? KIND_A : KIND_B)) != 0 && f1 >= cutoff && f2 < cutoff && freq != 0u;
}

/* ------------------------------------------------------------------ K1: partition */

/* Number of A records among the first `diag` records of merge(A, B) with A first on ties,
 * searched inside [lo, hi] (the caller's bracket must contain the answer). */
__device__ __forceinline__ u64 merge_path (const u32 *__restrict__ A, u64 nA, const u32 *__restrict__ B, u64 nB, u64 diag, u64 lo, u64 hi)
{
  const u64 lo0 = diag > nB ? diag - nB : 0, hi0 = diag < nA ? diag : nA;
  if (lo < lo0) lo = lo0;
  if (hi > hi0) hi = hi0;
  while (lo < hi) {
    const u64 mid = (lo + hi) >> 1;
    if (load_key (A, mid) <= load_key (B, diag - 1 - mid)) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

constexpr u64 PART_COARSE = 64; /* tiles per coarse partition interval */

/* level 1: the co-rank of every PART_COARSE-th tile boundary, full-range search (few threads) */
__global__ void k_partition_coarse (const u32 *__restrict__ A, u64 nA, const u32 *__restrict__ B, u64 nB, u64 num_tiles, u64 tile_records, u64 *__restrict__ coarse)
{
  const u64 c = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  const u64 n_coarse = (num_tiles + PART_COARSE - 1) / PART_COARSE;
  if (c > n_coarse) return;
  const u64 total = nA + nB;
  u64 diag = c * PART_COARSE * tile_records;
  if (diag > total) diag = total;
  coarse[c] = merge_path (A, nA, B, nB, diag, 0, nA);
}

/* level 2: every tile boundary, searched only between its two coarse neighbours (the co-rank
 * is monotone in the diagonal): ~17 dependent reads instead of ~31, and the threads of one
 * coarse interval probe the same few cache lines */
__global__ void k_partition (const u32 *__restrict__ A, u64 nA, const u32 *__restrict__ B, u64 nB,
                             u64 num_tiles, u64 tile_records, const u64 *__restrict__ coarse, u64 *__restrict__ part)
{
  const u64 t = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (t > num_tiles) return;
  const u64 total = nA + nB;
  u64 diag = t * tile_records;
  if (diag > total) diag = total;
  const u64 c = t / PART_COARSE;
  u64 a = (t % PART_COARSE) ? merge_path (A, nA, B, nB, diag, coarse[c], coarse[c + 1]) : coarse[c], b = diag - a;
  /* keys are unique inside a list, so a key present in both lists sits as an adjacent (A, B)
   * pair in the merged order; if the diagonal falls between them pull the B record back into
   * the earlier tile so that one tile owns the pair. */
  if (a > 0 && b < nB && load_key (A, a - 1) == load_key (B, b)) b += 1;
  part[2 * t] = a;
  part[2 * t + 1] = b;
}

/* ------------------------------------------------------------------ K2: tile merge by rank search */

/*
 * Persistent workgroups; each processes one merge-path tile (<= CAP - 64 records of A and B
 * together) at a time; worker w takes tiles w, w + W, ... (static dealing):
 *
 *  phase 0  the tile's two packed record ranges, fetched one tile AHEAD into registers with
 *           16-byte buffer loads, are copied to LDS as they are (A from dword 0, B from the next
 *           16-byte boundary).  The next tile's loads are issued during phases 1-2.
 *  phase 1  position space: A records at [0, na), B records from the next multiple of 64, so a
 *           64-position chunk (one wavefront pass) never mixes the lists.  One record per lane:
 *           rank r = number of records of the OTHER list with a smaller key (rank_group), match
 *           test at r, classification {A only, B only, both}, per-stream keep predicate; the keep
 *           flags of each chunk go to LDS as one wavefront ballot.  Chunks that cannot keep a
 *           record do no per-record work.
 *  phase 2  popcount-scan of the chunk ballots (single-output kernels: by every wavefront; else
 *           one wavefront per stream), tile total published for the scanner wavefront.
 *  phase 3  output slot of a kept record = (kept records before it in its own list)
 *           + (kept records of the other list before its rank) -- two ballot/prefix lookups, no
 *           sort; records are scattered into an LDS staging slot in output order.
 *
 * Write-out is DEFERRED.  Single-output kernels (OPS = union, intersection or first complement)
 * keep LAG staging slots and write tile i at the top of iteration i + LAG, when the scanner has
 * long published its offset.  The any-combination kernel (OPS = 0) stages all requested streams of
 * a tile in one area (at most 2 x tile records) and writes them out during the next tile, after its
 * ranking.  No wavefront waits on the chain unless the device is oversubscribed; then a bounded
 * wait gives up and the host reruns the call on the two-pass path.
 *
 * Keys are unique inside a list (reference precondition), so "both" pairs are found by the match
 * test alone; the B record of a pair keeps nothing (its A partner carries both counts).
 */
/* launch bound (waves per SIMD): count-only kernels of the small geometry fit 85 VGPRs and 26 KB of
 * LDS -> three workgroups per CU; everything else runs at 4 waves per SIMD */
__host__ __device__ constexpr int merge_waves_per_simd (int nt, int mode, int ops = 0, int fast = 0)
{
  /* the small geometry's folded intersection fits 80 registers and 50 KB too: three workgroups per CU */
  return (nt == 512 && (mode == MODE_COUNT || (ops == 2 && fast == 1 && GT4_IPT_INTERSECT_SMALL <= 4))) ? 6 : 4;
}

/* records per thread: an intersection does per-record work on the A half of a tile only and
 * stages at most half a tile, so its tiles are 1.5x as long (6 positions per thread, 6080 records:
 * the per-tile costs -- barriers, ring, scan, fetch set-up -- are paid two thirds as often) */
__host__ __device__ constexpr int merge_ipt (int nt, int ops_class)
{
  return (nt == 1024 && ops_class == 2) ? GT4_IPT_INTERSECT
         : ((nt == 1024 && ops_class == 1) ? GT4_IPT_UNION : ((nt == 512 && ops_class == 2) ? GT4_IPT_INTERSECT_SMALL : MERGE_VT)); /* 0 (any) and 4 (complement): MERGE_VT */
}

/* staging layout of a kernel's LDS (third parameter of RankShared) */
enum : int { STAGE_NONE = 0, STAGE_UNION = 1, STAGE_INTRSEC = 2, STAGE_ANY = 3, STAGE_UNION_LATE = 4, STAGE_COMPLEMENT = 5 };

template <int NT, int IPT, int OPS>
struct RankShared {
  static constexpr int CAP = NT * IPT;
  static constexpr int NCH = CAP / WAVE;
  /* deferred staging: an intersection keeps at most one record per pair, a union at most CAP */
  /* OPS == 3 (any combination of outputs): union + intersection + both complements of one tile are
   * at most 2 x tile records (union = tile - pairs, intersection = pairs, complements = tile - 2 pairs);
   * each stream's start is rounded up to 4 records (16-byte LDS reads in the write-out) */
  static constexpr int STAGE_DW = OPS == STAGE_INTRSEC ? ((3 * (CAP / 2 + 1) + 3) & ~3)
                                  : ((OPS == STAGE_UNION || OPS == STAGE_COMPLEMENT) ? 3 * CAP /* a complement may keep the whole tile */
                                     : (OPS == STAGE_ANY ? 3 * (2 * CAP + 16) : (OPS == STAGE_UNION_LATE ? 3 * (CAP + 16) : 4))); /* 16-byte multiples */
  /* write-out lags this many tiles behind ranking (LAG): two slots in general; an intersection on
   * 4-position tiles has room for four half-size slots; the late-written layouts use one */
  static constexpr int STAGE_SLOTS = (OPS == STAGE_INTRSEC && IPT <= 4 && NT > 512) ? 4 : ((OPS == STAGE_ANY || OPS == STAGE_UNION_LATE) ? 1 : 2);
  /* input view: the tile's packed records exactly as they lie in HBM (12-byte AoS), the A range
   * from dword 0, the B range from the next 16-byte boundary */
  alignas (16) u32 raw[3 * CAP];
  u32 stage[STAGE_SLOTS][STAGE_DW];
  u64 kmask[4][NCH + 1];  /* keep-flag ballot per 64-record chunk, per stream (+1: empty sentinel chunk) */
  u32 cpre[4][NCH + 1];   /* exclusive prefix of popcount(kmask) over chunks, per stream               */
  u64 excl[4];            /* global exclusive output offset of the tile being written out              */
  u32 tot[4];             /* records the current tile keeps, per stream                                */
  u32 tick[3];            /* [0]: role election scratch (tile ids themselves are arithmetic: static dealing)      */
  u64 rng[3][4];          /* their record ranges {a0, b0, a1, b1} (part[] entries), fetched ahead        */
  u32 tile_id[3];         /* and their tile numbers (0xffffffff: none)                                    */
  u64 trash[WAVE];        /* where the lanes that have nothing to store put it (see store_lane0)         */
};

/* Lane 0 stores v at dst; the other lanes store into a scratch row.  `if (lane == 0)` costs three
 * scalar instructions of exec bookkeeping on the CU's one scalar unit, every lane storing to the SAME
 * address is serialised by the LDS; distinct scratch addresses cost one select.  (Measured: the
 * intersection kernel 11.37 -> 11.21 ms against `if (lane == 0)`, 12.6 with the all-lane store; the union
 * and the any-combination kernel are scalar-bound and marginally better off with the all-lane store.) */
__device__ __forceinline__ void store_lane0 (u64 *dst, u64 *trash, u64 v, int lane)
{
  u64 *const w = lane == 0 ? dst : trash + lane;
  *w = v;
}

/* number of kept records among the concatenated tile positions [0, z) */
__device__ __forceinline__ u32 kept_before (const u64 *km, const u32 *cp, u32 z)
{
  const u32 c = z >> 6;
  return cp[c] + (u32) __popcll (km[c] & ((1ull << (z & 63u)) - 1ull));
}

template <int S, int NT, int IPT, int OPS, int FAST, class Shared>
__device__ __forceinline__ void scatter_stream (Shared &sh, u32 *dst32, const PairParams &p, u32 nbs, int lane, int wid,
                                                const u64 (&key)[IPT], const u32 (&fa)[IPT], const u32 (&fb)[IPT], const u32 (&meta)[IPT])
{
  constexpr int NW = NT / WAVE;
  const StreamCoef c = make_coef<S> (p);
  /* the intersection and the first complement keep A records only, in every kernel: a pair is evaluated at its A record
   * (its B partner is KIND_SKIP) and a key that only B holds belongs to neither -- their slots need no look at the other
   * list's kept records (round 6: the any-combination kernel spent two prefix lookups per record on a term that is 0) */
  constexpr bool A_ONLY = S == 1 || S == 2;
  const u32 pna = A_ONLY ? 0u : kept_before (sh.kmask[S], sh.cpre[S], nbs); /* nbs: tile position of the first B record */
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const u32 chunk = (u32) k * NW + (u32) wid;
    /* the chunk's ballot is the same in every lane: as a scalar it IS the lane mask of the kept
     * records (no per-lane bit test) and the count of kept lanes below is one mbcnt pair */
    const u64 mv = sh.kmask[S][chunk];
    const u32 m_lo = __builtin_amdgcn_readfirstlane ((u32) mv), m_hi = __builtin_amdgcn_readfirstlane ((u32) (mv >> 32));
    if (__builtin_amdgcn_inverse_ballot_w64 (((u64) m_hi << 32) | m_lo)) {
      const u32 own = sh.cpre[S][chunk] + __builtin_amdgcn_mbcnt_hi (m_hi, __builtin_amdgcn_mbcnt_lo (m_lo, 0u));
      u32 slot = own;
      if (!A_ONLY) { /* (A-only streams: nothing of the other list comes before) */
        const u32 r = meta[k] & 0xffffu;
        const u32 z = ((meta[k] >> 18) & 1u) ? nbs + r : r;
        slot += kept_before (sh.kmask[S], sh.cpre[S], z) - pna;
      }
      u32 f;
      if (OPS != 0) f = fa[k]; /* single-output kernels carry the stream's count itself */
      else if (FAST) eval_default<S> ((meta[k] >> 16) & 3u, fa[k], fb[k], p.cutoff, f);
      else eval_stream<S> ((meta[k] >> 16) & 3u, fa[k], fb[k], c, f);
      dst32[3 * slot] = (u32) key[k];
      dst32[3 * slot + 1] = (u32) (key[k] >> 32);
      dst32[3 * slot + 2] = f;
    }
  }
}

/* Ranks of G keys, each in its own sorted run of packed records in LDS (lower bound: the number
 * of records with a smaller key), as byte offsets 12 * rank into the run.  The runs' starts and
 * lengths are wave-uniform, which makes the whole search plan scalar: with P = 2^bitlen(n) > n,
 * the first probe is record n - P/2 and leaves a window of exactly P/2 - 1 records on either side,
 * then every probe is the middle of a window of 2h - 1 records (h = P/4 ... 1).  bitlen(n) probes
 * in all, every one of them inside the run -- no bounds test -- and for h <= 64 the probe's offset
 * from the running position fits the LDS instruction's offset field: compare, select, add per
 * step.  Steps a short run does not need add 0 (their probes read this workgroup's LDS beyond
 * the run, and the value is ignored).  The G searches are interleaved: G LDS reads in flight per lane. */
template <int CAP, int G>
__device__ __forceinline__ void rank_group (const u32 *lds32, const u32 (&sbase)[G], const u32 (&sn)[G], const u64 (&ky)[G], u32 (&lo)[G], bool narrow, u32 base_lo)
{
  auto key_at = [&] (u32 byte_off) -> u64 {
    const u32 *const q = reinterpret_cast<const u32 *> (reinterpret_cast<const char *> (lds32) + byte_off);
    return (u64) q[0] | ((u64) q[1] << 32);
  };
  constexpr u32 HTOP = (CAP & (CAP - 1)) ? (1u << (31 - __builtin_clz ((unsigned) CAP))) / 2 : CAP / 4; /* P/4 of the longest run */
  u32 at[G];
  /* one probe step of all G searches with the compile-time stride h */
  auto step = [&] (u32 h) {
    u64 pv[G];
#pragma unroll
    for (int u = 0; u < G; u++) pv[u] = key_at (at[u] + 12u * (h - 1u));
#pragma unroll
    for (int u = 0; u < G; u++) {
      u32 cand = at[u] + 12u * h;     /* independent of the probe: issued under its latency */
      asm volatile ("" : "+v"(cand)); /* keep add + select */
      at[u] = pv[u] < ky[u] ? cand : at[u];
    }
  };
  bool same = true; /* wave-uniform: every chunk of the group is ranked in the same run (the usual group) */
#pragma unroll
  for (int u = 1; u < G; u++) same &= sn[u] == sn[0] && sbase[u] == sbase[0];
  if (same) {
    /* ONE plan for the group: the scalar unit is shared by the CU's sixteen wavefronts and is this
     * kernel's scarcest resource, so the per-search selects of the general form below
     * are worth avoiding: a step is either skipped by one scalar branch or runs on constants */
    const u32 n = sn[0], base = 4u * sbase[0];
    const u32 half = n ? 1u << (31 - __builtin_clz (n)) : 0u; /* P/2 = the largest power of two <= n */
    const u32 q4 = half >> 1, i0 = n - half, inc = n ? 12u * (i0 + 1u) : 0u;
    const u64 pv0 = key_at (base + 12u * i0); /* the same record for every lane and search */
#pragma unroll
    for (int u = 0; u < G; u++) at[u] = base + (pv0 < ky[u] ? inc : 0u);
    if (q4 >= HTOP / 2 && narrow) {
      /* the usual tile whose keys all lie within 2^32 of its smallest one (wave-uniform, tested once per
       * tile by the caller): keys compare as (low dword - base_lo) mod 2^32, so a probe reads 4 bytes
       * instead of 8 -- the ranking phase is bound by LDS cycles (50 extra LDS reads per wavefront and
       * tile cost this kernel 30 %), and the probes are most of them */
      u32 k32[G];
#pragma unroll
      for (int u = 0; u < G; u++) k32[u] = (u32) ky[u] - base_lo;
      auto key32_at = [&] (u32 byte_off) -> u32 {
        return *reinterpret_cast<const u32 *> (reinterpret_cast<const char *> (lds32) + byte_off) - base_lo;
      };
      auto step32 = [&] (u32 h) {
        u32 pv[G];
#pragma unroll
        for (int u = 0; u < G; u++) pv[u] = key32_at (at[u] + 12u * (h - 1u));
#pragma unroll
        for (int u = 0; u < G; u++) {
          u32 cand = at[u] + 12u * h;
          asm volatile ("" : "+v"(cand));
          at[u] = pv[u] < k32[u] ? cand : at[u];
        }
      };
      if (q4 >= HTOP) step32 (HTOP);
#pragma unroll
      for (u32 h = HTOP / 2; h >= 1; h >>= 1) step32 (h);
    } else if (q4 >= HTOP / 2) {
      /* the usual tile (two lists of similar density: every run holds at least a quarter of the tile's
       * capacity): one test for the top step, the others run unconditionally */
      if (q4 >= HTOP) step (HTOP);
#pragma unroll
      for (u32 h = HTOP / 2; h >= 1; h >>= 1) step (h);
    } else {
#pragma unroll
      for (u32 h = HTOP / 4; h >= 1; h >>= 1) {
        if (h > q4) continue;
        step (h);
      }
    }
  } else {
    /* general form: every search has its own plan; steps a short run does not need add 0 */
    u32 q4[G];
#pragma unroll
    for (int u = 0; u < G; u++) {
      const u32 n = sn[u];
      const u32 half = n ? 1u << (31 - __builtin_clz (n)) : 0u;
      q4[u] = half >> 1;
      const u32 i0 = n - half;
      const u32 inc = n ? 12u * (i0 + 1u) : 0u;
      const u64 pv = key_at (4u * sbase[u] + 12u * i0);
      at[u] = 4u * sbase[u] + (pv < ky[u] ? inc : 0u);
    }
#pragma unroll
    for (u32 h = HTOP; h >= 1; h >>= 1) {
      bool any = false; /* wave-uniform */
#pragma unroll
      for (int u = 0; u < G; u++) any |= h <= q4[u];
      if (!any) continue;
      u64 pv[G];
#pragma unroll
      for (int u = 0; u < G; u++) pv[u] = key_at (at[u] + 12u * (h - 1u));
#pragma unroll
      for (int u = 0; u < G; u++) {
        const u32 hs = h <= q4[u] ? 12u * h : 0u; /* scalar */
        u32 cand = at[u] + hs;
        asm volatile ("" : "+v"(cand));
        at[u] = pv[u] < ky[u] ? cand : at[u];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < G; u++) lo[u] = at[u] - 4u * sbase[u];
}

struct TileRange {
  u64 a0, b0; /* BYTE offsets of the tile's first records in the two lists */
  u32 na, nb;
};


/* OPS != 0 fixes the set of output streams at compile time (the common single-output calls get a
 * kernel without the other streams' code and registers); OPS == 0 takes it from p.ops. */
/* FAST != 0 (single-output kernels, chosen by the launcher from the call's parameters): the count
 * rule and keep test of the commonest calls as two or three instructions instead of the general
 * coefficient form -- 1: the reference predicate with the operation's default rule (union: ADD,
 * intersection: MIN, first complement: SUBTRACT without -du; any-combination kernel: every
 * requested stream on its default rule), any cutoff; 2 / 3: ADD keeping every key / every sum
 * >= cutoff (intermediate and final N-way union levels), and for the intersection the running minimum
 * of intersect_multi's chain keeping every shared key / every count >= cutoff. */
/* OPSET != 0 (any-combination kernel only): the set of output streams fixed at compile time -- the
 * other streams' code, registers and scalar branches disappear (glistcompare -u -d, BASELINE config 2) */
template <int NT, int IPT, int MODE, int OPS, int FAST = 0, int OPSET = 0>
__global__ __launch_bounds__ (NT, merge_waves_per_simd (NT, MODE, OPS, FAST)) void
k_pair_merge (const u32 *__restrict__ A, u64 nA, const u32 *__restrict__ B, u64 nB, u64 *part, u64 num_tiles,
              PairParams p, PairOutputs outs, u64 *desc, PairControl *ctl)
{
  constexpr int CAP = NT * IPT;
  constexpr int NW = NT / WAVE;
  constexpr int NCH = CAP / WAVE;
  constexpr int NLOAD4 = (3 * IPT + 3) / 4;      /* 16-byte chunks each thread fetches per tile */
  constexpr bool LATE1 = OPS == 1 && IPT > 4;    /* union on long tiles: one staging slot, written out late (as the any-combination kernel) */
  constexpr bool DEFER = OPS != 0 && !LATE1 && MODE != MODE_COUNT;
  constexpr int S0 = OPS == 2 ? 1 : (OPS == 4 ? 2 : 0); /* the stream of a single-output kernel (OPS = 1 union, 2 intersection, 4 first complement) */
  static_assert (OPS == 0 || OPS == 1 || OPS == 2 || OPS == 4, "specialised kernels exist for one of these outputs alone");
  /* any-combination kernel: all requested streams of a tile are staged in one LDS area and written
   * out during the NEXT tile (after its ranking), when their global offsets have long been published */
  constexpr bool GDEFER = (OPS == 0 || LATE1) && MODE != MODE_COUNT;
  constexpr int G = IPT % 3 == 0 ? 3 : 2; /* chunks searched together (four at a time is no faster for the union and spills the 85-register count kernels) */
  /* staggered fetch: the first F_TOP of the NLOAD4 + 1 parts are issued at the top of the iteration,
   * F_MID between the search groups, the rest behind the ranking.  Measured per kernel class (2 x 2e9
   * records): the union is fastest with one part at the top and one in the middle, the intersection
   * (after the scalar diet) with two at the top and none in the middle (11.6 -> 11.4 ms); the
   * first complement with all whole parts at the top (16.1 -> 14.3 ms); the any-combination kernel,
   * which also resolves its offsets behind the ranking, with everything at the top (29.8 -> 28.7 ms). */
  constexpr int F_TOP = GDEFER ? NLOAD4 + 1 : (OPS == 4 ? NLOAD4 : (OPS == 2 ? 2 : 1));
  constexpr int F_MID = (GDEFER || OPS == 4 || OPS == 2) ? 0 : 1;
  constexpr bool STAGGER = NT >= 1024;           /* spread the fetch over the iteration (measured: helps 16-wave workgroups only) */
  static_assert (NW >= 4, "one wavefront per output stream in phase 2");
  static_assert (NCH <= 2 * WAVE, "chunk scan is a single wavefront pass");
  /* count-only kernels stage nothing: no staging slots in their LDS */
  typedef RankShared<NT, IPT, (MODE == MODE_COUNT ? STAGE_NONE : (OPS == 0 ? STAGE_ANY : (OPS == 4 ? STAGE_COMPLEMENT : (OPS == 2 ? STAGE_INTRSEC : (IPT > 4 ? STAGE_UNION_LATE : STAGE_UNION)))))> Shared;
  __shared__ Shared sh;
  u32 *const lds32 = sh.raw;

  /* The general any-combination kernels sit at their register bound (128 at sixteen wavefronts per CU, 85 for the
   * count-only geometry's three workgroups): there the thread number is made opaque once per tile, so that
   * addresses and masks derived from it are recomputed where they are used (a few VALU each) instead of living in
   * registers across the whole loop, hoisted by the compiler -- which had two to six of them in scratch memory.
   * tools/kernel_resources.py / tests/test_kernel_resources.py: no instantiation may spill a vector register. */
  constexpr bool OPAQUE_TID = OPS == 0 && !(FAST == 1 && (OPSET == 3 || OPSET == 5) && MODE != MODE_COUNT);
  int tid = threadIdx.x, lane = tid & (WAVE - 1); /* (not const: OPAQUE_TID) */
  const int wid = __builtin_amdgcn_readfirstlane (tid / WAVE); /* wave-uniform: scalar branches on it */
  static_assert (OPSET == 0 || OPS == 0, "a compile-time stream set belongs to the any-combination kernel");
  const u32 ops = OPS ? (u32) OPS : (OPSET ? (u32) OPSET : p.ops);
  /* every "both" pair is evaluated at its A record, so A records always matter; B records only
   * where a B-only key can be kept (union, diff2) */
  const bool need_b = (ops & 9u) != 0;

  /* single pass: agg u32[4][rows * 64] then carry u64[4][rows + 1] inside desc (zeroed by the host) */
  const u64 n_rows = (num_tiles + WAVE - 1) / WAVE;
  const u32 spin_limit = p.spin_limit ? p.spin_limit : SPIN_LIMIT;
  u32 *const agg = reinterpret_cast<u32 *> (desc);
  u64 *const carry = desc + 2 * n_rows * WAVE; /* 4 * rows * 64 u32 */

  u32 role = 0;
  if (MODE == MODE_LOOKBACK) {
    /* role election: the first workgroup to arrive is running by definition, it becomes the scanner */
    if (tid == 0) sh.tick[0] = atomicAdd (&ctl->role, 1u);
    __syncthreads ();
    role = sh.tick[0];
    __syncthreads ();
    if (role == 0) {
      /* the workgroup's wavefronts are dealt to the requested streams round-robin: the first one of a
       * stream chains its carries, the others sum its rows (scanner_part) */
      const u32 n_str = (u32) __builtin_popcount (ops);
      const u32 k = (u32) wid % n_str, sub = (u32) wid / n_str, n_sub = ((u32) NW - k + n_str - 1) / n_str;
      u32 s = 0;
      for (u32 seen = 0, b = 0; b < 4; b++)
        if ((ops >> b) & 1u) {
          if (seen == k) s = b;
          seen++;
        }
      u64 *const rowsum = carry + 4 * (n_rows + 1);
      if (sub < 8u && (p.scan_group || sub == 0))
        scanner_part (agg + (u64) s * n_rows * WAVE, rowsum + (u64) s * n_rows, carry + (u64) s * (n_rows + 1), num_tiles, ctl, lane, spin_limit, sub,
                      p.scan_group ? (n_sub > 8u ? 8u : n_sub) : 1u);
      return;
    }
  }

  u64 acc_sum0 = 0, acc_sum1 = 0, acc_sum2 = 0, acc_sum3 = 0; /* per-thread sums of kept counts */
  u64 blk_cnt = 0;                                              /* lane 0 of wave s: records kept in stream s */

  /* Two ways of dealing tiles, chosen by the host per launch (p.dynamic).
   * Round-robin: worker w processes tiles w, w + W, w + 2W, ...  With the scanner this needs every
   * worker resident, which the host guarantees by sizing the grid from the kernel's occupancy;
   * worker ids come from arrival order, and every spin is bounded, so a non-resident worker shows up
   * as an error flag (the host then reruns the call on the two-pass path), never as a hang.
   * By ticket: one returning atomic per tile on a shared counter.  It saturates near 88 atomics per
   * microsecond (MI355X_MICROARCH.md, row dequeue), which caps the small geometry's 200+ tiles per
   * microsecond but not the ~40 of the kernels that write records; there arrival order means that
   * no worker ever waits for a tile of a slower one that it could have taken itself.  (Counters
   * sharded by worker class drift apart; claiming tiles in groups puts a group's last tile behind
   * the next group's write-out.)
   * Either way a three-deep ring keeps every dependent global round trip off the critical path:
   * while tile i is processed, the records of tile i+1 are in flight, the range of tile i+2 is
   * being read, and (by ticket) the number of tile i+3 is being drawn. */
  const u32 n_workers = MODE == MODE_LOOKBACK ? gridDim.x - 1 : gridDim.x;
  const u32 wk = MODE == MODE_LOOKBACK ? role - 1 : blockIdx.x;
  const u32 ntl = (u32) num_tiles; /* the host refuses calls with 2^32 - 1 tiles or more */
  /* static dealing: the tile of iteration j is known without asking anybody (0xffffffff: none) */
  auto tile_at = [&] (int j) -> u32 {
    const u64 t = (u64) wk + (u64) j * n_workers;
    return t < num_tiles ? (u32) t : 0xffffffffu;
  };
  /* dynamic dealing (p.dynamic): tiles are handed out by one returning atomic per tile, taken three
   * iterations ahead of the tile's processing (its round trip is never waited for); tile order then
   * follows arrival order, so a worker that is slow -- for a moment or for the whole launch -- takes
   * fewer tiles instead of making everybody behind it in the chained scan wait */
  auto deal = [&] (int j) -> u32 {
    if (MODE != MODE_LOOKBACK || !p.dynamic) return tile_at (j); /* the two-pass path's kernels stay round-robin */
    const u32 t = atomicAdd (&ctl->ticket, 1u);
    return t < ntl ? t : 0xffffffffu;
  };
  u32 tk_next = 0xffffffffu; /* thread 0: the tile of iteration it + 2 */
  if (tid == 0) {
    u32 t3[3];
    for (int q = 0; q < 3; q++) t3[q] = deal (q);
    for (int q = 0; q < 2; q++) {
      sh.tile_id[q] = t3[q];
      if (t3[q] < ntl) {
        {
          const u64 e0 = part[2 * (u64) t3[q]], e1 = part[2 * (u64) t3[q] + 1], e2 = part[2 * (u64) t3[q] + 2], e3 = part[2 * (u64) t3[q] + 3];
          sh.rng[q][0] = 12 * e0; /* byte offsets and record counts, ready for the descriptors: */
          sh.rng[q][1] = 12 * e1; /* computed once here, not by sixteen wavefronts' scalar code */
          sh.rng[q][2] = e2 - e0;
          sh.rng[q][3] = e3 - e1;
        }
      }
    }
    tk_next = t3[2];
  }
  __syncthreads ();
  u32 cur = uniform32 (sh.tile_id[0]);
  TileRange tr = { 0, 0, 0, 0 };
  if (cur < ntl) {
    tr.a0 = uniform64 (sh.rng[0][0]);
    tr.b0 = uniform64 (sh.rng[0][1]);
    tr.na = uniform32 ((u32) sh.rng[0][2]);
    tr.nb = uniform32 ((u32) sh.rng[0][3]);
  }
  u32x4 pre[NLOAD4];
  u32x4 pre_x = { 0, 0, 0, 0 }; /* B half of the one wave-instruction per tile that straddles the two ranges */

  /* 16-byte chunk q of the tile: chunks [0, cA) cover the A range (3*na dwords), the rest the B
   * range.  The packed 12-byte records are only 4-byte aligned, which a buffer_load_dwordx4 takes
   * as is; one range-checked descriptor per range makes every byte past the range read as 0
   * without touching memory, so the last, partial chunk needs no special case.  A wavefront is
   * all-A, all-B, or (once per tile) straddles the two ranges; the choice is wave-uniform. */
  /* fetch_part (t, j): the j-th of the NLOAD4 wave-instructions (j == NLOAD4: the straddling
   * half); fetch (t): all of them.  In the main loop the parts are issued at different points of
   * the iteration: sixteen wavefronts issuing everything right behind the same barrier only queue
   * up in front of the CU's one address unit. */
  auto fetch_part = [&] (const TileRange &t, int j) {
    const u32 da = 3 * t.na, db = 3 * t.nb, cA = (da + 3) >> 2;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc ((void *) (reinterpret_cast<const char *> (A) + t.a0), 0, (int) (4 * da), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc ((void *) (reinterpret_cast<const char *> (B) + t.b0), 0, (int) (4 * db), 0x00020000);
#pragma unroll
    for (int jj = 0; jj < NLOAD4; jj++) {
      if (jj != j) continue;
      const u32 q0 = (u32) jj * NT + (u32) wid * WAVE; /* first chunk of this wavefront */
      const u32 q = q0 + (u32) lane;
      /* all-A and straddling wavefronts read the A range (lanes past it get zeros), all-B ones the B range */
      if (q0 < cA) pre[jj] = __builtin_amdgcn_raw_buffer_load_b128 (ra, 16 * q, 0, GT4_LOAD_AUX);
      else pre[jj] = __builtin_amdgcn_raw_buffer_load_b128 (rb, 16 * (q - cA), 0, GT4_LOAD_AUX);
    }
    if (j == NLOAD4) {
      /* the one wave-instruction per tile that straddles the two ranges also needs its B half; the
       * two halves are OR-ed when they are consumed (phase 0): combining them here, or loading
       * the B half from inside the loop above (several static writers of one register), would
       * make this wavefront wait for its loads on the spot -- a full memory round trip per tile */
      const u32 qs = cA & ~(u32) (WAVE - 1); /* first chunk of the straddling wave-instruction */
      /* lanes that hold A chunks must read nothing: give them an offset that is out of range
       * without wrapping (a "negative" 32-bit offset may wrap inside the range check) */
      const u32 qx = qs + (u32) lane;
      if ((cA & (WAVE - 1)) && (qs / WAVE) % NW == (u32) wid) pre_x = __builtin_amdgcn_raw_buffer_load_b128 (rb, qx >= cA ? 16 * (qx - cA) : 0x7ffffff0u, 0, 0);
    }
  };
  auto fetch = [&] (const TileRange &t) {
#pragma unroll
    for (int j = 0; j <= NLOAD4; j++) fetch_part (t, j);
  };

  if (cur < ntl) fetch (tr);
PROF (
  u64 ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  u64 t_last;
  asm volatile ("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last) :: "memory");
)
  /* DEFER: the LAG tiles whose output is staged in LDS but not yet written: their numbers (past[])
   * and record counts are remembered: a queue shifted once per iteration with static indices only (entry 0 =
   * oldest = staged LAG iterations ago, in slot it % LAG) -- indexing it by it % LAG instead turns
   * the array into scratch memory, which cost a second copy of the output in HBM traffic. */
  constexpr int LAG = Shared::STAGE_SLOTS;
  u32 pend_tot[LAG];
#pragma unroll
  for (int q = 0; q < LAG; q++) pend_tot[q] = 0;
  u32 past[LAG]; /* the tiles of the last LAG iterations, oldest first (wave-uniform) */
#pragma unroll
  for (int q = 0; q < LAG; q++) past[q] = 0;
  u32 g_tot[4] = { 0, 0, 0, 0 }, g_off[4] = { 0, 0, 0, 0 }; /* GDEFER: the staged tile's records per stream and their staging offsets (records) */
  int it = 0;
  /* DEFER: wavefront 4's look at the chain words of the tile the NEXT iteration writes out (its row's
   * counts and row carry; two-pass path: its offset), carried over the back edge */
  u32 dagg = 0;
  u64 dcarry = 0;
  int r_nxt = 1, r_nn = 2; /* ring slots of the next two tiles */
  /* count-only kernels: the wavefronts that share a SIMD with the housekeeping wavefront win the CU's
   * arbitration -- measured 3 - 5 % on every count-only call (9.34 -> 8.91 ms at 2 x 2e9), nothing on
   * the kernels that write records */
  if (MODE == MODE_COUNT && (wid == 0 || wid == 4)) __builtin_amdgcn_s_setprio (1);

  while (cur < ntl) {
    if (OPAQUE_TID) {
      asm volatile ("" : "+v"(tid));
      lane = tid & (WAVE - 1);
    }
    /* position space of the tile: A records at [0, na), B records from the next multiple of 64 on, so
     * that every 64-position chunk (one wavefront pass) holds records of one list only */
    const u32 na = tr.na, nb = tr.nb, nbs = (na + (u32) WAVE - 1u) & ~((u32) WAVE - 1u), npos = nbs + nb;
    if (npos > (u32) CAP) {
      if (tid == 0) atomicOr (&ctl->error, 2u);
      break;
    }
    /* ---- phase 0: registers -> LDS, a raw copy: chunk q of the tile lands at LDS byte 16 q, so the
     * A records sit at dword 3 i and the B records at dword OB + 3 j with OB = 4 cA.  Keys are
     * then read as two dwords (a 12-byte stride is conflict-free across lanes); no AoS -> SoA pass. */
    const u32 OB = 4 * ((3 * na + 3) >> 2);
    {
      const u32 cA = OB >> 2, cB = (3 * nb + 3) >> 2;
#pragma unroll
      for (int j = 0; j < NLOAD4; j++) {
        const u32 q = (u32) j * NT + (u32) tid;
        /* chunks past the tile's last one hold zeros (range-checked loads): writing them is harmless
         * wherever the part still lies inside the input view -- no per-part test, no exec juggling */
        if ((j + 1) * NT * 4 <= 3 * CAP || q < cA + cB) *reinterpret_cast<u32x4 *> (lds32 + 4 * q) = pre[j];
      }
      /* the one wave-instruction per tile that straddles the two ranges: its B half arrived in
       * pre_x (fetch_part) and goes over the zeros the A descriptor returned for those lanes -- one
       * uniform test per tile instead of a select in every part */
      const u32 qs = cA & ~(u32) (WAVE - 1);
      if ((cA & (WAVE - 1)) && (qs / WAVE) % NW == (u32) wid) {
        const u32 q = qs + (u32) lane;
        if (q >= cA && q < cA + cB) *reinterpret_cast<u32x4 *> (lds32 + 4 * q) = pre_x;
      }
    }
    if (DEFER && it >= LAG && wid == 4) {
      /* global offset of the tile written out below, from the words asked for during the previous
       * iteration: they came back with the prefetched records this wavefront has just waited for (the
       * memory counter retires in order -- consuming them any earlier would have made this wavefront
       * wait for its share of the prefetch in the middle of the iteration, and everybody else for it
       * at the next barrier).  The previous write-out read sh.excl a whole iteration ago. */
      u64 x = dcarry;
      if (MODE == MODE_LOOKBACK)
        x = resolve_offset (agg + (u64) S0 * n_rows * WAVE, carry + (u64) S0 * (n_rows + 1), past[0], lane, dagg, dcarry, ctl, spin_limit);
      if (lane == 0) sh.excl[S0] = 12 * x; /* bytes */
    }
    PHASE_STAMP (0); /* phase 0: wait for the prefetched records, LDS writes */
    __syncthreads (); /* B0 */
    PHASE_STAMP (1); /* barrier B0 */
    const int s_nxt = r_nxt, s_nn = r_nn; /* (it + 1) % 3, (it + 2) % 3 without the division */
    const u32 nxt = uniform32 (sh.tile_id[s_nxt]);
    TileRange tn = { 0, 0, 0, 0 };
    if (nxt < ntl) {
      tn.a0 = uniform64 (sh.rng[s_nxt][0]);
      tn.b0 = uniform64 (sh.rng[s_nxt][1]);
      tn.na = uniform32 ((u32) sh.rng[s_nxt][2]);
      tn.nb = uniform32 ((u32) sh.rng[s_nxt][3]);
    }
    /* housekeeping by thread 0, results consumed at the end of this iteration */
    u64 hk_rng[4] = { 0, 0, 0, 0 };
    bool hk_have_rng = false;
    u32 hk_tile = 0xffffffffu;
    if (tid == 0) {
      hk_tile = tk_next;
      tk_next = deal (it + 3);
      const u32 tnn = hk_tile;
      if (tnn < ntl) {
        hk_have_rng = true;
#pragma unroll
        /* relaxed atomic loads: plain vector loads the wave does not wait for here (a scalar load
         * of this uniform address would be waited for on the spot) */
        for (int i = 0; i < 4; i++) hk_rng[i] = __hip_atomic_load (&part[2 * (u64) tnn + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    /* DEFER: write out the tile staged LAG iterations ago (its offset was resolved in phase 0,
     * just above) BEFORE the next fetch is issued: the memory counter retires in
     * order, so the wait for the fetched records at the next phase 0 then only ever waits on
     * stores that are a whole iteration old */
    const u32 w_tot = pend_tot[0];                       /* written out now */
    const bool w_have = it >= LAG;
    const bool n_have = LAG > 1 && it >= LAG - 1;        /* written out next iteration */
    const u32 n_tile = past[LAG > 1 ? 1 : 0];
    if (DEFER && w_have) {
      constexpr int WK = (Shared::STAGE_DW / 4 + NT - 1) / NT;
      write_out_fixed<NT, WK> (outs.rec[S0], uniform64 (sh.excl[S0]), w_tot, sh.stage[it % LAG], tid);
    }
    u32 xagg = 0;
    u64 xcarry = 0;
    if (GDEFER && MODE == MODE_LOOKBACK && it >= 1 && wid >= 4 && wid < 8 && ((ops >> (wid - 4)) & 1u)) {
      /* wavefront 4 + s asks for the words stream s of the previous tile needs (published during
       * the previous iteration) and resolves them behind its own ranking, before the late fetch parts and B1 */
      const int s = wid - 4;
      const u32 pt = past[LAG - 1];
      const u64 prow = pt / WAVE;
      if ((u32) lane < pt % WAVE) xagg = peek_u32 (&agg[(u64) s * n_rows * WAVE + prow * WAVE + lane]);
      xcarry = peek_u64 (&carry[(u64) s * (n_rows + 1) + prow]);
    }
    /* in flight until the next iteration's phase 0; the large geometry staggers the parts */
    if (nxt < ntl) {
      if (STAGGER) {
#pragma unroll
        for (int j = 0; j < F_TOP; j++) fetch_part (tn, j);
      } else fetch (tn);
    }

    /* do all keys of the tile lie within 2^32 of its smallest one?  Both runs ascend: four reads of
     * uniform addresses (broadcast), computed on the vector side (the same in every lane) so that
     * only the final test reaches the scalar unit */
    bool narrow = false;
    u32 base_lo = 0;
    if (OPS == 2 && MODE != MODE_COUNT && na && nb) { /* (the intersection: 11.30 -> 11.11 ms; the scalar-bound kernels lose 2-3 % to the test itself) */
      const u32 la = 3 * (na - 1), lb = OB + 3 * (nb - 1);
      const u64 a_min = (u64) lds32[0] | ((u64) lds32[1] << 32), a_max = (u64) lds32[la] | ((u64) lds32[la + 1] << 32);
      const u64 b_min = (u64) lds32[OB] | ((u64) lds32[OB + 1] << 32), b_max = (u64) lds32[lb] | ((u64) lds32[lb + 1] << 32);
      const u64 t_min = a_min < b_min ? a_min : b_min, t_max = a_max > b_max ? a_max : b_max;
      narrow = __builtin_amdgcn_ballot_w64 (((t_max - t_min) >> 32) == 0) != 0;
      base_lo = (u32) t_min;
    }
    PHASE_STAMP (2); /* ring read, housekeeping issue, fetch issue */
    /* ---- phase 1: rank, classify, predicates.  Chunks are handled two at a time so that every
     * step of the search has two independent LDS reads in flight per lane. */
    u64 key[IPT];
    u32 fa[IPT], fb[IPT], meta[IPT]; /* meta: rank | kind << 16 | is_a << 18 */
    {
      const StreamCoef c0 = make_coef<0> (p), c1 = make_coef<1> (p), c2 = make_coef<2> (p), c3 = make_coef<3> (p);
      static_assert (IPT % G == 0, "chunks are searched in groups");
#pragma unroll
      for (int kk = 0; kk < IPT; kk += G) {
        if (STAGGER && kk == G && nxt < ntl) {
#pragma unroll
          for (int j = F_TOP; j < F_TOP + F_MID; j++) fetch_part (tn, j);
        } /* staggered fetch: see fetch_part */
        bool live[G], valid[G], any_live = false;
        u32 is_a[G], own[G], lim[G], lo[G]; /* is_a: wave-uniform (chunks never mix the lists); lim, lo: bytes (12 per record) */
        u32 sbase[G], sn[G];                /* wave-uniform: dword base and length of the run this chunk is ranked in */
        u32 plim[G];                        /* wave-uniform: positions below it hold a record this call looks at */
        u64 ky[G];
#pragma unroll
        for (int u = 0; u < G; u++) {
          const u32 cbeg = ((u32) (kk + u) * NW + (u32) wid) * WAVE; /* wave-uniform */
          is_a[u] = cbeg < nbs ? 1u : 0u;
          plim[u] = is_a[u] ? na : (need_b ? npos : 0u);
          live[u] = cbeg < plim[u];
          any_live |= live[u];
        }
        if (!any_live) {
          /* nothing in these chunks can be kept (padding, or B records of a call that keeps none
           * of them on their own: pairs are found from the A side): empty keep masks, no other work */
#pragma unroll
          for (int u = 0; u < G; u++) {
            const int k = kk + u;
            key[k] = 0;
            fa[k] = fb[k] = 0;
            meta[k] = KIND_SKIP << 16;
            {
              const u32 chunk = (u32) k * NW + (u32) wid;
#pragma unroll
              for (int s = 0; s < 4; s++)
                if ((ops >> s) & 1u) store_lane0 (&sh.kmask[s][chunk], sh.trash, 0ull, lane);
            }
          }
          continue;
        }
#pragma unroll
        for (int u = 0; u < G; u++) {
          /* one per-lane compare against a scalar limit; everything uniform stays a scalar select
           * (mixing uniform and per-lane conditions costs the shared scalar unit a mask operation each) */
          const u32 e = (u32) (kk + u) * NT + (u32) tid;
          u32 at;
          if (OPS == 2) { /* (the intersection alone measures 4 % FASTER with the condition spelled out) */
            valid[u] = live[u] && (is_a[u] ? e < na : e < npos);
            at = valid[u] ? (is_a[u] ? 3 * e : OB + 3 * (e - nbs)) : 0u;
          } else {
            valid[u] = e < plim[u]; /* (a chunk that is not live has no position below its limit) */
            const u32 off = is_a[u] ? 0u : OB - 3 * nbs;
            at = valid[u] ? 3 * e + off : 0u;
          }
          ky[u] = (u64) lds32[at] | ((u64) lds32[at + 1] << 32);
          own[u] = lds32[at + 2];
          sbase[u] = is_a[u] ? OB : 0u;
          sn[u] = is_a[u] ? nb : na;
          lim[u] = valid[u] ? 12 * sn[u] : 0u;
          lo[u] = 0;
        }
        rank_group<CAP, G> (lds32, sbase, sn, ky, lo, narrow, base_lo);
#pragma unroll
        for (int u = 0; u < G; u++) {
          const int k = kk + u;
          const u32 chunk = (u32) k * NW + (u32) wid;
          const u32 r = (lo[u] * 43691u) >> 19; /* lo / 12, exact for multiples of 12 below 2^16 */
          const bool in = lo[u] < lim[u];
          const u32 oat = sbase[u] + (in ? lo[u] >> 2 : 0u);
          const u64 okey = (u64) lds32[oat] | ((u64) lds32[oat + 1] << 32);
          const u32 ocnt = lds32[oat + 2];
          const bool matched = in & (okey == ky[u]);
          if (FAST && (OPS == 2 || OPS == 4)) {
            /* A-only kernels with the rule folded in: live chunks are A chunks (`in` implies a valid
             * lane), and the kept records are decided by the match alone */
            u32 f;
            bool keep;
            if (OPS == 2 && FAST == 1) {
              f = own[u] < ocnt ? own[u] : ocnt;                  /* MIN */
              keep = matched && f >= (p.cutoff ? p.cutoff : 1u); /* both counts >= cutoff <=> min >= cutoff; and min != 0 */
            } else if (OPS == 2) {
              /* FAST = 2 / 3: a step of intersect_multi's left-to-right chain (reference src/glistcompare.c:655-678): the
               * running minimum restarts at 0 (`if (!freq || c < freq) freq = c`, :669 = RULE_MINZ); intermediate steps
               * keep every shared key (FAST = 2), the last one those whose count reaches the cutoff (FAST = 3, :683) */
              const u32 mn = own[u] < ocnt ? own[u] : ocnt;
              f = own[u] == 0u ? ocnt : mn;
              keep = matched && (FAST == 2 || f >= p.cutoff);
            } else {
              const u32 xb = matched ? ocnt : 0u;
              f = own[u] - xb;                                   /* SUBTRACT: kept only when f1 >= cutoff > f2 */
              keep = valid[u] && own[u] >= p.cutoff && xb < p.cutoff && f != 0u;
            }
            key[k] = ky[u];
            fa[k] = f;
            fb[k] = 0;
            meta[k] = 0;
            const u64 m = __builtin_amdgcn_ballot_w64 (keep);
            store_lane0 (&sh.kmask[OPS == 2 ? 1 : 2][chunk], sh.trash, m, lane);
            if (OPS == 2) acc_sum1 += keep ? f : 0u;
            else acc_sum2 += keep ? f : 0u;
            continue;
          }
          if (FAST && OPS == 1) {
            /* union with ADD folded in: an A record carries its partner's count, a B record with a
             * partner keeps nothing; FAST == 2 (intermediate N-way level) keeps zero sums too */
            const u32 xo = (is_a[u] && matched) ? ocnt : 0u; /* the partner's count travels with the A record */
            const u32 f = own[u] + xo;
            const bool keep = valid[u] && (is_a[u] || !matched) &&
                              (FAST == 2 || (FAST == 3 ? f >= p.cutoff : ((own[u] >= p.cutoff || xo >= p.cutoff) && f != 0u)));
            key[k] = ky[u];
            fa[k] = f;
            fb[k] = 0;
            meta[k] = r | (is_a[u] << 18);
            const u64 m = __builtin_amdgcn_ballot_w64 (keep);
            sh.kmask[0][chunk] = m;
            acc_sum0 += keep ? f : 0u;
            continue;
          }
          u32 kind, xa, xb;
          if (is_a[u]) {
            kind = matched ? KIND_BOTH : KIND_A;
            xa = own[u];
            xb = matched ? ocnt : 0u;
          } else {
            kind = matched ? KIND_SKIP : KIND_B;
            xa = 0;
            xb = own[u];
          }
          if (!valid[u]) kind = KIND_SKIP;
          key[k] = ky[u];
          fa[k] = xa;
          fb[k] = xb;
          meta[k] = (OPS == 2 || OPS == 4) ? 0u : (r | (kind << 16) | (is_a[u] << 18)); /* A-only kernels place by the own prefix alone */
          u32 f;
          if (ops & 1u) {
            const bool keep = (FAST && OPS == 0) ? eval_default<0> (kind, xa, xb, p.cutoff, f) : eval_stream<0> (kind, xa, xb, c0, f);
            const u64 m = __builtin_amdgcn_ballot_w64 (keep);
            sh.kmask[0][chunk] = m;
            acc_sum0 += keep ? f : 0u;
          }
          if ((!FAST || OPS == 0) && (ops & 2u)) {
            const bool keep = (FAST && OPS == 0) ? eval_default<1> (kind, xa, xb, p.cutoff, f) : eval_stream<1> (kind, xa, xb, c1, f);
            const u64 m = __builtin_amdgcn_ballot_w64 (keep);
            sh.kmask[1][chunk] = m;
            acc_sum1 += keep ? f : 0u;
          }
          if ((!FAST || OPS == 0) && (ops & 4u)) {
            const bool keep = (FAST && OPS == 0) ? eval_default<2> (kind, xa, xb, p.cutoff, f) : eval_stream<2> (kind, xa, xb, c2, f);
            const u64 m = __builtin_amdgcn_ballot_w64 (keep);
            sh.kmask[2][chunk] = m;
            acc_sum2 += keep ? f : 0u;
          }
          if ((!FAST || OPS == 0) && (ops & 8u)) {
            const bool keep = (FAST && OPS == 0) ? eval_default<3> (kind, xa, xb, p.cutoff, f) : eval_stream<3> (kind, xa, xb, c3, f);
            const u64 m = __builtin_amdgcn_ballot_w64 (keep);
            sh.kmask[3][chunk] = m;
            acc_sum3 += keep ? f : 0u;
          }
          if (OPS != 0) fa[k] = f; /* the one stream's count: staging does not evaluate the rule again */
        }
      }
    }
    /* cut the value-numbering link between phase 1 and phase 3: without it the compiler keeps every
     * stream's count of every record alive across phase 2 instead of recomputing it (2x the VGPRs) */
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      if (OPS == 0) asm volatile ("" : "+v"(fa[k]), "+v"(fb[k]), "+v"(meta[k]));
      else if (OPS == 1) asm volatile ("" : "+v"(fa[k]), "+v"(meta[k]));
      else asm volatile ("" : "+v"(fa[k]));
    }
    /* any-combination kernel: resolved BEFORE the rest of the fetch is issued -- the memory counter
     * retires in order, so looking at the words asked for at the top of the iteration waits for every
     * load issued before this point: only the two parts issued at the top with them */
    if (GDEFER && it >= 1 && wid >= 4 && wid < 8 && ((ops >> (wid - 4)) & 1u)) {
      const int s = wid - 4;
      const u32 pt = past[LAG - 1];
      u64 x;
      if (MODE == MODE_LOOKBACK) x = resolve_offset (agg + (u64) s * n_rows * WAVE, carry + (u64) s * (n_rows + 1), pt, lane, xagg, xcarry, ctl, spin_limit);
      else x = desc[4 * (u64) pt + s];
      if (lane == 0) sh.excl[s] = 12 * x; /* bytes */
    }
    if (STAGGER && nxt < ntl) {
#pragma unroll
      for (int j = (G == IPT ? F_TOP : F_TOP + F_MID); j <= NLOAD4; j++) fetch_part (tn, j); /* (a single search group has no middle) */
    }
    if (DEFER && n_have && wid == 4) {
      /* wavefront 4 asks for the words the next iteration's write-out needs (row counts and row carry
       * of the tile staged LAG - 1 iterations ago, published then) and does NOT look at them before
       * the next phase 0: issued behind the fetch, they cost no wait of their own */
      if (MODE == MODE_LOOKBACK) {
        const u64 prow = n_tile / WAVE;
        dagg = 0;
        if ((u32) lane < n_tile % WAVE) dagg = peek_u32 (&agg[(u64) S0 * n_rows * WAVE + prow * WAVE + lane]);
        dcarry = peek_u64 (&carry[(u64) S0 * (n_rows + 1) + prow]);
      } else {
        dcarry = __hip_atomic_load (&desc[4 * (u64) n_tile + S0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    PHASE_STAMP (3); /* phase 1 */
    __syncthreads (); /* B1: all input reads done */
    PHASE_STAMP (4); /* barrier B1 */

    /* ---- phase 2.  Single-output (deferred) kernels: EVERY wavefront scans the chunk ballots
     * itself and writes the same prefix table to LDS, so nobody has to wait for anybody -- no
     * barrier between ranking and staging; wavefront 0 also publishes the tile total. */
    u32 my_total = 0;
    if (DEFER) {
      my_total = chunk_scan<NCH> (sh.kmask[S0], sh.cpre[S0], lane);
      blk_cnt += my_total; /* every lane of every wavefront holds the same sum; the kernel-total reduction reads lane 0 of wave S0 */
      if (wid == 0 && MODE == MODE_LOOKBACK) { /* a uniform branch first: fifteen wavefronts skip the exec bookkeeping */
        if (lane == 0) publish_u32 (&agg[(u64) S0 * n_rows * WAVE + cur], AGG_READY | my_total);
      }
    }
    /* any-combination kernel: wavefront s owns stream s: chunk scan, tile total, publish */
    if (!DEFER && wid < 4 && ((ops >> wid) & 1u)) {
      const int s = wid;
      const u32 total = chunk_scan<NCH> (sh.kmask[s], sh.cpre[s], lane);
      if (lane == 0) {
        sh.tot[s] = total;
        blk_cnt += total;
        if (MODE == MODE_COUNT) {
          if (desc) desc[4 * (u64) cur + s] = total; /* pass 1 of the two-pass path: counts for the scan kernel */
        } else if (MODE == MODE_LOOKBACK) {
          publish_u32 (&agg[(u64) s * n_rows * WAVE + cur], AGG_READY | total);
        }
      }
    }

    PHASE_STAMP (5); /* phase 2 */
    if (GDEFER) {
      /* write the previous tile's streams out of the staging area, then (B2) stage this tile's */
      if (it >= 1) {
#pragma unroll
        for (int s = 0; s < 4; s++)
          if ((ops >> s) & 1u) write_out_fixed<NT, (3 * CAP / 4 + NT - 1) / NT> (outs.rec[s], uniform64 (sh.excl[s]), g_tot[s], sh.stage[0] + 3 * g_off[s], tid);
      }
      __syncthreads (); /* B2: staging area free, this tile's totals and prefix tables complete */
      u32 run = 0;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        g_off[s] = run;
        g_tot[s] = ((ops >> s) & 1u) ? uniform32 (sh.tot[s]) : 0u;
        run += (g_tot[s] + 3u) & ~3u;
      }
#pragma unroll
      for (int s = 0; s < 4; s++) {
        if (!((ops >> s) & 1u)) continue;
        u32 *const dst = sh.stage[0] + 3 * g_off[s];
        switch (s) {
          case 0: scatter_stream<0, NT, IPT, OPS, FAST> (sh, dst, p, nbs, lane, wid, key, fa, fb, meta); break;
          case 1: scatter_stream<1, NT, IPT, OPS, FAST> (sh, dst, p, nbs, lane, wid, key, fa, fb, meta); break;
          case 2: scatter_stream<2, NT, IPT, OPS, FAST> (sh, dst, p, nbs, lane, wid, key, fa, fb, meta); break;
          default: scatter_stream<3, NT, IPT, OPS, FAST> (sh, dst, p, nbs, lane, wid, key, fa, fb, meta); break;
        }
      }
    } else if (DEFER) {
      /* stage this tile in the slot the write-out at the top of this iteration freed */
      u32 *const slot = sh.stage[it % LAG];
      const u32 my_tot = my_total;
      scatter_stream<S0, NT, IPT, OPS, FAST> (sh, slot, p, nbs, lane, wid, key, fa, fb, meta);
#pragma unroll
      for (int q = 0; q + 1 < LAG; q++) pend_tot[q] = pend_tot[q + 1];
      pend_tot[LAG - 1] = my_tot;
    }
    PHASE_STAMP (6); /* (any-combination kernel: previous tile's write-out, B2) staging scatter */
    if (tid == 0) {
      sh.tile_id[s_nn] = hk_tile;
      if (hk_have_rng) {
        {
          sh.rng[s_nn][0] = 12 * hk_rng[0];
          sh.rng[s_nn][1] = 12 * hk_rng[1];
          sh.rng[s_nn][2] = hk_rng[2] - hk_rng[0];
          sh.rng[s_nn][3] = hk_rng[3] - hk_rng[1];
        }
      }
    }
#pragma unroll
    for (int q = 0; q + 1 < LAG; q++) past[q] = past[q + 1];
    past[LAG - 1] = cur;
    PHASE_STAMP (7); /* housekeeping results */
    cur = nxt;
    tr = tn;
    it++;
    {
      const int r_cur = r_nxt;
      r_nxt = r_nn;
      r_nn = r_cur == 0 ? 2 : r_cur - 1; /* 0 1 2 -> the slot before r_nxt */
    }
  }
PROF (
  if (tid == GT4_STAMP_TID)
    for (int i = 0; i < 8; i++) atomicAdd (&ctl->phase_cycles[i], ph[i]);
)

  if (GDEFER && it >= 1) {
    /* drain: the last tile's streams are still staged */
    const u32 pt = past[LAG - 1];
    __syncthreads ();
    if (wid < 4 && ((ops >> wid) & 1u)) {
      const int s = wid;
      u64 x;
      if (MODE == MODE_LOOKBACK) x = resolve_offset (agg + (u64) s * n_rows * WAVE, carry + (u64) s * (n_rows + 1), pt, lane, 0, 0, ctl, spin_limit);
      else x = desc[4 * (u64) pt + s];
      if (lane == 0) sh.excl[s] = x;
    }
    __syncthreads ();
#pragma unroll
    for (int s = 0; s < 4; s++)
      if ((ops >> s) & 1u) write_out_tile<NT> (outs.rec[s], uniform64 (sh.excl[s]), g_tot[s], sh.stage[0] + 3 * g_off[s], tid);
  }
  if (DEFER) {
    /* drain: the tiles still staged, oldest first */
#pragma unroll
    for (int q = 0; q < LAG; q++) {
      if (it - LAG + q < 0) continue;
      const u32 tile = past[q];
      const u32 tot = pend_tot[q];
      __syncthreads ();
      if (wid == 0) {
        u64 x;
        if (MODE == MODE_LOOKBACK) x = resolve_offset (agg + (u64) S0 * n_rows * WAVE, carry + (u64) S0 * (n_rows + 1), tile, lane, 0, 0, ctl, spin_limit);
        else x = desc[4 * (u64) tile + S0];
        if (lane == 0) sh.excl[S0] = x;
      }
      __syncthreads ();
      write_out_tile<NT> (outs.rec[S0], uniform64 (sh.excl[S0]), tot, sh.stage[(it + q) % LAG], tid);
    }
  }

  /* ---- kernel totals: header n_words / total_count (reference :801-802, :909-910) */
  {
    const u64 sums[4] = { acc_sum0, acc_sum1, acc_sum2, acc_sum3 };
#pragma unroll
    for (int s = 0; s < 4; s++) {
      if (!((ops >> s) & 1u)) continue;
      const u64 v = wave_sum (sums[s]);
      if (lane == 0 && v) atomicAdd (&ctl->total_count[s], v);
      if (lane == 0 && wid == s && blk_cnt) atomicAdd (&ctl->n_words[s], blk_cnt);
    }
  }
}

/* ------------------------------------------------------------------ K3: tile-count scan (two-pass path) */

constexpr int SCAN_NT = 256;
constexpr int SCAN_PER_BLOCK = SCAN_NT * 8;

/* desc[4*t + s] holds per-tile counts; phase 0 reduces each block's span per stream into block_sums,
 * phase 1 (one block) turns block_sums into exclusive offsets, phase 2 rewrites desc as exclusive offsets. */
__global__ __launch_bounds__ (SCAN_NT) void k_scan_reduce (const u64 *__restrict__ desc, u64 num_tiles, u64 *__restrict__ block_sums)
{
  __shared__ u64 red[4][SCAN_NT / WAVE];
  const u64 first = (u64) blockIdx.x * SCAN_PER_BLOCK;
  u64 s4[4] = { 0, 0, 0, 0 };
  for (u64 t = first + threadIdx.x; t < first + SCAN_PER_BLOCK && t < num_tiles; t += SCAN_NT)
    for (int s = 0; s < 4; s++) s4[s] += desc[4 * t + s];
  for (int s = 0; s < 4; s++) {
    const u64 v = wave_sum (s4[s]);
    if ((threadIdx.x & 63) == 0) red[s][threadIdx.x / WAVE] = v;
  }
  __syncthreads ();
  if (threadIdx.x < 4) {
    u64 v = 0;
    for (int w = 0; w < SCAN_NT / WAVE; w++) v += red[threadIdx.x][w];
    block_sums[4 * (u64) blockIdx.x + threadIdx.x] = v;
  }
}

__global__ void k_scan_blocks (u64 *__restrict__ block_sums, u64 n_blocks)
{
  /* few thousand entries at most: one thread per stream walks them */
  const int s = threadIdx.x;
  if (s >= 4) return;
  u64 run = 0;
  for (u64 b = 0; b < n_blocks; b++) {
    const u64 v = block_sums[4 * b + s];
    block_sums[4 * b + s] = run;
    run += v;
  }
}

__global__ __launch_bounds__ (SCAN_NT) void k_scan_apply (u64 *__restrict__ desc, u64 num_tiles, const u64 *__restrict__ block_sums)
{
  /* one wave per stream walks the block's span in 64-tile steps */
  const int s = threadIdx.x / WAVE, lane = threadIdx.x & 63;
  if (s >= 4) return;
  const u64 first = (u64) blockIdx.x * SCAN_PER_BLOCK;
  u64 run = block_sums[4 * (u64) blockIdx.x + s];
  for (u64 base = first; base < first + SCAN_PER_BLOCK && base < num_tiles; base += WAVE) {
    const u64 t = base + lane;
    const u64 v = t < num_tiles ? desc[4 * t + s] : 0;
    const u64 inc = wave_inclusive_scan (v, lane);
    if (t < num_tiles) desc[4 * t + s] = run + inc - v;
    run += __shfl ((u32) inc, 63, WAVE) | ((u64) __shfl ((u32) (inc >> 32), 63, WAVE) << 32);
  }
}

/* ------------------------------------------------------------------ K0: synthetic lists */

__device__ __forceinline__ u64 mix64 (u64 x)
{
  /* splitmix64 finaliser */
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

__global__ void k_generate (u32 *__restrict__ rec, u64 n, u64 stride, u64 seed, u64 count_seed, u32 max_count, u64 mult, u64 add)
{
  const u64 step = (u64) gridDim.x * blockDim.x;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const u64 key = (i * stride + mix64 (seed ^ (i * 0x2545f4914f6cdd1dull)) % stride) * mult + add;
    const u32 cnt = 1u + (u32) (mix64 (count_seed ^ (i * 0x9e3779b97f4a7c15ull)) % max_count);
    rec[3 * i] = (u32) key;
    rec[3 * i + 1] = (u32) (key >> 32);
    rec[3 * i + 2] = cnt;
  }
}

/* ------------------------------------------------------------------ small utilities */

__global__ void k_sum_counts (const u32 *__restrict__ rec, u64 n, u64 *sum)
{
  const u64 step = (u64) gridDim.x * blockDim.x;
  u64 acc = 0;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) acc += rec[3 * i + 2];
  acc = wave_sum (acc);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd (sum, acc);
}

__global__ void k_check_sorted (const u32 *__restrict__ rec, u64 n, u32 *bad)
{
  const u64 step = (u64) gridDim.x * blockDim.x;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n; i += step)
    if (load_key (rec, i) >= load_key (rec, i + 1)) atomicOr (bad, 1u);
}

__global__ void k_lower_bound (const u32 *__restrict__ rec, u64 n, u64 key, u64 *idx)
{
  if (threadIdx.x || blockIdx.x) return;
  u64 lo = 0, hi = n;
  while (lo < hi) {
    const u64 mid = (lo + hi) >> 1;
    if (load_key (rec, mid) < key) lo = mid + 1;
    else hi = mid;
  }
  *idx = lo;
}

/* column `column` of the per-key count table from a list aligned with the table's keys */
__global__ void k_extract_column (const u32 *__restrict__ rec, u64 n, u32 *__restrict__ counts, u32 n_lists, u32 column)
{
  const u64 step = (u64) gridDim.x * blockDim.x;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) counts[i * n_lists + column] = rec[3 * i + 2];
}

__global__ void k_extract_keys (const u32 *__restrict__ rec, u64 n, u64 *__restrict__ keys)
{
  const u64 step = (u64) gridDim.x * blockDim.x;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) keys[i] = load_key (rec, i);
}

/* GT4I index k-mer table -> packed list records: entry i = (word, first location); its count is
 * the distance to the next entry's first location, the last one's to num_locations, truncated to
 * 32 bits (imap_get_word / imap_get_count, reference src/index-map.c:123-139). */
__global__ void k_decode_index (const u64 *__restrict__ kmers, u64 n, u64 num_locations, u32 *__restrict__ rec)
{
  const u64 step = (u64) gridDim.x * blockDim.x;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const u64 word = kmers[2 * i], loc = kmers[2 * i + 1];
    const u64 next = i + 1 < n ? kmers[2 * i + 3] : num_locations;
    rec[3 * i] = (u32) word;
    rec[3 * i + 1] = (u32) (word >> 32);
    rec[3 * i + 2] = (u32) (next - loc);
  }
}

inline int grid_for (u64 n, int block, int cap)
{
  u64 g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > (u64) cap) g = cap;
  return (int) g;
}

}  // namespace

/* ------------------------------------------------------------------ host launchers */

hipError_t launch_partition (hipStream_t s, const uint32_t *A, uint64_t nA, const uint32_t *B, uint64_t nB,
                             uint64_t num_tiles, uint64_t tile_records, uint64_t *part)
{
  /* the coarse table lives behind the (num_tiles + 1) tile ranges in the same workspace */
  u64 *const coarse = (u64 *) part + 2 * (num_tiles + 1);
  const u64 n_coarse = (num_tiles + PART_COARSE - 1) / PART_COARSE + 1;
  hipLaunchKernelGGL (k_partition_coarse, dim3 ((unsigned) ((n_coarse + 255) / 256)), dim3 (256), 0, s, A, nA, B, nB, num_tiles, tile_records, coarse);
  const u64 threads = num_tiles + 1;
  hipLaunchKernelGGL (k_partition, dim3 ((unsigned) ((threads + 255) / 256)), dim3 (256), 0, s, A, nA, B, nB, num_tiles, tile_records,
                      (const u64 *) coarse, (u64 *) part);
  return hipGetLastError ();
}

/* Two workgroup geometries (measured): count-only calls run fastest with 512 threads
 * and 2048-record tiles (two workgroups per CU overlap their phases); calls that materialise
 * records run fastest with 1024 threads and 4096-record tiles (half as many tiles on the scan
 * chain, whose hop latency is fixed, and room for the staging slots in one workgroup per CU). */
/* the rule-folded variant a call takes (see FAST in k_pair_merge); 0: the general coefficient form */
static int fast_variant (int ops_cls, const PairParams &p)
{
  int fast = 0;
  if (p.filter == FILTER_REFERENCE) {
    if (ops_cls == 1 && p.rule[0] == 1u) fast = 1;
    if (ops_cls == 2 && p.rule[1] == 3u) fast = 1;
    if (ops_cls == 4 && p.rule[2] == 2u && !p.subtract) fast = 1;
  } else if (ops_cls == 1 && p.rule[0] == 1u) {
    fast = p.filter == FILTER_RAW ? 2 : 3; /* N-way union levels: keep every key / keep sums >= cutoff (union_multi, :574) */
  } else if (ops_cls == 2 && p.rule[1] == RULE_MINZ) {
    fast = p.filter == FILTER_RAW ? 2 : 3; /* the steps of intersect_multi's chain under its default rule (:655-683) */
  }
  /* any combination of outputs with every requested stream on its default rule, any cutoff, no -du */
  if (ops_cls == 0 && p.filter == FILTER_REFERENCE && !p.subtract && (!(p.ops & 1u) || p.rule[0] == 1u) && (!(p.ops & 2u) || p.rule[1] == 3u) &&
      (!(p.ops & 4u) || p.rule[2] == 2u) && (!(p.ops & 8u) || p.rule[3] == 2u))
    fast = 1;
  return fast;
}

template <int NT, int OPS>
static hipError_t launch_pair_merge_ops (hipStream_t s, int mode, int grid, const uint32_t *A, uint64_t nA, const uint32_t *B, uint64_t nB,
                                         const uint64_t *part, uint64_t num_tiles, const PairParams &p, const PairOutputs &o,
                                         unsigned long long *desc, PairControl *ctl)
{
  /* the commonest single-output calls take the variant with the rule folded in (see FAST) */
  const int fast = fast_variant (OPS, p);
  constexpr int F1 = 1, F2 = (OPS == 1 || OPS == 2) ? 2 : 0, F3 = (OPS == 1 || OPS == 2) ? 3 : 0;
#define GT4_LAUNCH_MERGE(M, F) hipLaunchKernelGGL ((k_pair_merge<NT, merge_ipt (NT, OPS), M, OPS, F>), dim3 (grid), dim3 (NT), 0, s, A, nA, B, nB, (u64 *) part, num_tiles, p, o, desc, ctl)
  if (OPS == 0 && fast == 1 && (p.ops == 3u || p.ops == 5u || p.ops == 15u) && (mode == MODE_COUNT ? NT == 512 : NT == 1024)) {
    /* the commonest output sets with the default rules (-u -i, -u -d, all four): the stream set is a
     * compile-time constant */
#define GT4_LAUNCH_SET(M, SET) hipLaunchKernelGGL ((k_pair_merge<NT, merge_ipt (NT, OPS), M, OPS, 1, (OPS == 0 ? SET : 0)>), dim3 (grid), dim3 (NT), 0, s, A, nA, B, nB, (u64 *) part, num_tiles, p, o, desc, ctl)
#define GT4_LAUNCH_SET_MODE(SET)                          \
    do {                                                  \
      if (mode == MODE_COUNT) GT4_LAUNCH_SET (MODE_COUNT, SET);          \
      else if (mode == MODE_LOOKBACK) GT4_LAUNCH_SET (MODE_LOOKBACK, SET); \
      else GT4_LAUNCH_SET (MODE_OFFSETS, SET);            \
    } while (0)
    if (p.ops == 3u) GT4_LAUNCH_SET_MODE (3);
    else if (p.ops == 5u) GT4_LAUNCH_SET_MODE (5);
    else GT4_LAUNCH_SET_MODE (15);
#undef GT4_LAUNCH_SET_MODE
#undef GT4_LAUNCH_SET
    return hipGetLastError ();
  }
  if (mode == MODE_COUNT) {
    if (fast == 1 && F1) GT4_LAUNCH_MERGE (MODE_COUNT, F1);
    else if (fast == 2 && F2) GT4_LAUNCH_MERGE (MODE_COUNT, F2);
    else if (fast == 3 && F3) GT4_LAUNCH_MERGE (MODE_COUNT, F3);
    else GT4_LAUNCH_MERGE (MODE_COUNT, 0);
  } else if (mode == MODE_LOOKBACK) {
    if (fast == 1 && F1) GT4_LAUNCH_MERGE (MODE_LOOKBACK, F1);
    else if (fast == 2 && F2) GT4_LAUNCH_MERGE (MODE_LOOKBACK, F2);
    else if (fast == 3 && F3) GT4_LAUNCH_MERGE (MODE_LOOKBACK, F3);
    else GT4_LAUNCH_MERGE (MODE_LOOKBACK, 0);
  } else {
    if (fast == 1 && F1) GT4_LAUNCH_MERGE (MODE_OFFSETS, F1);
    else if (fast == 2 && F2) GT4_LAUNCH_MERGE (MODE_OFFSETS, F2);
    else if (fast == 3 && F3) GT4_LAUNCH_MERGE (MODE_OFFSETS, F3);
    else GT4_LAUNCH_MERGE (MODE_OFFSETS, 0);
  }
#undef GT4_LAUNCH_MERGE
  return hipGetLastError ();
}

template <int NT, int OPS>
static int blocks_per_cu_ops (int mode)
{
  int n = 0;
  hipError_t e;
  if (mode == MODE_COUNT) e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_pair_merge<NT, merge_ipt (NT, OPS), MODE_COUNT, OPS>, NT, 0);
  else if (mode == MODE_LOOKBACK) e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_pair_merge<NT, merge_ipt (NT, OPS), MODE_LOOKBACK, OPS>, NT, 0);
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_pair_merge<NT, merge_ipt (NT, OPS), MODE_OFFSETS, OPS>, NT, 0);
  if (e != hipSuccess || n < 1) n = 1;
  /* never more than the register file admits for the declared launch bounds */
  const int by_regs = merge_waves_per_simd (NT, mode) * 4 / (NT / 64);
  if (by_regs >= 1 && n > by_regs) n = by_regs;
  return n;
}

/* the kernel specialisation of a set of outputs: one of {union, intersection, first complement} alone, else 0 */
static int ops_class (uint32_t ops) { return (ops == 1u || ops == 2u || ops == 4u) ? (int) ops : 0; }

uint64_t merge_tile_records (int geom, uint32_t ops)
{
  const int nt = geom ? 1024 : 512;
  return (uint64_t) nt * merge_ipt (nt, ops_class (ops)) - MERGE_TILE_SLACK;
}

/* workgroups of the merge kernel that are resident per CU (the single-pass path needs every
 * worker resident: see k_pair_merge) */
int merge_blocks_per_cu (int geom, int mode, uint32_t ops, const PairParams *p)
{
  /* the small geometry's folded intersection is built for three workgroups per CU (80 registers, 50 KB) */
  if (!geom && mode != MODE_COUNT && ops == 2u && p && fast_variant (2, *p) == 1) {
    static int c3 = 0;
    if (!c3) {
      int n = 0;
      const hipError_t e = mode == MODE_LOOKBACK ? hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_pair_merge<512, merge_ipt (512, 2), MODE_LOOKBACK, 2, 1>, 512, 0)
                                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_pair_merge<512, merge_ipt (512, 2), MODE_OFFSETS, 2, 1>, 512, 0);
      c3 = (e != hipSuccess || n < 1) ? 1 : (n > 3 ? 3 : n);
    }
    return c3;
  }
  static int cache[2][3][5];
  const int oi = ops_class (ops);
  int &c = cache[geom ? 1 : 0][mode][oi];
  if (!c) {
    if (geom) c = oi == 1 ? blocks_per_cu_ops<1024, 1> (mode) : (oi == 2 ? blocks_per_cu_ops<1024, 2> (mode) : (oi == 4 ? blocks_per_cu_ops<1024, 4> (mode) : blocks_per_cu_ops<1024, 0> (mode)));
    else c = oi == 1 ? blocks_per_cu_ops<512, 1> (mode) : (oi == 2 ? blocks_per_cu_ops<512, 2> (mode) : (oi == 4 ? blocks_per_cu_ops<512, 4> (mode) : blocks_per_cu_ops<512, 0> (mode)));
  }
  return c;
}

hipError_t launch_pair_merge (hipStream_t s, int geom, int mode, int grid, const uint32_t *A, uint64_t nA,
                              const uint32_t *B, uint64_t nB, const uint64_t *part, uint64_t num_tiles,
                              const PairParams &p, const PairOutputs &o, unsigned long long *desc,
                              PairControl *ctl)
{
  /* single-output calls (glistcompare -u / -i, every N-way level) take a specialised kernel */
  if (geom) {
    if (p.ops == 1u) return launch_pair_merge_ops<1024, 1> (s, mode, grid, A, nA, B, nB, part, num_tiles, p, o, desc, ctl);
    if (p.ops == 2u) return launch_pair_merge_ops<1024, 2> (s, mode, grid, A, nA, B, nB, part, num_tiles, p, o, desc, ctl);
    if (p.ops == 4u) return launch_pair_merge_ops<1024, 4> (s, mode, grid, A, nA, B, nB, part, num_tiles, p, o, desc, ctl);
    return launch_pair_merge_ops<1024, 0> (s, mode, grid, A, nA, B, nB, part, num_tiles, p, o, desc, ctl);
  }
  if (p.ops == 1u) return launch_pair_merge_ops<512, 1> (s, mode, grid, A, nA, B, nB, part, num_tiles, p, o, desc, ctl);
  if (p.ops == 2u) return launch_pair_merge_ops<512, 2> (s, mode, grid, A, nA, B, nB, part, num_tiles, p, o, desc, ctl);
  if (p.ops == 4u) return launch_pair_merge_ops<512, 4> (s, mode, grid, A, nA, B, nB, part, num_tiles, p, o, desc, ctl);
  return launch_pair_merge_ops<512, 0> (s, mode, grid, A, nA, B, nB, part, num_tiles, p, o, desc, ctl);
}

hipError_t launch_scan_tiles (hipStream_t s, unsigned long long *desc, uint64_t num_tiles, unsigned long long *block_sums)
{
  const u64 n_blocks = (num_tiles + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
  if (!n_blocks) return hipSuccess;
  hipLaunchKernelGGL (k_scan_reduce, dim3 ((unsigned) n_blocks), dim3 (SCAN_NT), 0, s, desc, num_tiles, block_sums);
  hipLaunchKernelGGL (k_scan_blocks, dim3 (1), dim3 (64), 0, s, block_sums, n_blocks);
  hipLaunchKernelGGL (k_scan_apply, dim3 ((unsigned) n_blocks), dim3 (SCAN_NT), 0, s, desc, num_tiles, block_sums);
  return hipGetLastError ();
}

hipError_t launch_generate (hipStream_t s, uint32_t *rec, uint64_t n, uint64_t stride, uint64_t seed, uint64_t count_seed,
                            uint32_t max_count, uint64_t mult, uint64_t add)
{
  hipLaunchKernelGGL (k_generate, dim3 (grid_for (n, 256, 8192)), dim3 (256), 0, s, rec, n, stride, seed, count_seed, max_count, mult, add);
  return hipGetLastError ();
}

hipError_t launch_sum_counts (hipStream_t s, const uint32_t *rec, uint64_t n, unsigned long long *sum)
{
  hipLaunchKernelGGL (k_sum_counts, dim3 (grid_for (n, 256, 4096)), dim3 (256), 0, s, rec, n, sum);
  return hipGetLastError ();
}

hipError_t launch_check_sorted (hipStream_t s, const uint32_t *rec, uint64_t n, unsigned int *bad)
{
  hipLaunchKernelGGL (k_check_sorted, dim3 (grid_for (n, 256, 4096)), dim3 (256), 0, s, rec, n, bad);
  return hipGetLastError ();
}

hipError_t launch_lower_bound (hipStream_t s, const uint32_t *rec, uint64_t n, uint64_t key, unsigned long long *idx)
{
  hipLaunchKernelGGL (k_lower_bound, dim3 (1), dim3 (64), 0, s, rec, n, key, idx);
  return hipGetLastError ();
}

hipError_t launch_extract_column (hipStream_t s, const uint32_t *rec, uint64_t n, uint32_t *counts, uint32_t n_lists, uint32_t column)
{
  hipLaunchKernelGGL (k_extract_column, dim3 (grid_for (n, 256, 4096)), dim3 (256), 0, s, rec, n, counts, n_lists, column);
  return hipGetLastError ();
}

hipError_t launch_extract_keys (hipStream_t s, const uint32_t *rec, uint64_t n, unsigned long long *keys)
{
  hipLaunchKernelGGL (k_extract_keys, dim3 (grid_for (n, 256, 4096)), dim3 (256), 0, s, rec, n, keys);
  return hipGetLastError ();
}

hipError_t launch_decode_index (hipStream_t s, const unsigned long long *kmers, uint64_t n, uint64_t num_locations, uint32_t *rec)
{
  hipLaunchKernelGGL (k_decode_index, dim3 (grid_for (n, 256, 4096)), dim3 (256), 0, s, (const u64 *) kmers, n, num_locations, rec);
  return hipGetLastError ();
}

}  // namespace gt4
