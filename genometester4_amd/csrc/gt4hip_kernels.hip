/*
 * gt4hip_kernels.hip -- hand-written gfx950 (MI355X, CDNA4) kernels for sorted k-mer list
 * set operations.  wave64 everywhere; no MFMA (integer streaming merge, HBM-bound).
 *
 * The hot path restates, per merged key, what the reference's single-threaded loop does
 * (compare_wordmaps, reference src/glistcompare.c:843-905; predicates :433-489):
 *
 *   K1 k_partition    merge-path co-ranking: tile t starts at (a_t, b_t) with a_t + b_t = t*TILE,
 *                     "A first on ties"; a matching A/B pair is never split between tiles.
 *   K2 k_pair_merge   persistent workgroups pull tiles by ticket; per tile: coalesced loads of the
 *                     two record ranges -> LDS (SoA: 8-byte-aligned keys + counts), one record per
 *                     lane: rank in the other list by LDS binary search, classification
 *                     {A only, B only, both}, up to four output predicates, wavefront ballots +
 *                     popcount prefix for output slots; a scanner wavefront chains the tile totals
 *                     into global output offsets; LDS-staged compaction, coalesced record stores.
 *   K3 k_scan_*       tile-count scan for the two-pass fallback.
 *   K0 k_generate     synthetic ascending lists written straight into HBM (bench only).
 */
#include "gt4hip_internal.h"

namespace gt4 {

namespace {

constexpr int WAVE = 64;

typedef unsigned long long u64;
typedef unsigned int u32;

/* ------------------------------------------------------------------ record access */

/* record i of a packed list viewed as dwords: key = words 3i, 3i+1; count = word 3i+2
 * (reference src/word-map.h:89-99: u64 at +0, u32 at +8, stride 12) */
__device__ __forceinline__ u64 load_key (const u32 *__restrict__ rec, u64 i)
{
  const u32 *p = rec + 3 * i;
  return (u64) p[0] | ((u64) p[1] << 32);
}

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

/* ------------------------------------------------------------------ K1: partition */

/* Number of A records among the first `diag` records of merge(A, B) with A first on ties. */
__device__ __forceinline__ u64 merge_path (const u32 *__restrict__ A, u64 nA, const u32 *__restrict__ B, u64 nB, u64 diag)
{
  u64 lo = diag > nB ? diag - nB : 0, hi = diag < nA ? diag : nA;
  while (lo < hi) {
    const u64 mid = (lo + hi) >> 1;
    if (load_key (A, mid) <= load_key (B, diag - 1 - mid)) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

__global__ void k_partition (const u32 *__restrict__ A, u64 nA, const u32 *__restrict__ B, u64 nB,
                             u64 num_tiles, u64 *__restrict__ part)
{
  const u64 t = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (t > num_tiles) return;
  const u64 total = nA + nB;
  u64 diag = t * (u64) MERGE_TILE;
  if (diag > total) diag = total;
  u64 a = merge_path (A, nA, B, nB, diag), b = diag - a;
  /* keys are unique inside a list, so a key present in both lists sits as an adjacent (A, B)
   * pair in the merged order; if the diagonal falls between them pull the B record back into
   * the earlier tile so that one tile owns the pair. */
  if (a > 0 && b < nB && load_key (A, a - 1) == load_key (B, b)) b += 1;
  part[2 * t] = a;
  part[2 * t + 1] = b;
}

/* ------------------------------------------------------------------ wave / block scans */

__device__ __forceinline__ u64 shfl_up_u64 (u64 v, int d)
{
  const u32 lo = __shfl_up ((u32) v, d, WAVE), hi = __shfl_up ((u32) (v >> 32), d, WAVE);
  return (u64) lo | ((u64) hi << 32);
}

__device__ __forceinline__ u64 shfl_xor_u64 (u64 v, int m)
{
  const u32 lo = __shfl_xor ((u32) v, m, WAVE), hi = __shfl_xor ((u32) (v >> 32), m, WAVE);
  return (u64) lo | ((u64) hi << 32);
}

__device__ __forceinline__ u64 wave_inclusive_scan (u64 v, int lane)
{
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const u64 o = shfl_up_u64 (v, d);
    if (lane >= d) v += o;
  }
  return v;
}

__device__ __forceinline__ u64 wave_sum (u64 v)
{
#pragma unroll
  for (int m = WAVE / 2; m > 0; m >>= 1) v += shfl_xor_u64 (v, m);
  return v;
}

/* ------------------------------------------------------------------ tile descriptors (chained scan) */

/* Per output stream s and tile t, two words in device memory, zeroed before every launch:
 *   agg[s][t]   u32: bit 31 = published, low bits = records the tile keeps in stream s
 *   excl[s][t]  u64: bit 63 = published, low bits = records kept by tiles 0..t-1 (global offset)
 * Workers publish agg; ONE scanner wavefront per stream walks agg in tile order and publishes
 * excl.  Every word is written once by a single relaxed agent-scope store and read by relaxed
 * agent-scope loads: value and flag travel in the same naturally aligned word, so no fence is
 * needed (cdna_hip_programming.md Guideline 16, form R2). */
constexpr u32 AGG_READY = 1u << 31;
constexpr u64 EXCL_READY = 1ull << 63;

__device__ __forceinline__ void publish_u32 (u32 *p, u32 v) { __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void publish_u64 (u64 *p, u64 v) { __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 peek_u32 (u32 *p) { return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 peek_u64 (u64 *p) { return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr u32 SPIN_LIMIT = 1u << 22; /* bounded: ~seconds; sets ctl->error instead of hanging */
constexpr int SCAN_ROWS = 16;        /* rows of 64 tiles a scanner wavefront keeps in flight */

/* The scanner: one wavefront per stream.  Loads SCAN_ROWS x 64 tile counts at once (so that its
 * rate is set by L2 bandwidth, not by one round trip per 64 tiles), waits for stragglers, scans,
 * publishes the exclusive offsets. */
__device__ void scanner_wave (u32 *agg, u64 *excl, u64 num_tiles, PairControl *ctl, int lane)
{
  u64 carry = 0;
  const u64 rows = (num_tiles + WAVE - 1) / WAVE;
  for (u64 r0 = 0; r0 < rows; r0 += SCAN_ROWS) {
    u32 v[SCAN_ROWS];
#pragma unroll
    for (int j = 0; j < SCAN_ROWS; j++) {
      const u64 idx = (r0 + j) * WAVE + lane;
      v[j] = idx < num_tiles ? peek_u32 (&agg[idx]) : AGG_READY;
    }
#pragma unroll
    for (int j = 0; j < SCAN_ROWS; j++) {
      if (r0 + j >= rows) break;
      const u64 idx = (r0 + j) * WAVE + lane;
      /* Publish every tile as soon as all tiles before it have reported -- never wait for the
       * whole row: a worker may be blocked on tile t's offset while it still holds the ticket of
       * tile t+1 (same row). */
      u32 published = 0, spins = 0, incl = 0;
      for (;;) {
        const u64 ready = __ballot ((v[j] & AGG_READY) != 0);
        const u32 f = ~ready ? (u32) __ffsll ((long long) ~ready) - 1u : (u32) WAVE; /* length of the ready prefix */
        if (f > published) {
          const u32 val = (u32) lane < f ? (v[j] & ~AGG_READY) : 0u;
          incl = val;
#pragma unroll
          for (int d = 1; d < WAVE; d <<= 1) {
            const u32 o = __shfl_up (incl, d, WAVE);
            if (lane >= d) incl += o;
          }
          if ((u32) lane >= published && (u32) lane < f && idx < num_tiles) publish_u64 (&excl[idx], EXCL_READY | (carry + incl - val));
          published = f;
        }
        if (f == (u32) WAVE) break;
        if (++spins > SPIN_LIMIT) {
          if (lane == 0) atomicOr (&ctl->error, 4u);
          return;
        }
        if (spins > 2) __builtin_amdgcn_s_sleep (2);
        if (!(v[j] & AGG_READY)) v[j] = peek_u32 (&agg[idx]);
      }
      carry += __shfl (incl, WAVE - 1, WAVE);
    }
  }
}

/* ------------------------------------------------------------------ K2: tile merge by rank search */

/*
 * Persistent workgroups; each processes one merge-path tile (<= CAP records of A and B together)
 * at a time, tiles handed out in index order by an atomic ticket:
 *
 *  phase 0  the tile's two packed record ranges, fetched one tile AHEAD into registers with
 *           coalesced dword loads, go to LDS as SoA (keys[] 8-byte aligned, counts[]); A records
 *           first, then B records.  The next tile's loads are issued right away and stay in
 *           flight during phases 1-3.
 *  phase 1  one record per lane (striped: consecutive lanes hold consecutive sorted keys, so the
 *           64 binary searches of a wavefront walk nearly the same LDS addresses -> broadcasts, few
 *           bank conflicts): rank r = number of records of the OTHER list with a smaller key,
 *           match test at r, classification {A only, B only, both}, per-stream keep predicate;
 *           the keep flags of each 64-record chunk go to LDS as one wavefront ballot.
 *  phase 2  one wavefront per stream: popcount-scan of the chunk ballots, tile total published
 *           for the scanner wavefront (chained scan, above).
 *  phase 3  output slot of a kept record = (kept records before it in its own list)
 *           + (kept records of the other list before its rank) -- two ballot/prefix lookups, no
 *           sort; records are scattered into an LDS staging area in output order and leave with
 *           coalesced dword stores at the tile's global offset.
 *
 * Single-output kernels (OPS = union or intersection) DEFER the write-out of tile i until tile
 * i+1 has been ranked: by then the scanner has long published tile i's offset, so no wavefront
 * ever waits on the chain.  The any-combination kernel (OPS = 0) writes out in place (staging
 * aliases the dead input view) and waits for its offset right after publishing its totals.
 *
 * Keys are unique inside a list (reference precondition), so "both" pairs are found by the match
 * test alone; the B record of a pair keeps nothing (its A partner carries both counts).
 */
template <int NT, int IPT, int OPS>
struct RankShared {
  static constexpr int CAP = NT * IPT;
  static constexpr int NCH = CAP / WAVE;
  /* deferred staging: an intersection keeps at most one record per pair, a union at most CAP */
  static constexpr int STAGE_DW = OPS == 2 ? 3 * (CAP / 2 + 1) : (OPS == 1 ? 3 * CAP : 4);
  u64 keys[CAP];          /* input view; OPS == 0: the output view (3 * CAP dwords) starts here too */
  u32 cnts[CAP];
  u32 stage[STAGE_DW];
  u64 kmask[4][NCH + 1];  /* keep-flag ballot per 64-record chunk, per stream (+1: empty sentinel chunk) */
  u32 cpre[4][NCH + 1];   /* exclusive prefix of popcount(kmask) over chunks, per stream               */
  u64 excl[4];            /* global exclusive output offset of the tile being written out              */
  u32 tot[4];             /* records the current tile keeps, per stream                                */
  u32 tick[2];            /* ticket ring: tile of the next iteration / the one after                   */
};

/* number of kept records among the concatenated tile positions [0, z) */
__device__ __forceinline__ u32 kept_before (const u64 *km, const u32 *cp, u32 z)
{
  const u32 c = z >> 6;
  return cp[c] + (u32) __popcll (km[c] & ((1ull << (z & 63u)) - 1ull));
}

template <int S, int NT, int IPT, int OPS>
__device__ __forceinline__ void scatter_stream (RankShared<NT, IPT, OPS> &sh, u32 *dst32, const PairParams &p, u32 na, int lane, int wid,
                                                const u64 (&key)[IPT], const u32 (&fa)[IPT], const u32 (&fb)[IPT], const u32 (&meta)[IPT])
{
  constexpr int NW = NT / WAVE;
  const StreamCoef c = make_coef<S> (p);
  const u32 pna = kept_before (sh.kmask[S], sh.cpre[S], na);
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const u32 chunk = (u32) k * NW + (u32) wid;
    const u64 m = sh.kmask[S][chunk];
    if ((m >> lane) & 1ull) {
      const u32 own = sh.cpre[S][chunk] + (u32) __popcll (m & ((1ull << lane) - 1ull));
      const u32 r = meta[k] & 0xffffu;
      const u32 z = ((meta[k] >> 18) & 1u) ? na + r : r;
      const u32 slot = own + kept_before (sh.kmask[S], sh.cpre[S], z) - pna;
      u32 f;
      eval_stream<S> ((meta[k] >> 16) & 3u, fa[k], fb[k], c, f);
      dst32[3 * slot] = (u32) key[k];
      dst32[3 * slot + 1] = (u32) (key[k] >> 32);
      dst32[3 * slot + 2] = f;
    }
  }
}

struct TileRange {
  u64 a0, b0;
  u32 na, nb;
};

__device__ __forceinline__ TileRange load_tile_range (const u64 *__restrict__ part, u64 tile)
{
  TileRange t;
  t.a0 = part[2 * tile];
  t.b0 = part[2 * tile + 1];
  t.na = (u32) (part[2 * tile + 2] - t.a0);
  t.nb = (u32) (part[2 * tile + 3] - t.b0);
  return t;
}

/* OPS != 0 fixes the set of output streams at compile time (the common single-output calls get a
 * kernel without the other streams' code and registers); OPS == 0 takes it from p.ops. */
template <int NT, int IPT, int MODE, int OPS>
__global__ __launch_bounds__ (NT, OPS ? MERGE_WAVES_PER_SIMD : MERGE_WAVES_PER_SIMD_GENERIC) void
k_pair_merge (const u32 *__restrict__ A, u64 nA, const u32 *__restrict__ B, u64 nB, const u64 *__restrict__ part, u64 num_tiles,
              PairParams p, PairOutputs outs, u64 *desc, PairControl *ctl)
{
  constexpr int CAP = NT * IPT;
  constexpr int NW = NT / WAVE;
  constexpr int NCH = CAP / WAVE;
  constexpr int NLOAD = 3 * IPT;                 /* dwords each thread fetches per tile */
  constexpr bool DEFER = (OPS == 1 || OPS == 2) && MODE != MODE_COUNT;
  constexpr int S0 = OPS == 2 ? 1 : 0;           /* the stream of a single-output kernel */
  static_assert (NW >= 4, "one wavefront per output stream in phase 2");
  static_assert (NCH <= WAVE, "chunk scan is a single wavefront pass");
  __shared__ RankShared<NT, IPT, OPS> sh;
  u32 *const lds32 = reinterpret_cast<u32 *> (&sh.keys[0]);

  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
  const u32 ops = OPS ? (u32) OPS : p.ops;
  /* every "both" pair is evaluated at its A record, so A records always matter; B records only
   * where a B-only key can be kept (union, diff2) */
  const bool need_b = (ops & 9u) != 0;

  /* single pass: agg u32[4][T] then excl u64[4][T] inside desc (zeroed by the host) */
  u32 *const agg = reinterpret_cast<u32 *> (desc);
  u64 *const excl = desc + 2 * num_tiles; /* 4 * T u32 = 2 * T u64 */

  if (MODE == MODE_LOOKBACK) {
    /* role election: the first workgroup to arrive is running by definition, it becomes the scanner */
    if (tid == 0) sh.tick[0] = atomicAdd (&ctl->role, 1u);
    __syncthreads ();
    const u32 role = sh.tick[0];
    __syncthreads ();
    if (role == 0) {
      if (wid < 4 && ((ops >> wid) & 1u)) scanner_wave (agg + (u64) wid * num_tiles, excl + (u64) wid * num_tiles, num_tiles, ctl, lane);
      return;
    }
  }

  u64 acc_sum0 = 0, acc_sum1 = 0, acc_sum2 = 0, acc_sum3 = 0; /* per-thread sums of kept counts */
  u64 blk_cnt = 0;                                              /* lane 0 of wave s: records kept in stream s */

  /* tickets: tiles are claimed in index order, so every tile the scanner waits on belongs to a
   * workgroup that is already running (no residency assumption).  Two tickets are held: the tile
   * being processed and the one whose records are being prefetched. */
  if (tid == 0) {
    sh.tick[0] = atomicAdd (&ctl->ticket, 1u);
    sh.tick[1] = atomicAdd (&ctl->ticket, 1u);
  }
  __syncthreads ();
  u64 cur = sh.tick[0];
  TileRange tr = { 0, 0, 0, 0 };
  u32 pre[NLOAD];

  auto fetch = [&] (const TileRange &t) {
    const u32 *__restrict__ srcA = A + 3 * t.a0;
    const u32 *__restrict__ srcB = B + 3 * t.b0;
    const u32 da = 3 * t.na, dt = 3 * (t.na + t.nb);
#pragma unroll
    for (int j = 0; j < NLOAD; j++) {
      const u32 d = (u32) j * NT + (u32) tid;
      u32 w = 0;
      if (d < da) w = srcA[d];
      else if (d < dt) w = srcB[d - da];
      pre[j] = w;
    }
  };

  if (cur < num_tiles) {
    tr = load_tile_range (part, cur);
    fetch (tr);
  }
  u64 prev_tile = 0;   /* DEFER: tile whose output is staged but not yet written */
  u32 prev_tot = 0;
  bool have_prev = false;
  int it = 0;

  while (cur < num_tiles) {
    const u32 na = tr.na, nb = tr.nb, nt = na + nb;
    if (nt > (u32) CAP) {
      if (tid == 0) atomicOr (&ctl->error, 2u);
      break;
    }
    /* ---- phase 0: registers -> LDS, AoS dwords -> SoA (an 8-byte key read at a 4-byte aligned
     * LDS address would replay at 64 cycles, Guideline 17) */
#pragma unroll
    for (int j = 0; j < NLOAD; j++) {
      const u32 d = (u32) j * NT + (u32) tid;
      if (d < 3 * nt) {
        const u32 r = d / 3, f = d - 3 * r;
        lds32[(f == 2) ? (2 * CAP + r) : (2 * r + f)] = pre[j];
      }
    }
    const u64 nxt = sh.tick[(it + 1) & 1];
    __syncthreads (); /* B0 */
    if (tid == 0) sh.tick[it & 1] = atomicAdd (&ctl->ticket, 1u);
    TileRange tn = { 0, 0, 0, 0 };
    if (nxt < num_tiles) {
      tn = load_tile_range (part, nxt);
      fetch (tn); /* in flight until the next iteration's phase 0 */
    }

    /* ---- phase 1: rank, classify, predicates */
    u64 key[IPT];
    u32 fa[IPT], fb[IPT], meta[IPT]; /* meta: rank | kind << 16 | is_a << 18 */
    {
      const StreamCoef c0 = make_coef<0> (p), c1 = make_coef<1> (p), c2 = make_coef<2> (p), c3 = make_coef<3> (p);
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const u32 e = (u32) k * NT + (u32) tid;
        const u32 chunk = (u32) k * NW + (u32) wid; /* wave-uniform */
        const u32 cbeg = chunk * WAVE;
        const bool chunk_live = cbeg < nt && (need_b || cbeg < na);
        u64 ky = 0;
        u32 xa = 0, xb = 0, kind = KIND_SKIP, r = 0, is_a = 0;
        if (chunk_live) {
          const bool valid = e < nt;
          const u32 ec = valid ? e : 0u;
          is_a = ec < na ? 1u : 0u;
          ky = sh.keys[ec];
          const u32 own = sh.cnts[ec];
          const u32 obase = is_a ? na : 0u, on = is_a ? nb : na;
          /* lower bound of ky in the other list */
          u32 lo = 0, len = valid ? on : 0u;
          while (len > 0) {
            const u32 half = len >> 1;
            if (sh.keys[obase + lo + half] < ky) {
              lo += half + 1;
              len -= half + 1;
            } else {
              len = half;
            }
          }
          r = lo;
          const bool in = r < on;
          const u32 oat = obase + (in ? r : 0u);
          const bool matched = in && sh.keys[oat] == ky;
          const u32 ocnt = sh.cnts[oat];
          if (is_a) {
            kind = matched ? KIND_BOTH : KIND_A;
            xa = own;
            xb = matched ? ocnt : 0u;
          } else {
            kind = matched ? KIND_SKIP : KIND_B;
            xb = own;
          }
          if (!valid) kind = KIND_SKIP;
        }
        key[k] = ky;
        fa[k] = xa;
        fb[k] = xb;
        meta[k] = r | (kind << 16) | (is_a << 18);
        u32 f;
        if (ops & 1u) {
          const bool keep = eval_stream<0> (kind, xa, xb, c0, f);
          const u64 m = __ballot (keep);
          if (lane == 0) sh.kmask[0][chunk] = m;
          acc_sum0 += keep ? f : 0u;
        }
        if (ops & 2u) {
          const bool keep = eval_stream<1> (kind, xa, xb, c1, f);
          const u64 m = __ballot (keep);
          if (lane == 0) sh.kmask[1][chunk] = m;
          acc_sum1 += keep ? f : 0u;
        }
        if (ops & 4u) {
          const bool keep = eval_stream<2> (kind, xa, xb, c2, f);
          const u64 m = __ballot (keep);
          if (lane == 0) sh.kmask[2][chunk] = m;
          acc_sum2 += keep ? f : 0u;
        }
        if (ops & 8u) {
          const bool keep = eval_stream<3> (kind, xa, xb, c3, f);
          const u64 m = __ballot (keep);
          if (lane == 0) sh.kmask[3][chunk] = m;
          acc_sum3 += keep ? f : 0u;
        }
      }
    }
    /* cut the value-numbering link between phase 1 and phase 3: without it the compiler keeps every
     * stream's count of every record alive across phase 2 instead of recomputing it (2x the VGPRs) */
#pragma unroll
    for (int k = 0; k < IPT; k++) asm volatile ("" : "+v"(fa[k]), "+v"(fb[k]), "+v"(meta[k]));
    __syncthreads (); /* B1: all input reads done */

    /* ---- phase 2: wavefront s owns stream s: chunk scan, tile total, publish for the scanner */
    if (wid < 4 && ((ops >> wid) & 1u)) {
      const int s = wid;
      const u32 v = lane < NCH ? (u32) __popcll (sh.kmask[s][lane]) : 0u;
      const u32 incl = (u32) wave_inclusive_scan (v, lane);
      const u32 total = __shfl (incl, WAVE - 1, WAVE);
      if (lane < NCH) sh.cpre[s][lane] = incl - v;
      if (lane == 0) {
        sh.cpre[s][NCH] = total;
        sh.kmask[s][NCH] = 0;
        sh.tot[s] = total;
        blk_cnt += total;
        if (MODE == MODE_COUNT) {
          if (desc) desc[4 * cur + s] = total; /* pass 1 of the two-pass path: counts for the scan kernel */
        } else if (MODE == MODE_LOOKBACK) {
          publish_u32 (&agg[(u64) s * num_tiles + cur], AGG_READY | total);
        }
      }
      if (MODE != MODE_COUNT && !DEFER) {
        /* the tile's own offset: wait for the scanner (or read the pre-scanned offsets) */
        if (lane == 0) {
          u64 x;
          if (MODE == MODE_LOOKBACK) {
            u64 *const w = &excl[(u64) s * num_tiles + cur];
            u32 spins = 0;
            while (!((x = peek_u64 (w)) & EXCL_READY)) {
              if (++spins > SPIN_LIMIT) {
                atomicOr (&ctl->error, 1u);
                break;
              }
              __builtin_amdgcn_s_sleep (2);
            }
            x &= ~EXCL_READY;
          } else {
            x = desc[4 * cur + s];
          }
          sh.excl[s] = x;
        }
      }
    } else if (DEFER && have_prev && wid == 4) {
      /* offset of the PREVIOUS tile, published long ago by the scanner */
      if (lane == 0) {
        u64 x;
        if (MODE == MODE_LOOKBACK) {
          u64 *const w = &excl[(u64) S0 * num_tiles + prev_tile];
          u32 spins = 0;
          while (!((x = peek_u64 (w)) & EXCL_READY)) {
            if (++spins > SPIN_LIMIT) {
              atomicOr (&ctl->error, 1u);
              break;
            }
            __builtin_amdgcn_s_sleep (2);
          }
          x &= ~EXCL_READY;
        } else {
          x = desc[4 * prev_tile + S0];
        }
        sh.excl[S0] = x;
      }
    }

    if (MODE != MODE_COUNT) {
      __syncthreads (); /* B2 */
      if (DEFER) {
        /* write out the previous tile from the staging area, then stage this one */
        if (have_prev) {
          u32 *__restrict__ dst = outs.rec[S0] + 3 * sh.excl[S0];
          const u32 nd = 3 * prev_tot;
          for (u32 d = tid; d < nd; d += NT) dst[d] = sh.stage[d];
        }
        const u32 my_tot = sh.tot[S0];
        __syncthreads (); /* B3: staging area free */
        if (S0 == 0) scatter_stream<0, NT, IPT, OPS> (sh, sh.stage, p, na, lane, wid, key, fa, fb, meta);
        else scatter_stream<1, NT, IPT, OPS> (sh, sh.stage, p, na, lane, wid, key, fa, fb, meta);
        prev_tile = cur;
        prev_tot = my_tot;
        have_prev = true;
      } else {
#pragma unroll
        for (int s = 0; s < 4; s++) {
          if (!((ops >> s) & 1u)) continue;
          switch (s) {
            case 0: scatter_stream<0, NT, IPT, OPS> (sh, lds32, p, na, lane, wid, key, fa, fb, meta); break;
            case 1: scatter_stream<1, NT, IPT, OPS> (sh, lds32, p, na, lane, wid, key, fa, fb, meta); break;
            case 2: scatter_stream<2, NT, IPT, OPS> (sh, lds32, p, na, lane, wid, key, fa, fb, meta); break;
            default: scatter_stream<3, NT, IPT, OPS> (sh, lds32, p, na, lane, wid, key, fa, fb, meta); break;
          }
          __syncthreads ();
          u32 *__restrict__ dst = outs.rec[s] + 3 * sh.excl[s];
          const u32 nd = 3 * sh.tot[s];
          for (u32 d = tid; d < nd; d += NT) dst[d] = lds32[d];
          __syncthreads ();
        }
      }
    }
    cur = nxt;
    tr = tn;
    it++;
  }

  if (DEFER && have_prev) {
    /* drain: the last staged tile */
    __syncthreads ();
    if (tid == 0) {
      u64 x;
      if (MODE == MODE_LOOKBACK) {
        u64 *const w = &excl[(u64) S0 * num_tiles + prev_tile];
        u32 spins = 0;
        while (!((x = peek_u64 (w)) & EXCL_READY)) {
          if (++spins > SPIN_LIMIT) {
            atomicOr (&ctl->error, 1u);
            break;
          }
          __builtin_amdgcn_s_sleep (2);
        }
        x &= ~EXCL_READY;
      } else {
        x = desc[4 * prev_tile + S0];
      }
      sh.excl[S0] = x;
    }
    __syncthreads ();
    u32 *__restrict__ dst = outs.rec[S0] + 3 * sh.excl[S0];
    const u32 nd = 3 * prev_tot;
    for (u32 d = tid; d < nd; d += NT) dst[d] = sh.stage[d];
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

/* column `column` of the per-key count table: count of each union key in `list`, or 0 */
__global__ void k_counts_table (const u32 *__restrict__ keys_rec, u64 n_keys, const u32 *__restrict__ list, u64 n_list,
                                u32 *__restrict__ counts, u32 n_lists, u32 column)
{
  const u64 step = (u64) gridDim.x * blockDim.x;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n_keys; i += step) {
    const u64 key = load_key (keys_rec, i);
    u64 lo = 0, hi = n_list;
    while (lo < hi) {
      const u64 mid = (lo + hi) >> 1;
      if (load_key (list, mid) < key) lo = mid + 1;
      else hi = mid;
    }
    counts[i * n_lists + column] = (lo < n_list && load_key (list, lo) == key) ? list[3 * lo + 2] : 0u;
  }
}

__global__ void k_extract_keys (const u32 *__restrict__ rec, u64 n, u64 *__restrict__ keys)
{
  const u64 step = (u64) gridDim.x * blockDim.x;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) keys[i] = load_key (rec, i);
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
                             uint64_t num_tiles, uint64_t *part)
{
  const u64 threads = num_tiles + 1;
  const unsigned grid = (unsigned) ((threads + 255) / 256);
  hipLaunchKernelGGL (k_partition, dim3 (grid), dim3 (256), 0, s, A, nA, B, nB, num_tiles, (u64 *) part);
  return hipGetLastError ();
}

template <int OPS>
static hipError_t launch_pair_merge_ops (hipStream_t s, int mode, int grid, const uint32_t *A, uint64_t nA, const uint32_t *B, uint64_t nB,
                                         const uint64_t *part, uint64_t num_tiles, const PairParams &p, const PairOutputs &o,
                                         unsigned long long *desc, PairControl *ctl)
{
  if (mode == MODE_COUNT)
    hipLaunchKernelGGL ((k_pair_merge<MERGE_NT, MERGE_VT, MODE_COUNT, OPS>), dim3 (grid), dim3 (MERGE_NT), 0, s, A, nA, B, nB,
                        (const u64 *) part, num_tiles, p, o, desc, ctl);
  else if (mode == MODE_LOOKBACK)
    hipLaunchKernelGGL ((k_pair_merge<MERGE_NT, MERGE_VT, MODE_LOOKBACK, OPS>), dim3 (grid), dim3 (MERGE_NT), 0, s, A, nA, B, nB,
                        (const u64 *) part, num_tiles, p, o, desc, ctl);
  else
    hipLaunchKernelGGL ((k_pair_merge<MERGE_NT, MERGE_VT, MODE_OFFSETS, OPS>), dim3 (grid), dim3 (MERGE_NT), 0, s, A, nA, B, nB,
                        (const u64 *) part, num_tiles, p, o, desc, ctl);
  return hipGetLastError ();
}

int merge_blocks_per_cu ()
{
  static int cached = 0;
  if (!cached) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_pair_merge<MERGE_NT, MERGE_VT, MODE_LOOKBACK, 0>, MERGE_NT, 0) != hipSuccess || n < 1) n = 1;
    cached = n;
  }
  return cached;
}

hipError_t launch_pair_merge (hipStream_t s, int mode, int grid, const uint32_t *A, uint64_t nA,
                              const uint32_t *B, uint64_t nB, const uint64_t *part, uint64_t num_tiles,
                              const PairParams &p, const PairOutputs &o, unsigned long long *desc,
                              PairControl *ctl)
{
  /* single-output calls (glistcompare -u / -i, every N-way level) take a specialised kernel */
  if (p.ops == 1u) return launch_pair_merge_ops<1> (s, mode, grid, A, nA, B, nB, part, num_tiles, p, o, desc, ctl);
  if (p.ops == 2u) return launch_pair_merge_ops<2> (s, mode, grid, A, nA, B, nB, part, num_tiles, p, o, desc, ctl);
  return launch_pair_merge_ops<0> (s, mode, grid, A, nA, B, nB, part, num_tiles, p, o, desc, ctl);
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

hipError_t launch_counts_table (hipStream_t s, const uint32_t *keys_rec, uint64_t n_keys, const uint32_t *list,
                                uint64_t n_list, uint32_t *counts, uint32_t n_lists, uint32_t column)
{
  hipLaunchKernelGGL (k_counts_table, dim3 (grid_for (n_keys, 256, 4096)), dim3 (256), 0, s, keys_rec, n_keys, list, n_list,
                      counts, n_lists, column);
  return hipGetLastError ();
}

hipError_t launch_extract_keys (hipStream_t s, const uint32_t *rec, uint64_t n, unsigned long long *keys)
{
  hipLaunchKernelGGL (k_extract_keys, dim3 (grid_for (n, 256, 4096)), dim3 (256), 0, s, rec, n, keys);
  return hipGetLastError ();
}

}  // namespace gt4
