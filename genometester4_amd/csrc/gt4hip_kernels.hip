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
 *                     two record ranges -> LDS (SoA: 8-byte-aligned keys + counts), per-thread
 *                     merge-path search + serial merge of VT items in LDS, classification
 *                     {A only, B only, both}, up to four output predicates, packed block scan,
 *                     decoupled look-back for the tile's global output offsets, LDS-staged
 *                     compaction and coalesced record stores.
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

/* per-thread pass over its VT merged items for one stream: which are kept, and their count sum */
template <int S, int VT>
__device__ __forceinline__ void count_stream (u32 kinds, const u32 (&fa)[VT], const u32 (&fb)[VT], const PairParams &p, u32 &mask, u64 &sum)
{
  /* an empty volatile asm pins this body behind its wave-uniform branch: without it the compiler
   * speculates all four streams' bodies into one straight-line block (3x the registers) */
  asm volatile ("" ::: "memory");
  const StreamCoef c = make_coef<S> (p);
#pragma unroll
  for (int i = 0; i < VT; i++) {
    u32 f;
    const bool keep = eval_stream<S> ((kinds >> (2 * i)) & 3u, fa[i], fb[i], c, f);
    mask |= keep ? (1u << i) : 0u;
    sum += keep ? f : 0u;
  }
}

/* writes this thread's kept records of one stream into the LDS output view at slots pos, pos+1, ... */
template <int S, int VT>
__device__ __forceinline__ void scatter_stream (u32 kinds, u32 mask, u32 pos, const u64 (&key)[VT], const u32 (&fa)[VT],
                                                const u32 (&fb)[VT], const PairParams &p, u32 *lds32)
{
  asm volatile ("" ::: "memory");
  const StreamCoef c = make_coef<S> (p);
#pragma unroll
  for (int i = 0; i < VT; i++) {
    u32 f;
    eval_stream<S> ((kinds >> (2 * i)) & 3u, fa[i], fb[i], c, f);
    if ((mask >> i) & 1u) {
      lds32[3 * pos] = (u32) key[i];
      lds32[3 * pos + 1] = (u32) (key[i] >> 32);
      lds32[3 * pos + 2] = f;
      pos++;
    }
  }
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

/* ------------------------------------------------------------------ look-back descriptors */

/* One 64-bit word per (tile, stream): status in bits 63:62, value in bits 61:0.  Written and read
 * as single relaxed agent-scope 8-byte accesses: the value IS the flag, so no fence is needed
 * (cdna_hip_programming.md Guideline 16, form R2). */
constexpr u64 DESC_AGG = 1ull << 62;    /* tile's own count is known          */
constexpr u64 DESC_PREFIX = 2ull << 62; /* inclusive prefix over tiles 0..t   */
constexpr u64 DESC_VALUE = (1ull << 62) - 1;

__device__ __forceinline__ void desc_store (u64 *p, u64 v)
{
  __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ u64 desc_load (u64 *p)
{
  return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr u32 SPIN_LIMIT = 1u << 22; /* bounded: ~seconds; sets ctl->error instead of hanging */

/* ------------------------------------------------------------------ K2: tile merge */

template <int NT, int VT>
struct MergeShared {
  static constexpr int CAP = NT * VT;
  /* input view: keys[CAP] (u64, 8-byte aligned) followed by counts[CAP] (u32);
   * output view (after the merge, same bytes): 3*CAP dwords of packed records */
  u64 keys[CAP];
  u32 cnts[CAP];
  u64 wave_tot[NT / WAVE];
  u64 tile_excl[4];   /* global exclusive offset of this tile per stream */
  u32 tile;           /* ticket */
};

template <int NT, int VT, int MODE>
__global__ __launch_bounds__ (NT, MERGE_WAVES_PER_SIMD) void k_pair_merge (const u32 *__restrict__ A, u64 nA, const u32 *__restrict__ B, u64 nB,
                                                      const u64 *__restrict__ part, u64 num_tiles, PairParams p,
                                                      PairOutputs outs, u64 *desc, PairControl *ctl)
{
  constexpr int CAP = NT * VT;
  constexpr int NW = NT / WAVE;
  __shared__ MergeShared<NT, VT> sh;
  u32 *const lds32 = reinterpret_cast<u32 *> (&sh.keys[0]);

  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;

  u64 acc_sum0 = 0, acc_sum1 = 0, acc_sum2 = 0, acc_sum3 = 0; /* per-thread sums of emitted counts */
  u64 blk_cnt[4] = { 0, 0, 0, 0 };                              /* thread 0: records emitted by this workgroup */

  for (;;) {
    /* ---- ticket: tiles are claimed in index order, so every predecessor a look-back waits on
     * belongs to a workgroup that is already running (no residency assumption). */
    if (tid == 0) sh.tile = atomicAdd (&ctl->ticket, 1u);
    __syncthreads ();
    const u64 tile = sh.tile;
    if (tile >= num_tiles) break;

    const u64 a0 = part[2 * tile], b0 = part[2 * tile + 1];
    const u64 a1 = part[2 * tile + 2], b1 = part[2 * tile + 3];
    const u32 na = (u32) (a1 - a0), nb = (u32) (b1 - b0), nt = na + nb;
    if (nt > (u32) CAP) {
      if (tid == 0) atomicOr (&ctl->error, 2u);
      break;
    }

    /* ---- stage both record ranges into LDS, AoS dwords -> SoA (keys 8-byte aligned: an
     * unaligned ds_read_b64 would replay at 64 cycles, Guideline 17) */
    {
      const u32 *__restrict__ srcA = A + 3 * a0;
      for (u32 d = tid; d < 3 * na; d += NT) {
        const u32 w = srcA[d];
        const u32 r = d / 3, f = d - 3 * r;
        const u32 at = (f == 2) ? (2 * CAP + r) : (2 * r + f);
        lds32[at] = w;
      }
      const u32 *__restrict__ srcB = B + 3 * b0;
      for (u32 d = tid; d < 3 * nb; d += NT) {
        const u32 w = srcB[d];
        const u32 r = d / 3 + na, f = d % 3;
        const u32 at = (f == 2) ? (2 * CAP + r) : (2 * r + f);
        lds32[at] = w;
      }
    }
    __syncthreads ();

    /* ---- per-thread merge path inside the tile */
    const u64 *const keyA = sh.keys, *const keyB = sh.keys + na;
    u32 diag = (u32) tid * VT;
    if (diag > nt) diag = nt;
    u32 ai, bi;
    {
      u32 lo = diag > nb ? diag - nb : 0, hi = diag < na ? diag : na;
      while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (keyA[mid] <= keyB[diag - 1 - mid]) lo = mid + 1;
        else hi = mid;
      }
      ai = lo;
      bi = diag - lo;
    }

    /* ---- serial merge of VT items; classification per merged record.  Straight-line: every
     * step does two count reads and one key read at selected (clamped) LDS indices. */
    u64 item_key[VT];
    u32 item_fa[VT], item_fb[VT];
    u32 kinds = 0; /* 2 bits per item */
    {
      constexpr u32 LAST = (u32) CAP - 1;
      bool have_prev = ai > 0;
      u64 prev_a = sh.keys[have_prev ? ai - 1 : 0];
      u64 ka = sh.keys[ai < LAST ? ai : LAST], kb = sh.keys[na + bi < LAST ? na + bi : LAST];
#pragma unroll
      for (int s = 0; s < VT; s++) {
        const bool valid = diag + s < nt;
        const bool a_ok = ai < na, b_ok = bi < nb;
        const bool take_a = a_ok && (!b_ok || ka <= kb);
        const bool both = take_a && b_ok && ka == kb;
        const u32 b_at = na + bi;
        const u32 own_at = take_a ? ai : b_at;
        const u32 c_own = sh.cnts[own_at < LAST ? own_at : LAST];
        const u32 c_b = sh.cnts[b_at < LAST ? b_at : LAST];
        /* a B record whose key equals the A record consumed just before it is that record's
         * partner: the pair was already classified BOTH at the A record */
        const bool partner = have_prev && prev_a == kb;
        u32 kind = take_a ? (both ? KIND_BOTH : KIND_A) : (partner ? KIND_SKIP : KIND_B);
        kind = valid ? kind : KIND_SKIP;
        item_key[s] = take_a ? ka : kb;
        item_fa[s] = take_a ? c_own : 0u;
        item_fb[s] = take_a ? (both ? c_b : 0u) : c_own;
        kinds |= kind << (2 * s);
        prev_a = take_a ? ka : prev_a;
        have_prev = have_prev || take_a;
        ai += take_a ? 1u : 0u;
        bi += take_a ? 0u : 1u;
        const u32 nxt = take_a ? ai : na + bi;
        const u64 nk = sh.keys[nxt < LAST ? nxt : LAST];
        ka = take_a ? nk : ka;
        kb = take_a ? kb : nk;
      }
    }

    /* ---- predicates: per-thread emit masks (4 x 16 bit), counts (4 x 16 bit) and count sums.
     * One pass per requested stream behind a wave-uniform branch keeps register pressure flat. */
    u64 emit_packed = 0, packed = 0;
#pragma unroll 1
    for (int s = 0; s < 4; s++) {
      if (!((p.ops >> s) & 1u)) continue;
      u32 m = 0;
      u64 sum = 0;
      switch (s) {
        case 0: count_stream<0, VT> (kinds, item_fa, item_fb, p, m, sum); break;
        case 1: count_stream<1, VT> (kinds, item_fa, item_fb, p, m, sum); break;
        case 2: count_stream<2, VT> (kinds, item_fa, item_fb, p, m, sum); break;
        default: count_stream<3, VT> (kinds, item_fa, item_fb, p, m, sum); break;
      }
      emit_packed |= (u64) m << (16 * s);
      packed |= (u64) __popc (m) << (16 * s);
      if (s == 0) acc_sum0 += sum;
      else if (s == 1) acc_sum1 += sum;
      else if (s == 2) acc_sum2 += sum;
      else acc_sum3 += sum;
    }

    /* ---- block exclusive scan of the packed counts (all four streams at once; totals <= CAP < 2^16) */
    const u64 incl = wave_inclusive_scan (packed, lane);
    if (lane == WAVE - 1) sh.wave_tot[wid] = incl;
    __syncthreads (); /* also: every thread is done reading the input view of LDS */
    u64 wave_off = 0, tile_tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const u64 t = sh.wave_tot[w];
      if (w < wid) wave_off += t;
      tile_tot += t;
    }
    const u64 excl = wave_off + incl - packed; /* this thread's first slot per stream, packed */

    if (tid == 0) {
#pragma unroll
      for (int s = 0; s < 4; s++) blk_cnt[s] += (tile_tot >> (16 * s)) & 0xffffu;
    }

    if (MODE == MODE_COUNT) {
      /* pass 1 of the two-pass path (and --count_only): leave the tile's counts for the scan */
      if (desc && tid < 4) desc[4 * tile + tid] = (tile_tot >> (16 * tid)) & 0xffffu;
      __syncthreads ();
      continue;
    }

    /* ---- global offsets of this tile */
    if (MODE == MODE_LOOKBACK) {
      if (wid == 0) {
        /* lanes 0..3 own one stream each for publishing; the look-back itself runs per stream
         * with all 64 lanes inspecting 64 predecessors at a time */
        const u64 my_agg = (lane < 4) ? ((tile_tot >> (16 * lane)) & 0xffffu) : 0;
        if (tile == 0) {
          if (lane < 4) {
            desc_store (&desc[lane], DESC_PREFIX | my_agg);
            sh.tile_excl[lane] = 0;
          }
        } else {
          if (lane < 4) desc_store (&desc[4 * tile + lane], DESC_AGG | my_agg);
          u64 excl_s[4] = { 0, 0, 0, 0 };
          bool failed = false;
#pragma unroll
          for (int s = 0; s < 4; s++) {
            if (!((p.ops >> s) & 1u)) continue;
            u64 running = 0;
            long long base = (long long) tile - 1; /* nearest predecessor examined by lane 0 */
            for (;;) {
              const long long idx = base - lane;
              u64 d = DESC_PREFIX; /* before tile 0: prefix 0 */
              u32 spins = 0;
              if (idx >= 0) {
                d = desc_load (&desc[4 * (u64) idx + s]);
                while ((d >> 62) == 0) {
                  if (++spins > SPIN_LIMIT) { failed = true; break; }
                  __builtin_amdgcn_s_sleep (1);
                  d = desc_load (&desc[4 * (u64) idx + s]);
                }
              }
              if (__any (failed)) { failed = true; break; }
              const u64 has_prefix = __ballot ((d >> 62) == 2);
              if (has_prefix) {
                const int first = __ffsll ((long long) has_prefix) - 1; /* nearest tile with a full prefix */
                running += wave_sum (lane <= first ? (d & DESC_VALUE) : 0);
                break;
              }
              running += wave_sum (d & DESC_VALUE);
              base -= WAVE;
            }
            if (failed) break;
            excl_s[s] = running;
          }
          if (failed) {
            if (lane == 0) atomicOr (&ctl->error, 1u);
          }
          if (lane < 4) {
            u64 mine = 0;
#pragma unroll
            for (int s = 0; s < 4; s++) if (lane == s) mine = excl_s[s];
            desc_store (&desc[4 * tile + lane], DESC_PREFIX | ((mine + my_agg) & DESC_VALUE));
            sh.tile_excl[lane] = mine;
          }
        }
      }
    } else { /* MODE_OFFSETS: desc holds the scanned exclusive offsets */
      if (tid < 4) sh.tile_excl[tid] = desc[4 * tile + tid];
    }
    __syncthreads ();

    /* ---- compaction: per stream, scatter kept records into LDS in output order, then store the
     * tile's run with coalesced dword stores at its global offset */
#pragma unroll 1
    for (int s = 0; s < 4; s++) {
      if (!((p.ops >> s) & 1u)) continue;
      const u32 cnt_s = (u32) ((tile_tot >> (16 * s)) & 0xffffu);
      const u32 pos = (u32) ((excl >> (16 * s)) & 0xffffu);
      const u32 m = (u32) ((emit_packed >> (16 * s)) & 0xffffu);
      switch (s) {
        case 0: scatter_stream<0, VT> (kinds, m, pos, item_key, item_fa, item_fb, p, lds32); break;
        case 1: scatter_stream<1, VT> (kinds, m, pos, item_key, item_fa, item_fb, p, lds32); break;
        case 2: scatter_stream<2, VT> (kinds, m, pos, item_key, item_fa, item_fb, p, lds32); break;
        default: scatter_stream<3, VT> (kinds, m, pos, item_key, item_fa, item_fb, p, lds32); break;
      }
      __syncthreads ();
      u32 *__restrict__ dst = outs.rec[s] + 3 * sh.tile_excl[s];
      for (u32 d = tid; d < 3 * cnt_s; d += NT) dst[d] = lds32[d];
      __syncthreads ();
    }
  }

  /* ---- kernel totals: header n_words / total_count (reference :801-802, :909-910) */
  {
    const u64 sums[4] = { acc_sum0, acc_sum1, acc_sum2, acc_sum3 };
#pragma unroll
    for (int s = 0; s < 4; s++) {
      if (!((p.ops >> s) & 1u)) continue;
      const u64 v = wave_sum (sums[s]);
      if (lane == 0 && v) atomicAdd (&ctl->total_count[s], v);
      if (tid == 0 && blk_cnt[s]) atomicAdd (&ctl->n_words[s], blk_cnt[s]);
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

int merge_blocks_per_cu ()
{
  static int cached = 0;
  if (!cached) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_pair_merge<MERGE_NT, MERGE_VT, MODE_LOOKBACK>, MERGE_NT, 0) != hipSuccess || n < 1) n = 1;
    cached = n;
  }
  return cached;
}

hipError_t launch_pair_merge (hipStream_t s, int mode, int grid, const uint32_t *A, uint64_t nA,
                              const uint32_t *B, uint64_t nB, const uint64_t *part, uint64_t num_tiles,
                              const PairParams &p, const PairOutputs &o, unsigned long long *desc,
                              PairControl *ctl)
{
  if (mode == MODE_COUNT)
    hipLaunchKernelGGL ((k_pair_merge<MERGE_NT, MERGE_VT, MODE_COUNT>), dim3 (grid), dim3 (MERGE_NT), 0, s, A, nA, B, nB,
                        (const u64 *) part, num_tiles, p, o, desc, ctl);
  else if (mode == MODE_LOOKBACK)
    hipLaunchKernelGGL ((k_pair_merge<MERGE_NT, MERGE_VT, MODE_LOOKBACK>), dim3 (grid), dim3 (MERGE_NT), 0, s, A, nA, B, nB,
                        (const u64 *) part, num_tiles, p, o, desc, ctl);
  else
    hipLaunchKernelGGL ((k_pair_merge<MERGE_NT, MERGE_VT, MODE_OFFSETS>), dim3 (grid), dim3 (MERGE_NT), 0, s, A, nA, B, nB,
                        (const u64 *) part, num_tiles, p, o, desc, ctl);
  return hipGetLastError ();
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
