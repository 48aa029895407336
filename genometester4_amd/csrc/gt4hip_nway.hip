/*
 * gt4hip_nway.hip -- N-way union of up to eight sorted lists in ONE pass over HBM.
 *
 * What it restates: union_multi (reference src/glistcompare.c:500-603; hot loop :545-591) and
 * gt4_write_union (src/set-operations.c:40-129): for every distinct key ascending, the count is
 * the sum / maximum / override over the lists that hold the key, kept iff count >= cutoff.
 *
 * Why a second kernel: the pairwise tree of k_pair_merge moves every record log2(N) times through
 * HBM (246 GB for eight 5e8-record lists whose algorithmic traffic is 78 GB), and a rank search per
 * record and level is what the pair kernel spends its instructions on.  Here a workgroup owns one
 * TILE of the merged key sequence -- a key range cut out of all the lists at once -- and orders the
 * tile's records WITHOUT any search, at a cost per record that does not depend on the number of lists:
 *
 *   1. every lane holds one record (fetched one tile ahead with 12-byte buffer loads, 64 consecutive
 *      records of ONE list per wave-instruction);
 *   2. a monotone bucket number from the key by interpolation inside the tile's key range (two
 *      buckets per position on average), one LDS atomic per record counts the bucket and hands the
 *      record its arrival number; a workgroup scan turns the counts into bucket starts;
 *   3. the keys are stored grouped by bucket; a record's rank inside its bucket is the number of
 *      smaller keys there -- a handful of INDEPENDENT 8-byte LDS reads (no dependent chain), all
 *      lanes running the same number of steps;
 *   4. position = bucket start + rank = number of smaller keys in the tile.  Equal keys of different
 *      lists get the same position: the key is stored there once and the counts are FOLDED by an LDS
 *      atomic (add: u32 wrap as the reference's unsigned sum; max), a byte marks the position live;
 *   5. positions in order: live ones with count >= cutoff are the tile's output -- ballots and prefix
 *      sums compact them into a staging area, written out during the next tile at the offset the
 *      chained scan of tile totals (gt4hip_device.h) has published by then.
 *
 * The interpolation is only a heuristic for SPEED: a tile whose keys cluster (a bucket with more than
 * NWAY_LIMIT keys) takes a bounded fallback -- the records go back to LDS as sorted runs and every
 * record adds up its lower bounds in all the runs (binary searches) -- and continues at step 4.
 *
 *   K5 k_nway_sample     every S-th key of every list -> "sample lists" (1/S of the data)
 *   K6 k_nway_partition  tile boundaries: the merged samples' every G-th key, located in every list
 *                        by binary search (all records with a key <= the boundary key go left), plus
 *                        the tile's key range and interpolation constants;
 *      k_nway_check      no tile may exceed the LDS capacity (else the host retries with fewer samples
 *                        per tile, down to the number for which it cannot happen)
 *   K7 k_nway_merge      the tile kernel; NWAY_DUPS keeps every record (it is how the sample lists
 *                        themselves are merged, one level up: the recursion ends when a level fits
 *                        one tile), NWAY_UNION / NWAY_COUNT fold equal keys and apply the rule.
 */
#include "gt4hip_device.h"
#include "gt4hip_host.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

namespace gt4 {

namespace {

constexpr int NWAY_MAX = 8;       /* lists per launch */
#ifndef GT4_NWAY_SAMPLE
#define GT4_NWAY_SAMPLE 128
#endif
constexpr int NWAY_SAMPLE = GT4_NWAY_SAMPLE; /* S: one sample per S records */
constexpr int NWAY_PSTRIDE = 10;  /* u64 per tile boundary in the partition table: eight cuts, the tile's smallest possible key, interpolation constants */
constexpr int NWAY_LIMIT = 24;    /* keys per bucket the search-free path handles */

enum : int { NWAY_COUNT = 0, NWAY_UNION = 1, NWAY_DUPS = 2 };

struct NwayParams {
  const u32 *list[NWAY_MAX];
  u64 n[NWAY_MAX];
  u32 k;
  u32 rule;            /* 1 ADD, 4 MAX, 7 NUMBER */
  u32 cutoff;
  u32 count_override;
  u32 filter;          /* FILTER_RAW: keep every key; FILTER_RESULT: count >= cutoff */
  u32 spin_limit;
  u32 num_tiles;
  u32 dynamic;         /* tiles by ticket (ctl->ticket) instead of round-robin */
  u32 force_fallback;  /* tests: every tile takes the search path */
};

/* ------------------------------------------------------------------ K5 / K6: samples and tile boundaries */

__global__ void k_nway_sample (const u32 *__restrict__ list, u64 n_samples, u32 *__restrict__ out)
{
  const u64 step = (u64) gridDim.x * blockDim.x;
  for (u64 j = (u64) blockIdx.x * blockDim.x + threadIdx.x; j < n_samples; j += step) {
    const u64 src = (j + 1) * NWAY_SAMPLE - 1; /* the last key of every full block of S records */
    out[3 * j] = list[3 * src];
    out[3 * j + 1] = list[3 * src + 1];
    out[3 * j + 2] = 0;
  }
}

/* the boundary key in front of tile t (0 < t < num_tiles): merged_samples[t * G - 1]; the last boundary
 * is the very last sample, so that the final tile holds only the lists' tails behind their last samples */
__device__ __forceinline__ u64 nway_boundary_key (const u32 *__restrict__ merged, u64 m_total, u32 G, u32 num_tiles, u64 t)
{
  const u64 sidx = (t == (u64) num_tiles - 1) ? m_total - 1 : t * (u64) G - 1;
  return load_key (merged, sidx);
}

/* part[t][i], i < 8: first record of list i that belongs to tile t or a later one.  Tile t > 0 starts
 * behind the boundary key x_t: records with key <= x_t belong to earlier tiles (upper bound), equal
 * keys of different lists therefore always meet in one tile.
 * part[t][8]: the smallest key tile t can hold; part[t][9]: shift | direct << 8 | multiplier << 32 of
 * its bucket function (see nway_bucket). */
__global__ void k_nway_partition (NwayParams p, const u32 *__restrict__ merged, u64 m_total, u32 G, u32 n_buckets, u64 *__restrict__ part)
{
  const u64 id = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  const u64 t = id / NWAY_PSTRIDE;
  const u32 i = (u32) (id % NWAY_PSTRIDE);
  if (t > p.num_tiles) return;
  u64 v = 0;
  if (i < NWAY_MAX) {
    if (i >= p.k || t == 0) {
      v = 0;
    } else if (t == p.num_tiles) {
      v = p.n[i];
    } else {
      const u64 x = nway_boundary_key (merged, m_total, G, p.num_tiles, t);
      const u32 *__restrict__ L = p.list[i];
      u64 lo = 0, hi = p.n[i];
      while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (load_key (L, mid) <= x) lo = mid + 1;
        else hi = mid;
      }
      v = lo;
    }
  } else if (t < p.num_tiles) {
    /* key range [lo, hi] of the tile: between the boundary keys; the first tile starts at the smallest
     * first key, the last one ends at the largest last key */
    u64 lo, hi;
    if (t == 0) {
      lo = ~0ull;
      for (u32 j = 0; j < p.k; j++)
        if (p.n[j]) {
          const u64 f = load_key (p.list[j], 0);
          lo = f < lo ? f : lo;
        }
    } else {
      lo = nway_boundary_key (merged, m_total, G, p.num_tiles, t) + 1ull;
    }
    if (t + 1 == p.num_tiles) {
      hi = 0;
      for (u32 j = 0; j < p.k; j++)
        if (p.n[j]) {
          const u64 l = load_key (p.list[j], p.n[j] - 1);
          hi = l > hi ? l : hi;
        }
    } else {
      hi = nway_boundary_key (merged, m_total, G, p.num_tiles, t + 1);
    }
    if (i == NWAY_MAX) {
      v = lo;
    } else {
      const u64 D = hi >= lo ? hi - lo : 0ull;
      const u32 bl = D ? 64u - (u32) __builtin_clzll (D) : 0u;
      const u32 sh = bl > 32u ? bl - 32u : 0u;
      const u32 vmax = (u32) (D >> sh);
      if (vmax < n_buckets) v = (u64) sh | (1ull << 8);
      else v = (u64) sh | ((((u64) n_buckets << 32) / ((u64) vmax + 1ull)) << 32);
    }
  }
  part[t * NWAY_PSTRIDE + i] = v;
}

/* a tile fits when its records fit the position space with every run rounded up to whole wavefronts */
__global__ void k_nway_check (const u64 *__restrict__ part, u32 num_tiles, u32 cap, u32 *flag)
{
  const u64 t = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= num_tiles) return;
  u64 slots = 0;
  bool mono = true;
  for (int i = 0; i < NWAY_MAX; i++) {
    const u64 a = part[t * NWAY_PSTRIDE + i], b = part[(t + 1) * NWAY_PSTRIDE + i];
    mono &= b >= a;
    slots += (b - a + WAVE - 1) / WAVE;
  }
  if (slots * WAVE > cap || !mono) atomicOr (flag, 1u);
}

/* ------------------------------------------------------------------ K7: the tile kernel */

typedef u32 u32x3 __attribute__ ((ext_vector_type (3)));

__device__ __forceinline__ u32 dpp_wave_max_u32 (u32 v)
{
  auto mx = [] (u32 a, u32 b) { return a > b ? a : b; };
  v = mx (v, (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x111, 0xf, 0xf, false));
  v = mx (v, (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x112, 0xf, 0xf, false));
  v = mx (v, (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x114, 0xf, 0xf, false));
  v = mx (v, (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x118, 0xf, 0xf, false));
  v = mx (v, (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x142, 0xa, 0xf, false));
  v = mx (v, (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x143, 0xc, 0xf, false));
  return (u32) __builtin_amdgcn_readlane ((int) v, WAVE - 1);
}

template <int NT, int RPT, int NBF, int MODE>
struct NwayShared {
  static constexpr int CAP = NT * RPT;     /* positions = records a tile may hold (runs rounded up to 64) */
  static constexpr int NCH = CAP / WAVE;   /* wave slots */
  static constexpr int NB = NBF * CAP;     /* buckets */
  static constexpr int NW = NT / WAVE;
  /* the tile in key order: key and folded count per position -- or, in the fallback, the records
   * as sorted runs (packed 12 bytes at their positions) */
  union {
    struct {
      u64 skey[CAP];
      u32 scnt[CAP];
    } s;
    u32 raw[3 * CAP];
  };
  u64 g[CAP + NWAY_LIMIT];                 /* keys grouped by bucket (+ all-ones behind the last) */
  alignas (16) u32 cnt[NB / 2 + 4];        /* 16-bit bucket counters, then bucket starts, in pairs (+ the total) */
  alignas (16) u32 live[CAP / 4];          /* one byte per position: a key was stored there */
  alignas (16) u32 stage[MODE == NWAY_COUNT ? 4 : 3 * CAP + 4]; /* the kept records, packed, written out during the NEXT tile */
  u32 wtot[NW], wmax[NW], wkept[NW];
  /* the tile being fetched / processed, two deep: one 64-record wave slot per wave-instruction */
  u64 slot_addr[2][NCH];
  u32 slot_cnt[2][NCH];
  u32 tab_pbase[2][NWAY_MAX];              /* first position of each run */
  u32 tab_len[2][NWAY_MAX];
  u32 tab_n[2], tab_bk[2][2];              /* records; shift | direct << 8, multiplier */
  u64 tab_lo[2], tab_base[2];              /* smallest possible key; NWAY_DUPS: where the tile's output starts */
  u64 rng[3][2 * NWAY_PSTRIDE];            /* partition entries of the next tiles, three deep */
  u32 tile_id[3];
  u64 listbase[NWAY_MAX];
  u64 excl;
  u32 tick;
};

__host__ __device__ constexpr int nway_waves_per_simd (int nt) { return nt >= 1024 ? 4 : (nt >= 512 ? 4 : 4); }

template <int NT, int RPT, int NBF, int MODE>
__global__ __launch_bounds__ (NT, nway_waves_per_simd (NT)) void
k_nway_merge (NwayParams p, const u64 *__restrict__ part, u32 *__restrict__ out, u64 *desc, PairControl *ctl)
{
  typedef NwayShared<NT, RPT, NBF, MODE> Shared;
  constexpr int CAP = Shared::CAP, NW = Shared::NW, NCH = Shared::NCH, NB = Shared::NB;
  constexpr int NWORDS = NB / 2, WPT = NWORDS / NT;
  static_assert (NCH <= WAVE, "one lane per wave slot builds the slot table");
  static_assert (WPT * NT == NWORDS && WPT >= 1, "every thread scans the same number of counter words");
  static_assert (NW <= 16 && NW >= 2, "wave totals are reduced by one DPP row");
  static_assert (CAP <= 65535, "16-bit bucket counters and starts");
  __shared__ Shared sh;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wid = __builtin_amdgcn_readfirstlane (tid / WAVE);
  const u64 n_rows = ((u64) p.num_tiles + WAVE - 1) / WAVE;
  const u32 spin_limit = p.spin_limit ? p.spin_limit : SPIN_LIMIT;
  u32 *const agg = reinterpret_cast<u32 *> (desc);
  u64 *const carry = desc + 2 * n_rows * WAVE;

  u32 role = 0;
  if (MODE == NWAY_UNION) {
    if (tid == 0) sh.tick = atomicAdd (&ctl->role, 1u);
    __syncthreads ();
    role = sh.tick;
    __syncthreads ();
    if (role == 0) {
      if (wid < 8) scanner_part (agg, carry + 4 * (n_rows + 1), carry, p.num_tiles, ctl, lane, spin_limit, (u32) wid, NW < 8 ? (u32) NW : 8u);
      return;
    }
  }
  const u32 n_workers = MODE == NWAY_UNION ? gridDim.x - 1 : gridDim.x;
  const u32 wk = MODE == NWAY_UNION ? role - 1 : blockIdx.x;
  const u32 ntl = p.num_tiles;
  auto deal = [&] (int j) -> u32 {
    if (p.dynamic) {
      const u32 t = atomicAdd (&ctl->ticket, 1u);
      return t < ntl ? t : 0xffffffffu;
    }
    const u64 t = (u64) wk + (u64) j * n_workers;
    return t < (u64) ntl ? (u32) t : 0xffffffffu;
  };

  /* wave 0, one lane per wave slot: where the slot's 64 records lie (ring slot r -> table tb) */
  auto build_table = [&] (int r, int tb) {
    u32 len[NWAY_MAX];
    u32 n = 0;
    u64 base = 0;
#pragma unroll
    for (int q = 0; q < NWAY_MAX; q++) {
      const u64 s = sh.rng[r][q], e = sh.rng[r][NWAY_PSTRIDE + q];
      len[q] = (u32) q < p.k ? uniform32 ((u32) (e - s)) : 0u;
      n += len[q];
      base += (u32) q < p.k ? uniform64 (s) : 0ull;
    }
    u32 acc_w = 0, run = 0, first = 0, rl = 0;
    bool in_any = false;
#pragma unroll
    for (int q = 0; q < NWAY_MAX; q++) {
      const u32 nw = (len[q] + WAVE - 1) / WAVE;
      const bool in = (u32) lane >= acc_w && (u32) lane < acc_w + nw;
      run = in ? (u32) q : run;
      first = in ? ((u32) lane - acc_w) * WAVE : first;
      rl = in ? len[q] : rl;
      in_any |= in;
      if (lane == q) {
        sh.tab_pbase[tb][q] = acc_w * WAVE;
        sh.tab_len[tb][q] = len[q];
      }
      acc_w += nw;
    }
    if (lane < NCH) {
      const u32 c = in_any ? (rl - first < (u32) WAVE ? rl - first : (u32) WAVE) : 0u;
      sh.slot_cnt[tb][lane] = c;
      sh.slot_addr[tb][lane] = sh.listbase[run] + 12ull * (sh.rng[r][run] + first);
    }
    if (lane == 0) {
      sh.tab_n[tb] = acc_w <= (u32) NCH ? n : 0xffffffffu; /* more wave slots than the workgroup has: refused below */
      sh.tab_lo[tb] = sh.rng[r][NWAY_MAX];
      const u64 bk = sh.rng[r][NWAY_MAX + 1];
      sh.tab_bk[tb][0] = (u32) bk;
      sh.tab_bk[tb][1] = (u32) (bk >> 32);
      sh.tab_base[tb] = base;
    }
  };

  /* The tile's records, fetched one tile ahead into registers.  A wavefront fetches 64 consecutive
   * records of ONE run per instruction, so descriptor and addresses are scalar and the range-checked
   * descriptor zero-fills past the run's end: no per-lane bounds. */
  u32x3 pre[RPT];
  auto fetch = [&] (int tb) {
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int chunk = k * NW + wid;
      const u64 addr = uniform64 (sh.slot_addr[tb][chunk]);
      const u32 c = uniform32 (sh.slot_cnt[tb][chunk]);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc ((void *) addr, 0, (int) (12 * c), 0x00020000);
      pre[k] = __builtin_amdgcn_raw_buffer_load_b96 (rs, 12 * lane, 0, 0);
    }
  };
  auto load_ring = [&] (u32 tile, int lane_) -> u64 {
    return __hip_atomic_load (&part[((u64) tile + (u64) (lane_ / NWAY_PSTRIDE)) * NWAY_PSTRIDE + (u64) (lane_ % NWAY_PSTRIDE)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  /* ---- prologue */
  if (tid < NWAY_MAX) {
    u64 lb = (u64) p.list[0];
#pragma unroll
    for (int m = 1; m < NWAY_MAX; m++) lb = tid == m ? (u64) p.list[m] : lb; /* selects: no dynamic indexing of the kernel arguments */
    sh.listbase[tid] = lb;
  }
  u32 tk_next = 0xffffffffu; /* thread 0: the tile of iteration it + 2 */
  if (tid == 0) {
    u32 t3[3];
    for (int q = 0; q < 3; q++) t3[q] = deal (q);
    sh.tile_id[0] = t3[0];
    sh.tile_id[1] = t3[1];
    tk_next = t3[2];
  }
#pragma unroll
  for (int i = 0; i < WPT; i++) sh.cnt[tid * WPT + i] = 0;
  __syncthreads ();
  if (wid == 0 && lane < 2 * NWAY_PSTRIDE) {
    for (int q = 0; q < 2; q++) {
      const u32 t = sh.tile_id[q];
      if (t < ntl) sh.rng[q][lane] = load_ring (t, lane);
    }
  }
  __syncthreads ();
  u32 cur = uniform32 (sh.tile_id[0]);
  if (wid == 0 && cur < ntl) build_table (0, 0);
  __syncthreads ();
  if (cur < ntl) fetch (0);

  u64 acc_sum = 0; /* per-thread sum of kept counts */
  u64 blk_cnt = 0; /* records kept (the same in every thread) */
  u32 pend_tot = 0, pend_tile = 0;
  u64 pend_base = 0;
  bool pend = false;
  int it = 0;
  int r_nxt = 1, r_nn = 2;
#ifdef GT4_PROFILE_PHASES
  u64 ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  u64 t_last;
  asm volatile ("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last) :: "memory");
#endif

  while (cur < ntl) {
    const int tb = it & 1;
    const u32 n = uniform32 (sh.tab_n[tb]);
    if (n > (u32) CAP) {
      if (tid == 0) atomicOr (&ctl->error, 2u);
      break;
    }
    const u64 key_lo = uniform64 (sh.tab_lo[tb]);
    const u32 bk0 = uniform32 (sh.tab_bk[tb][0]), bk_mul = uniform32 (sh.tab_bk[tb][1]);
    const u32 bk_sh = bk0 & 0xffu;
    const bool bk_direct = (bk0 >> 8) & 1u;
    const u64 out_base = uniform64 (sh.tab_base[tb]);

    /* ---- phase 0: the prefetched records: bucket number, arrival number (one LDS atomic) */
    u64 key[RPT];
    u32 cnt[RPT], bkt[RPT], arr[RPT];
    bool valid[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int chunk = k * NW + wid;
      const u32 c = uniform32 (sh.slot_cnt[tb][chunk]);
      valid[k] = (u32) lane < c;
      key[k] = (u64) pre[k].x | ((u64) pre[k].y << 32);
      cnt[k] = pre[k].z;
      const u32 v = (u32) ((key[k] - key_lo) >> bk_sh);
      u32 b = bk_direct ? v : __umulhi (v, bk_mul);
      b = b < (u32) NB ? b : (u32) NB - 1u;
      bkt[k] = b;
      arr[k] = 0;
      if (valid[k]) {
        const u32 s16 = (b & 1u) * 16u;
        const u32 old = atomicAdd (&sh.cnt[b >> 1], 1u << s16);
        arr[k] = (old >> s16) & 0xffffu;
      }
    }
    /* own positions (RPT consecutive ones per thread) of the ordered tile: counts 0, nothing live */
#pragma unroll
    for (int i = 0; i < RPT; i++) sh.s.scnt[tid * RPT + i] = 0;
    if (RPT == 4) sh.live[tid] = 0;
    else
      for (int i = tid; i < CAP / 4; i += NT) sh.live[i] = 0;
    /* housekeeping by wavefront 0: the tile of iteration it + 2 (its partition entries are consumed at
     * the end of this iteration), the slot table of the next tile */
    u32 hk_tile = 0xffffffffu;
    u64 hk = 0;
    u32 nxt = 0xffffffffu;
    if (wid == 0) {
      hk_tile = uniform32 (tk_next);
      if (tid == 0) tk_next = deal (it + 3);
      if (hk_tile < ntl && lane < 2 * NWAY_PSTRIDE) hk = load_ring (hk_tile, lane);
      nxt = uniform32 (sh.tile_id[r_nxt]);
      if (nxt < ntl) build_table (r_nxt, tb ^ 1);
    }
    PHASE_STAMP (0);
    __syncthreads (); /* B1: every record is counted */
    nxt = uniform32 (sh.tile_id[r_nxt]);

    /* ---- scan of the bucket counters: WPT words (two 16-bit counters each) per thread */
    u32 ex[2 * WPT];
    u32 tsum = 0, tmax = 0;
    {
      u32 w[WPT];
#pragma unroll
      for (int i = 0; i < WPT; i++) w[i] = sh.cnt[tid * WPT + i];
#pragma unroll
      for (int i = 0; i < WPT; i++) {
        const u32 a = w[i] & 0xffffu, b = w[i] >> 16;
        ex[2 * i] = tsum;
        tsum += a;
        ex[2 * i + 1] = tsum;
        tsum += b;
        tmax = a > tmax ? a : tmax;
        tmax = b > tmax ? b : tmax;
      }
    }
    const u32 incl = dpp_inclusive_scan_u32 (tsum);
    const u32 wmx = dpp_wave_max_u32 (tmax);
    if (lane == WAVE - 1) {
      sh.wtot[wid] = incl;
      sh.wmax[wid] = wmx;
    }
    PHASE_STAMP (1);
    __syncthreads (); /* B2: wave totals */
    u32 mx;
    {
      const u32 x = lane < NW ? sh.wtot[lane] : 0u;
      const u32 y = lane < NW ? sh.wmax[lane] : 0u;
      const u32 wbase = dpp_wave_sum_u32 (lane < wid ? x : 0u);
      mx = dpp_wave_max_u32 (y);
      const u32 tbase = wbase + incl - tsum;
#pragma unroll
      for (int i = 0; i < WPT; i++) sh.cnt[tid * WPT + i] = (tbase + ex[2 * i]) | ((tbase + ex[2 * i + 1]) << 16);
      if (tid == NT - 1) sh.cnt[NWORDS] = tbase + tsum; /* start of the bucket behind the last = the tile's records */
    }
    /* the chain words of the tile written out below, asked for ahead of the next tile's records (the
     * memory counter retires in order) and looked at behind the rank loop */
    u32 xagg = 0;
    u64 xcarry = 0;
    if (MODE == NWAY_UNION && pend && wid == NW - 1) {
      const u64 prow = pend_tile / WAVE;
      if ((u32) lane < pend_tile % WAVE) xagg = peek_u32 (&agg[prow * WAVE + lane]);
      xcarry = peek_u64 (&carry[prow]);
    }
    if (nxt < ntl) fetch (tb ^ 1);
    PHASE_STAMP (2);
    __syncthreads (); /* B3: bucket starts */

    /* ---- the keys grouped by bucket */
    u32 st[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const u32 w0 = sh.cnt[bkt[k] >> 1];
      const u32 s = (bkt[k] & 1u) ? w0 >> 16 : w0; /* start of the bucket */
      st[k] = valid[k] ? (s & 0xffffu) : 0u;
      if (valid[k]) sh.g[st[k] + arr[k]] = key[k];
    }
    if (tid < NWAY_LIMIT) sh.g[n + tid] = ~0ull;
    PHASE_STAMP (3);
    __syncthreads (); /* B4: keys grouped */
#pragma unroll
    for (int i = 0; i < WPT; i++) sh.cnt[tid * WPT + i] = 0; /* the next tile's counters */

    /* ---- position of every record = number of smaller keys in the tile */
    u32 pos[RPT];
    if (mx <= (u32) NWAY_LIMIT && !p.force_fallback) {
      u32 lt[RPT];
#pragma unroll
      for (int k = 0; k < RPT; k++) lt[k] = 0;
      /* every lane runs the longest bucket's length: behind its own bucket a lane meets larger keys */
#pragma unroll
      for (int j = 0; j < NWAY_LIMIT; j++) {
        if ((u32) j >= mx) break; /* uniform */
#pragma unroll
        for (int k = 0; k < RPT; k++) lt[k] += sh.g[st[k] + j] < key[k] ? 1u : 0u;
      }
#pragma unroll
      for (int k = 0; k < RPT; k++) pos[k] = st[k] + lt[k];
    } else {
      /* clustered keys: the records back to LDS as the sorted runs they came as, and every record adds
       * up its lower bounds in all the runs */
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const u32 q = (u32) (k * NW + wid) * WAVE + (u32) lane;
        if (valid[k]) {
          sh.raw[3 * q] = (u32) key[k];
          sh.raw[3 * q + 1] = (u32) (key[k] >> 32);
          sh.raw[3 * q + 2] = cnt[k];
        }
        pos[k] = 0;
      }
      __syncthreads ();
      for (u32 q = 0; q < p.k; q++) {
        const u32 pb = uniform32 (sh.tab_pbase[tb][q]), len = uniform32 (sh.tab_len[tb][q]);
        const u32 steps = len ? 32u - (u32) __builtin_clz (len) : 0u;
        u32 lo[RPT], hi[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          lo[k] = 0;
          hi[k] = len;
        }
        for (u32 s = 0; s < steps; s++) {
#pragma unroll
          for (int k = 0; k < RPT; k++) {
            const bool act = lo[k] < hi[k];
            const u32 mid = (lo[k] + hi[k]) >> 1;
            const u32 at = 3 * (pb + (act ? mid : 0u));
            const u64 km = (u64) sh.raw[at] | ((u64) sh.raw[at + 1] << 32);
            const bool c = km < key[k];
            lo[k] = (act && c) ? mid + 1u : lo[k];
            hi[k] = (act && !c) ? mid : hi[k];
          }
        }
#pragma unroll
        for (int k = 0; k < RPT; k++) pos[k] += lo[k];
      }
      __syncthreads ();
#pragma unroll
      for (int i = 0; i < RPT; i++) sh.s.scnt[tid * RPT + i] = 0; /* (the runs lay over the counts) */
      __syncthreads ();
    }

    /* ---- the key once per position, the counts folded by LDS atomics */
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      if (!valid[k]) continue;
      u32 q = pos[k];
      if (MODE == NWAY_DUPS) q += atomicAdd (&sh.s.scnt[q], 1u); /* equal sample keys: one position each */
      else if (p.rule == 1u) atomicAdd (&sh.s.scnt[q], cnt[k]);
      else if (p.rule == 4u) atomicMax (&sh.s.scnt[q], cnt[k]);
      sh.s.skey[q] = key[k];
      reinterpret_cast<unsigned char *> (sh.live)[q] = 1;
    }
    if (MODE == NWAY_UNION && pend && wid == NW - 1) {
      const u64 x = resolve_offset (agg, carry, pend_tile, lane, xagg, xcarry, ctl, spin_limit);
      if (lane == 0) sh.excl = 12 * x; /* bytes */
    }
    PHASE_STAMP (4);
    __syncthreads (); /* B5: the tile in key order */

    /* the previous tile leaves its staging area */
    if (MODE != NWAY_COUNT && pend) {
      constexpr int WK = (3 * CAP / 4 + NT - 1) / NT;
      write_out_fixed<NT, WK> (out, MODE == NWAY_UNION ? uniform64 (sh.excl) : 12 * pend_base, pend_tot, sh.stage, tid);
    }

    /* ---- positions in order (RPT consecutive ones per thread): keep test, compaction */
    u64 okey[RPT];
    u32 ocnt[RPT];
    u32 keep_bits = 0;
    {
      const u32 lv = RPT == 4 ? sh.live[tid] : 0u;
#pragma unroll
      for (int i = 0; i < RPT; i++) {
        const u32 q = (u32) tid * RPT + (u32) i;
        const bool on = RPT == 4 ? ((lv >> (8 * i)) & 0xffu) != 0 : reinterpret_cast<const unsigned char *> (sh.live)[q] != 0;
        okey[i] = sh.s.skey[q];
        u32 f = sh.s.scnt[q];
        if (MODE == NWAY_DUPS) f = 0;
        else if (p.rule == 7u) f = p.count_override;
        ocnt[i] = f;
        const bool keep = on && (MODE == NWAY_DUPS || p.filter == FILTER_RAW || f >= p.cutoff);
        keep_bits |= keep ? 1u << i : 0u;
        acc_sum += keep ? f : 0u;
      }
    }
    const u32 kc = (u32) __builtin_popcount (keep_bits);
    const u32 kincl = dpp_inclusive_scan_u32 (kc);
    if (lane == WAVE - 1) sh.wkept[wid] = kincl;
    PHASE_STAMP (5);
    __syncthreads (); /* B6: kept per wavefront; the staging area is free */
    u32 tile_total;
    {
      const u32 x = lane < NW ? sh.wkept[lane] : 0u;
      const u32 incl2 = dpp_inclusive_scan_u32 (x);
      tile_total = (u32) __builtin_amdgcn_readlane ((int) incl2, WAVE - 1);
      const u32 wbase = dpp_wave_sum_u32 (lane < wid ? x : 0u);
      blk_cnt += tile_total;
      if (MODE == NWAY_UNION && wid == 0) {
        if (lane == 0) publish_u32 (&agg[cur], AGG_READY | tile_total);
      }
      if (MODE != NWAY_COUNT) {
        u32 slot = wbase + kincl - kc;
#pragma unroll
        for (int i = 0; i < RPT; i++) {
          if ((keep_bits >> i) & 1u) {
            sh.stage[3 * slot] = (u32) okey[i];
            sh.stage[3 * slot + 1] = (u32) (okey[i] >> 32);
            sh.stage[3 * slot + 2] = ocnt[i];
            slot++;
          }
        }
      }
    }
    pend = MODE != NWAY_COUNT;
    pend_tot = tile_total;
    pend_tile = cur;
    pend_base = out_base;
    if (wid == 0) {
      if (hk_tile < ntl && lane < 2 * NWAY_PSTRIDE) sh.rng[r_nn][lane] = hk;
      if (lane == 0) sh.tile_id[r_nn] = hk_tile;
    }
    PHASE_STAMP (6);
    cur = nxt;
    it++;
    {
      const int r_cur = r_nxt;
      r_nxt = r_nn;
      r_nn = r_cur == 0 ? 2 : r_cur - 1;
    }
  }
#ifdef GT4_PROFILE_PHASES
  if (tid == GT4_STAMP_TID)
    for (int i = 0; i < 8; i++) atomicAdd (&ctl->phase_cycles[i], ph[i]);
#endif
  /* drain: the last tile is still staged */
  if (MODE != NWAY_COUNT && pend) {
    __syncthreads ();
    if (MODE == NWAY_UNION && wid == 0) {
      const u64 x = resolve_offset (agg, carry, pend_tile, lane, 0, 0, ctl, spin_limit);
      if (lane == 0) sh.excl = x;
    }
    __syncthreads ();
    write_out_tile<NT> (out, MODE == NWAY_UNION ? uniform64 (sh.excl) : pend_base, pend_tot, sh.stage, tid);
  }
  if (MODE != NWAY_DUPS) {
    const u64 v = wave_sum (acc_sum);
    if (lane == 0 && v) atomicAdd (&ctl->total_count[0], v);
    if (tid == 0 && blk_cnt) atomicAdd (&ctl->n_words[0], blk_cnt);
  }
}

#ifndef GT4_NWAY_NT
#define GT4_NWAY_NT 1024
#endif
#ifndef GT4_NWAY_RPT
#define GT4_NWAY_RPT 4
#endif
#ifndef GT4_NWAY_NBF
#define GT4_NWAY_NBF 2
#endif
constexpr int NWAY_NT = GT4_NWAY_NT;
constexpr int NWAY_RPT = GT4_NWAY_RPT;
constexpr int NWAY_NBF = GT4_NWAY_NBF;
constexpr int NWAY_CAP = NWAY_NT * NWAY_RPT;

template <int MODE>
hipError_t launch_nway (hipStream_t s, int grid, const NwayParams &p, const u64 *part, u32 *out, u64 *desc, PairControl *ctl)
{
  hipLaunchKernelGGL ((k_nway_merge<NWAY_NT, NWAY_RPT, NWAY_NBF, MODE>), dim3 (grid), dim3 (NWAY_NT), 0, s, p, part, out, desc, ctl);
  return hipGetLastError ();
}

hipError_t launch_nway_mode (hipStream_t s, int mode, int grid, const NwayParams &p, const u64 *part, u32 *out, u64 *desc, PairControl *ctl)
{
  if (mode == NWAY_DUPS) return launch_nway<NWAY_DUPS> (s, grid, p, part, out, desc, ctl);
  if (mode == NWAY_COUNT) return launch_nway<NWAY_COUNT> (s, grid, p, part, out, desc, ctl);
  return launch_nway<NWAY_UNION> (s, grid, p, part, out, desc, ctl);
}

int nway_blocks_per_cu (int mode)
{
  static int cache[3] = { 0, 0, 0 };
  if (!cache[mode]) {
    int n = 0;
    hipError_t e;
    if (mode == NWAY_DUPS) e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_nway_merge<NWAY_NT, NWAY_RPT, NWAY_NBF, NWAY_DUPS>, NWAY_NT, 0);
    else if (mode == NWAY_COUNT) e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_nway_merge<NWAY_NT, NWAY_RPT, NWAY_NBF, NWAY_COUNT>, NWAY_NT, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_nway_merge<NWAY_NT, NWAY_RPT, NWAY_NBF, NWAY_UNION>, NWAY_NT, 0);
    if (e != hipSuccess || n < 1) n = 1;
    const int by_regs = nway_waves_per_simd (NWAY_NT) * 4 / (NWAY_NT / 64);
    if (by_regs >= 1 && n > by_regs) n = by_regs;
    cache[mode] = n;
  }
  return cache[mode];
}

}  // namespace

}  // namespace gt4

using namespace gt4;

/* ------------------------------------------------------------------ host orchestration */

namespace {

struct Level {
  NwayParams p;            /* lists of this level (level 0: the caller's; above: sample lists) */
  gt4hip_list *owned[NWAY_MAX];
  u64 total;
};

size_t nway_desc_bytes (u64 tiles)
{
  const u64 rows = (tiles + 63) / 64;
  return (((size_t) rows * 64 * 16 + (size_t) (rows + 1) * 32 + (size_t) rows * 32) + 255) & ~(size_t) 255; /* agg, carry, rowsum */
}

int nway_grow (gt4hip_context *ctx, void **p, size_t *have, size_t need)
{
  if (*have >= need) return GT4HIP_OK;
  if (*p) {
    HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
    HIPCHK (ctx, hipFree (*p));
    *p = NULL;
    *have = 0;
  }
  need += need / 8;
  if (gt4hip_dev_alloc (ctx, p, need) != hipSuccess) return gt4hip_fail (ctx, GT4HIP_ENOMEM, "workspace hipMalloc of %zu bytes failed", need);
  *have = need;
  return GT4HIP_OK;
}

/* samples per tile: a tile between two boundary keys G samples apart holds at most G + k - 1 samples
 * (ties at the boundaries), each list at most (its samples + 1) * S - 1 records, every run rounded up
 * to whole wavefronts.  `sure`: the G for which no tile can overflow; the first try takes the expected
 * tile (G * S records) plus five standard deviations of the lists' offsets against their sample grids. */
void nway_samples_per_tile (u32 k, u32 *first_try, u32 *sure)
{
  const double cap = (double) NWAY_CAP - 32.0 * k; /* half a wavefront of padding per run, on average */
  const double margin = 5.0 * NWAY_SAMPLE * sqrt ((double) k / 6.0);
  long g1 = (long) ((cap - margin) / NWAY_SAMPLE);
  long g0 = ((long) NWAY_CAP - 64L * k) / NWAY_SAMPLE - (2L * k - 1);
  if (g0 < 1) g0 = 1;
  if (g1 < g0) g1 = g0;
  *first_try = (u32) g1;
  *sure = (u32) g0;
}

}  // namespace

/* N-way union of 3..8 non-empty lists in one pass.  *used = 0 when the call must take the pairwise
 * tree instead (the single-pass chain gave up on a shared device).  `out`: capacity >= sum of the
 * lists (unless count_only). */
int gt4hip_nway_union (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t k, uint32_t rule, uint32_t cutoff, uint32_t ovr,
                       uint32_t filter, bool count_only, gt4hip_list *out, uint64_t *n_words, uint64_t *total_count, double *device_ms,
                       int *used)
{
  *used = 0;
  if (k < 2 || k > NWAY_MAX) return GT4HIP_OK;
  hipStream_t st = ctx->stream;
  std::vector<Level> levels;
  Level l0;
  memset (&l0, 0, sizeof l0);
  l0.p.k = k;
  for (uint32_t i = 0; i < k; i++) {
    l0.p.list[i] = (const u32 *) lists[i]->dev;
    l0.p.n[i] = lists[i]->n_words;
    l0.total += lists[i]->n_words;
  }
  levels.push_back (l0);
  int rc = GT4HIP_OK;
  auto cleanup = [&] () {
    for (Level &lv : levels)
      for (int i = 0; i < NWAY_MAX; i++)
        if (lv.owned[i]) gt4hip_list_free (lv.owned[i]);
  };
  HIPCHK (ctx, hipEventRecord (ctx->ev[0], st));
  /* sample levels until one fits a single tile */
  const u64 one_tile = (u64) NWAY_CAP - 64ull * k;
  while (levels.back ().total > one_tile) {
    const Level &lo = levels.back ();
    Level up;
    memset (&up, 0, sizeof up);
    up.p.k = k;
    for (uint32_t i = 0; i < k && !rc; i++) {
      const u64 m = lo.p.n[i] / NWAY_SAMPLE;
      rc = gt4hip_list_new (ctx, m ? m : 1, lists[0]->word_length, &up.owned[i]);
      if (rc) break;
      up.p.list[i] = (const u32 *) up.owned[i]->dev;
      up.p.n[i] = m;
      up.total += m;
      if (m) {
        u64 g = (m + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL (k_nway_sample, dim3 ((unsigned) g), dim3 (256), 0, st, lo.p.list[i], m, (u32 *) up.owned[i]->dev);
      }
    }
    levels.push_back (up);
    if (rc) {
      cleanup ();
      return rc;
    }
  }
  /* top-down: the merged samples of level l+1 cut level l into tiles */
  gt4hip_list *merged = NULL; /* merged sample records of the level above */
  u32 g_try, g_sure;
  nway_samples_per_tile (k, &g_try, &g_sure);
  if (ctx->kway_g > 0) g_try = (u32) ctx->kway_g;
  for (int l = (int) levels.size () - 1; l >= 0 && !rc; l--) {
    Level &lv = levels[l];
    const u64 m_total = merged ? merged->n_words : 0;
    u32 G = g_try;
    u64 tiles = 1;
    for (;;) {
      tiles = m_total ? m_total / G + 2 : 1;
      if (tiles >= 0xfffffff0ull) {
        rc = gt4hip_fail (ctx, GT4HIP_EINVAL, "lists too long: %llu tiles", (unsigned long long) tiles);
        break;
      }
      lv.p.num_tiles = (u32) tiles;
      if ((rc = nway_grow (ctx, (void **) &ctx->kway_part, &ctx->kway_part_bytes, (size_t) (tiles + 1) * NWAY_PSTRIDE * 8))) break;
      const u64 threads = (tiles + 1) * NWAY_PSTRIDE;
      hipLaunchKernelGGL (k_nway_partition, dim3 ((unsigned) ((threads + 255) / 256)), dim3 (256), 0, st, lv.p, merged ? (const u32 *) merged->dev : NULL,
                          m_total, G, (u32) (NWAY_NBF * NWAY_CAP), (u64 *) ctx->kway_part);
      hipMemsetAsync (ctx->scratch, 0, 64, st);
      hipLaunchKernelGGL (k_nway_check, dim3 ((unsigned) ((tiles + 255) / 256)), dim3 (256), 0, st, (const u64 *) ctx->kway_part, (u32) tiles, (u32) NWAY_CAP,
                          (u32 *) ctx->scratch);
      hipError_t e = hipMemcpyAsync (ctx->scratch_host, ctx->scratch, 8, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize (st);
      if (e != hipSuccess) {
        rc = gt4hip_fail (ctx, GT4HIP_EHIP, "N-way partition failed: %s", hipGetErrorString (e));
        break;
      }
      if (!(ctx->scratch_host[0] & 0xffffffffu)) break;
      /* a tile would overflow LDS: fewer samples per tile, down to the number that cannot overflow */
      ctx->kway_overflows++;
      if (G <= g_sure) {
        rc = gt4hip_fail (ctx, GT4HIP_EINTERNAL, "N-way partition: a tile exceeds the capacity at %u samples per tile", G);
        break;
      }
      const u32 g2 = G - (G + 7) / 8;
      G = g2 > g_sure ? g2 : g_sure;
    }
    if (rc) break;
    if (merged) {
      gt4hip_list_free (merged);
      merged = NULL;
    }
    lv.p.rule = rule;
    lv.p.cutoff = cutoff;
    lv.p.count_override = ovr;
    lv.p.filter = filter;
    lv.p.spin_limit = ctx->spin_limit;
    lv.p.force_fallback = ctx->kway_vt == 99 ? 1u : 0u; /* option "kway_vt" = 99: every tile takes the search path (tests) */
    const int mode = l > 0 ? NWAY_DUPS : (count_only ? NWAY_COUNT : NWAY_UNION);
    lv.p.dynamic = ctx->dynamic > 0 ? 1u : (ctx->dynamic < 0 ? 0u : (mode == NWAY_UNION ? 1u : 0u));
    u32 *dst = NULL;
    if (l > 0) {
      if ((rc = gt4hip_list_new (ctx, lv.total ? lv.total : 1, lists[0]->word_length, &merged))) break;
      merged->n_words = lv.total;
      dst = (u32 *) merged->dev;
    } else if (!count_only) {
      dst = (u32 *) out->dev;
    }
    int grid = ctx->n_cus * nway_blocks_per_cu (mode);
    if (ctx->grid_override > 0) grid = (int) ctx->grid_override;
    if (mode == NWAY_UNION) {
      if ((rc = nway_grow (ctx, (void **) &ctx->desc, &ctx->desc_bytes, nway_desc_bytes (tiles)))) break;
      hipMemsetAsync (ctx->desc, 0, nway_desc_bytes (tiles), st);
      if ((u64) grid > tiles + 1) grid = (int) tiles + 1;
    } else if ((u64) grid > tiles) {
      grid = (int) tiles;
    }
    hipMemsetAsync (ctx->ctl, 0, sizeof (PairControl), st);
    if (l == 0) hipEventRecord (ctx->ev[1], st);
    hipError_t e = launch_nway_mode (st, mode, grid, lv.p, (const u64 *) ctx->kway_part, dst, (u64 *) ctx->desc, ctx->ctl);
    if (e != hipSuccess) {
      rc = gt4hip_fail (ctx, GT4HIP_EHIP, "N-way merge launch failed: %s", hipGetErrorString (e));
      break;
    }
    if (l == 0) hipEventRecord (ctx->ev[2], st);
    /* every level reads its control block back: a refused tile or a wait that gave up must not go unseen */
    if (l == 0) HIPCHK (ctx, hipEventRecord (ctx->ev[3], st));
    e = hipMemcpyAsync (ctx->ctl_host, ctx->ctl, sizeof (PairControl), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize (st);
    if (e != hipSuccess) {
      rc = gt4hip_fail (ctx, GT4HIP_EHIP, "N-way merge failed: %s", hipGetErrorString (e));
      break;
    }
    if (ctx->ctl_host->error) {
      const unsigned flags = ctx->ctl_host->error;
      if (merged) gt4hip_list_free (merged);
      cleanup ();
      if (flags & 2u) return gt4hip_fail (ctx, GT4HIP_EINTERNAL, "N-way merge kernel reported error flags 0x%x", flags);
      /* a bounded wait gave up (shared device): the tree redoes the call */
      ctx->single_pass_fallbacks++;
      return GT4HIP_OK;
    }
    if (l == 0) {
#ifdef GT4_PROFILE_PHASES
      {
        static const char *names[8] = { "wait+bucket+atomic", "B1+scan", "B2+starts+fetch", "B3+group", "B4+rank+fold+resolve", "B5+writeout+order", "B6+stage", "-" };
        unsigned long long tot = 0;
        for (int i = 0; i < 8; i++) tot += ctx->ctl_host->phase_cycles[i];
        fprintf (stderr, "[nway phases] tiles %llu:", (unsigned long long) tiles);
        for (int i = 0; i < 8; i++) fprintf (stderr, " %s %.1f%%", names[i], tot ? 100.0 * ctx->ctl_host->phase_cycles[i] / tot : 0.0);
        fprintf (stderr, " | avg cycles/tile %.0f\n", tiles ? (double) tot / tiles : 0.0);
      }
#endif
      *n_words = ctx->ctl_host->n_words[0];
      *total_count = ctx->ctl_host->total_count[0];
      float ms = 0;
      if (hipEventElapsedTime (&ms, ctx->ev[0], ctx->ev[3]) == hipSuccess) *device_ms = ms;
      if (hipEventElapsedTime (&ms, ctx->ev[1], ctx->ev[2]) == hipSuccess) ctx->nway_kernel_ms = ms;
      ctx->nway_tiles = tiles;
      *used = 1;
    }
  }
  if (merged) gt4hip_list_free (merged);
  cleanup ();
  return rc;
}
