/*
 * gt4hip_kway.hip -- N-way union of up to eight sorted lists in ONE pass over HBM.
 *
 * What it restates: union_multi (reference src/glistcompare.c:500-603; hot loop :545-591) and
 * gt4_write_union (src/set-operations.c:40-129): for every distinct key ascending, the count is
 * the sum / maximum / override over the lists that hold the key, kept iff count >= cutoff.
 *
 * Why a second kernel: the pairwise tree of k_pair_merge moves every record log2(N) times through
 * HBM (246 GB for eight 5e8-record lists whose algorithmic traffic is 78 GB).  Here a workgroup
 * owns one TILE of the merged key sequence -- a key range cut out of all the lists at once -- loads
 * the tile's up to eight sorted runs into LDS, merges them pairwise INSIDE LDS (three passes for
 * eight runs, each pass a merge-path split per thread followed by a short serial merge, records
 * ping-ponging between two LDS buffers), combines equal keys, compacts, and writes the tile out
 * once.  HBM sees every input record once and every output record once.
 *
 *   K5 k_kway_sample     every S-th key of every list -> "sample lists" (1/S of the data)
 *   K6 k_kway_partition  tile boundaries: the merged samples' every G-th key, located in every list
 *                        by binary search (all records with a key <= the boundary key go left);
 *      k_kway_check      no tile may exceed the LDS capacity (else the host takes the pairwise tree)
 *   K7 k_kway_merge      the tile merge; MODE_DUPS keeps every record (it is how the sample lists
 *                        themselves are merged, one level up: the recursion ends when a level fits
 *                        one tile), MODE_LOOKBACK / MODE_COUNT combine equal keys and apply the rule.
 *
 * Global output offsets come from the same chained scan as the pair kernel (scanner wavefront +
 * per-row carries, gt4hip_device.h).
 */
#include "gt4hip_device.h"
#include "gt4hip_host.h"

#include <math.h>
#include <string.h>

namespace gt4 {

namespace {

constexpr int KWAY_MAX = 8;
constexpr int KWAY_SAMPLE = 256; /* S: one sample per 256 records (3 KB): the strided gather costs ~4 % of a streaming read */

enum : int { KWAY_COUNT = 0, KWAY_UNION = 1, KWAY_DUPS = 2 };

struct KwayParams {
  const u32 *list[KWAY_MAX];
  u64 n[KWAY_MAX];
  u32 k;
  u32 rule;            /* 1 ADD, 4 MAX, 7 NUMBER */
  u32 cutoff;
  u32 count_override;
  u32 filter;          /* FILTER_RAW: keep every key; FILTER_RESULT: count >= cutoff */
  u32 spin_limit;
  u32 num_tiles;
};

/* ------------------------------------------------------------------ K5 / K6: samples and tile boundaries */

__global__ void k_kway_sample (const u32 *__restrict__ list, u64 n_samples, u32 *__restrict__ out)
{
  const u64 step = (u64) gridDim.x * blockDim.x;
  for (u64 j = (u64) blockIdx.x * blockDim.x + threadIdx.x; j < n_samples; j += step) {
    const u64 src = (j + 1) * KWAY_SAMPLE - 1; /* the last key of every full block of S records */
    out[3 * j] = list[3 * src];
    out[3 * j + 1] = list[3 * src + 1];
    out[3 * j + 2] = 0;
  }
}

/* part[t][i] = first record of list i that belongs to tile t or a later one.  Tile t > 0 starts
 * behind the boundary key x_t = merged_samples[t * G - 1] (the last boundary is the very last
 * sample, so that the final tile holds only the lists' tails behind their last samples):
 * records with key <= x_t belong to earlier tiles (upper bound), equal keys of different lists
 * therefore always meet in one tile. */
__global__ void k_kway_partition (KwayParams p, const u32 *__restrict__ merged, u64 m_total, u32 G, u64 *__restrict__ part)
{
  const u64 id = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  const u64 t = id / KWAY_MAX;
  const u32 i = (u32) (id % KWAY_MAX);
  if (t > p.num_tiles) return;
  u64 v;
  if (i >= p.k || t == 0) {
    v = 0;
  } else if (t == p.num_tiles) {
    v = p.n[i];
  } else {
    const u64 sidx = (t == p.num_tiles - 1) ? m_total - 1 : t * (u64) G - 1;
    const u64 x = load_key (merged, sidx);
    const u32 *__restrict__ L = p.list[i];
    u64 lo = 0, hi = p.n[i];
    while (lo < hi) {
      const u64 mid = (lo + hi) >> 1;
      if (load_key (L, mid) <= x) lo = mid + 1;
      else hi = mid;
    }
    v = lo;
  }
  part[t * KWAY_MAX + i] = v;
}

__global__ void k_kway_check (const u64 *__restrict__ part, u32 num_tiles, u32 cap, u32 *flag)
{
  const u64 t = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= num_tiles) return;
  u64 s = 0;
  bool mono = true;
  for (int i = 0; i < KWAY_MAX; i++) {
    const u64 a = part[t * KWAY_MAX + i], b = part[(t + 1) * KWAY_MAX + i];
    mono &= b >= a;
    s += b - a;
  }
  if (s > cap || !mono) atomicOr (flag, 1u);
}

/* ------------------------------------------------------------------ K7: tile merge in LDS */

template <int NT, int CAP>
struct KwayShared {
  alignas (16) u32 buf[2][3 * CAP + 4 * KWAY_MAX]; /* two record buffers (the runs of the first pass start on 16-byte boundaries) */
  u64 rng[2][KWAY_MAX];
  u32 wave_tot[NT / WAVE];
  u64 excl;
  u32 tick[2];
};

__device__ __forceinline__ u64 lds_key (const u32 *b, u32 dw) { return (u64) b[dw] | ((u64) b[dw + 1] << 32); }

/* P = number of pairwise passes = log2 (run slots): 2 for three or four lists, 3 for five to eight */
template <int NT, int CAP, int MODE, int P>
__global__ __launch_bounds__ (NT, (NT / 256)) void
k_kway_merge (KwayParams p, const u64 *__restrict__ part, u32 *__restrict__ out, u64 *desc, PairControl *ctl)
{
  constexpr int NW = NT / WAVE;
  constexpr int SLOTS = 1 << P;
  typedef KwayShared<NT, CAP> Shared;
  __shared__ Shared sh;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wid = __builtin_amdgcn_readfirstlane (tid / WAVE);
  const u64 n_rows = ((u64) p.num_tiles + WAVE - 1) / WAVE;
  const u32 spin_limit = p.spin_limit ? p.spin_limit : SPIN_LIMIT;
  u32 *const agg = reinterpret_cast<u32 *> (desc);
  u64 *const carry = desc + 2 * n_rows * WAVE;

  u32 role = 0;
  if (MODE == KWAY_UNION) {
    if (tid == 0) sh.tick[0] = atomicAdd (&ctl->role, 1u);
    __syncthreads ();
    role = sh.tick[0];
    __syncthreads ();
    if (role == 0) {
      if (wid == 0) scanner_wave (agg, carry, p.num_tiles, ctl, lane, spin_limit);
      return;
    }
  }
  const u32 n_workers = MODE == KWAY_UNION ? gridDim.x - 1 : gridDim.x;
  const u32 wk = MODE == KWAY_UNION ? role - 1 : blockIdx.x;

  u64 acc_sum = 0; /* per-thread sum of kept counts */
  u64 blk_cnt = 0; /* thread 0: records kept */

  for (u64 tile = wk; tile < p.num_tiles; tile += n_workers) {
    /* ---- the tile's ranges */
    if (tid < KWAY_MAX) {
      sh.rng[0][tid] = part[tile * KWAY_MAX + tid];
      sh.rng[1][tid] = part[(tile + 1) * KWAY_MAX + tid];
    }
    __syncthreads ();
    u32 rlen[SLOTS], roff[SLOTS]; /* wave-uniform: current runs (records, first dword in the current buffer) */
    u32 total = 0, dw = 0;
    u64 out_base = 0;
#pragma unroll
    for (int i = 0; i < SLOTS; i++) {
      const u64 s = i < KWAY_MAX ? uniform64 (sh.rng[0][i]) : 0, e = i < KWAY_MAX ? uniform64 (sh.rng[1][i]) : 0;
      rlen[i] = (u32) i < p.k ? (u32) (e - s) : 0u;
      roff[i] = dw;
      dw += (3 * rlen[i] + 3) & ~3u;
      total += rlen[i];
      out_base += (u32) i < p.k ? s : 0;
    }
    if (total > (u32) CAP) {
      if (tid == 0) atomicOr (&ctl->error, 2u);
      break;
    }
    /* ---- load: run i lies in buffer 0 from dword roff[i] exactly as in HBM (packed 12-byte records) */
    u32 *X = sh.buf[0], *Y = sh.buf[1];
#pragma unroll
    for (int i = 0; i < SLOTS; i++) {
      if (i >= KWAY_MAX || !rlen[i]) continue;
      const u64 s = uniform64 (sh.rng[0][i]);
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc ((void *) (p.list[i] + 3 * s), 0, (int) (12 * rlen[i]), 0x00020000);
      const u32 chunks = (3 * rlen[i] + 3) >> 2;
      for (u32 q = (u32) tid; q < chunks; q += NT) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128 (r, 16 * q, 0, 0);
        *reinterpret_cast<u32x4 *> (X + roff[i] + 4 * q) = v;
      }
    }
    __syncthreads ();

    /* ---- P pairwise passes inside LDS: runs (2m, 2m+1) of buffer X -> run m of buffer Y, which is
     * contiguous (run m starts where the records of runs 0 .. 2m-1 end) */
#pragma unroll
    for (int pass = 0; pass < P; pass++) {
      constexpr int DUMMY = 0;
      (void) DUMMY;
      const int M = SLOTS >> (pass + 1); /* merges of this pass */
      /* positions per thread; every merge rounds its thread count up, hence the M spare threads */
      const u32 VT = total ? (total + (u32) NT - (u32) M - 1u) / ((u32) NT - (u32) M) : 1u;
      u32 thr[SLOTS / 2 + 1], oo[SLOTS / 2 + 1], iters = 0;
      thr[0] = 0;
      oo[0] = 0;
#pragma unroll
      for (int m = 0; m < SLOTS / 2; m++) {
        if (m >= M) continue;
        const u32 lx = rlen[2 * m], ly = rlen[2 * m + 1], s = lx + ly;
        thr[m + 1] = thr[m] + (s + VT - 1) / VT;
        oo[m + 1] = oo[m] + s;
        const u32 mn = lx < ly ? lx : ly;
        const u32 it = mn ? 32u - (u32) __builtin_clz (mn) : 0u;
        iters = it > iters ? it : iters;
      }
      /* this thread's merge and its descriptor (per-lane selects over at most four merges) */
      u32 xo = roff[0], lx = rlen[0], yo = roff[1], ly = rlen[1], ob = 0, t0 = 0;
#pragma unroll
      for (int m = 1; m < SLOTS / 2; m++) {
        if (m >= M) continue;
        const bool g = (u32) tid >= thr[m];
        xo = g ? roff[2 * m] : xo;
        lx = g ? rlen[2 * m] : lx;
        yo = g ? roff[2 * m + 1] : yo;
        ly = g ? rlen[2 * m + 1] : ly;
        ob = g ? oo[m] : ob;
        t0 = g ? thr[m] : t0;
      }
      const u32 s = lx + ly;
      const u32 d0 = ((u32) tid - t0) * VT;
      const bool work = (u32) tid < thr[M] && d0 < s;
      /* merge-path split: a = records of X among the first d0 of merge (X, Y), X first on ties */
      u32 lo = d0 > ly ? d0 - ly : 0u, hi = d0 < lx ? d0 : lx;
      if (!work) lo = hi = 0;
      for (u32 it = 0; it < iters; it++) {
        const bool act = lo < hi;
        const u32 mid = (lo + hi) >> 1;
        const u32 ia = act ? mid : 0u, ib = act ? d0 - 1u - mid : 0u;
        const u64 kx = lds_key (X, xo + 3 * ia), ky = lds_key (X, yo + 3 * ib);
        const bool c = kx <= ky;
        lo = (act && c) ? mid + 1u : lo;
        hi = (act && !c) ? mid : hi;
      }
      u32 pa = xo + 3 * lo, pb = yo + 3 * (d0 - lo);
      const u32 ea = xo + 3 * lx, eb = yo + 3 * ly;
      u32 po = 3 * (ob + d0);
      const u32 pend = 3 * (ob + (d0 + VT < s ? d0 + VT : s));
      if (work) {
        for (u32 j = 0; j < VT; j++) {
          if (po >= pend) break;
          const bool ax = pa < ea, bx = pb < eb;
          const u32 qa = ax ? pa : xo, qb = bx ? pb : yo; /* exhausted side: any address inside the buffer */
          const u64 kx = lds_key (X, qa), ky = lds_key (X, qb);
          const u32 cx = X[qa + 2], cy = X[qb + 2];
          const bool take = ax && (!bx || kx <= ky);
          const u64 key = take ? kx : ky;
          Y[po] = (u32) key;
          Y[po + 1] = (u32) (key >> 32);
          Y[po + 2] = take ? cx : cy;
          pa += take ? 3u : 0u;
          pb += take ? 0u : 3u;
          po += 3;
        }
      }
      __syncthreads ();
      /* the merged runs are the next pass's inputs */
#pragma unroll
      for (int m = 0; m < SLOTS / 2; m++) {
        if (m >= M) continue;
        rlen[m] = oo[m + 1] - oo[m];
        roff[m] = 3 * oo[m];
      }
      u32 *const tmp = X;
      X = Y;
      Y = tmp;
    }
    /* X: the tile's records in key order, equal keys of different lists next to each other */

    if (MODE == KWAY_DUPS) {
      write_out_tile<NT> (out, out_base, total, X, tid);
      __syncthreads (); /* the next tile's load may overwrite this buffer */
      continue;
    }

    /* ---- combine equal keys (union_multi :548-571), keep test (:574), compaction */
    const u32 VTF = (total + (u32) NT - 1u) / (u32) NT;
    const u32 p0 = (u32) tid * VTF, p1 = p0 + VTF < total ? p0 + VTF : total;
    u32 e = 0;
    if (p0 < total) {
      u32 q = p0;
      if (p0 > 0) {
        const u64 pk = lds_key (X, 3 * (p0 - 1));
        while (q < total && lds_key (X, 3 * q) == pk) q++; /* the tail of a key that started in an earlier thread's range */
      }
      while (q < p1) {
        const u64 key = lds_key (X, 3 * q);
        u32 f = p.rule == 7u ? p.count_override : X[3 * q + 2];
        q++;
        while (q < total && lds_key (X, 3 * q) == key) {
          const u32 c = X[3 * q + 2];
          f = p.rule == 1u ? f + c : (p.rule == 4u ? (c > f ? c : f) : p.count_override);
          q++;
        }
        const bool keep = p.filter == FILTER_RAW || f >= p.cutoff;
        if (keep) {
          if (MODE != KWAY_COUNT) {
            Y[3 * (p0 + e)] = (u32) key;
            Y[3 * (p0 + e) + 1] = (u32) (key >> 32);
            Y[3 * (p0 + e) + 2] = f;
          }
          e++;
          acc_sum += f;
        }
      }
    }
    const u32 incl = dpp_inclusive_scan_u32 (e);
    if (lane == WAVE - 1) sh.wave_tot[wid] = incl;
    __syncthreads ();
    u32 wbase = 0, tile_total = 0;
    {
      const u32 wt = lane < NW ? sh.wave_tot[lane] : 0u;
      tile_total = dpp_wave_sum_u32 (wt);
      wbase = dpp_wave_sum_u32 (lane < wid ? wt : 0u);
    }
    if (tid == 0) {
      blk_cnt += tile_total;
      if (MODE == KWAY_UNION) publish_u32 (&agg[tile], AGG_READY | tile_total);
    }
    if (MODE == KWAY_COUNT) {
      __syncthreads ();
      continue;
    }
    /* provisional per-thread positions (buffer Y) -> output order (buffer X, which everybody has
     * finished reading: the barrier above) */
    const u32 slot = wbase + incl - e;
    for (u32 i = 0; i < e; i++) {
      X[3 * (slot + i)] = Y[3 * (p0 + i)];
      X[3 * (slot + i) + 1] = Y[3 * (p0 + i) + 1];
      X[3 * (slot + i) + 2] = Y[3 * (p0 + i) + 2];
    }
    if (wid == 0) {
      const u64 x = resolve_offset (agg, carry, tile, lane, 0, 0, ctl, spin_limit);
      if (lane == 0) sh.excl = x;
    }
    __syncthreads ();
    write_out_tile<NT> (out, uniform64 (sh.excl), tile_total, X, tid);
    __syncthreads ();
  }

  if (MODE != KWAY_DUPS) {
    const u64 v = wave_sum (acc_sum);
    if (lane == 0 && v) atomicAdd (&ctl->total_count[0], v);
    if (tid == 0 && blk_cnt) atomicAdd (&ctl->n_words[0], blk_cnt);
  }
}

constexpr int KWAY_NT = 1024;
constexpr int KWAY_CAP = 6144;

template <int MODE, int P>
hipError_t launch_kway (hipStream_t s, int grid, const KwayParams &p, const u64 *part, u32 *out, u64 *desc, PairControl *ctl)
{
  hipLaunchKernelGGL ((k_kway_merge<KWAY_NT, KWAY_CAP, MODE, P>), dim3 (grid), dim3 (KWAY_NT), 0, s, p, part, out, desc, ctl);
  return hipGetLastError ();
}

hipError_t launch_kway_mode (hipStream_t s, int mode, int grid, const KwayParams &p, const u64 *part, u32 *out, u64 *desc, PairControl *ctl)
{
  const bool p3 = p.k > 4;
  if (mode == KWAY_DUPS) return p3 ? launch_kway<KWAY_DUPS, 3> (s, grid, p, part, out, desc, ctl) : launch_kway<KWAY_DUPS, 2> (s, grid, p, part, out, desc, ctl);
  if (mode == KWAY_COUNT) return p3 ? launch_kway<KWAY_COUNT, 3> (s, grid, p, part, out, desc, ctl) : launch_kway<KWAY_COUNT, 2> (s, grid, p, part, out, desc, ctl);
  return p3 ? launch_kway<KWAY_UNION, 3> (s, grid, p, part, out, desc, ctl) : launch_kway<KWAY_UNION, 2> (s, grid, p, part, out, desc, ctl);
}

}  // namespace

}  // namespace gt4

using namespace gt4;

/* ------------------------------------------------------------------ host orchestration */

namespace {

struct Level {
  KwayParams p;            /* lists of this level (level 0: the caller's; above: sample lists) */
  gt4hip_list *owned[KWAY_MAX];
  u64 total;
};

size_t kway_desc_bytes (u64 tiles)
{
  const u64 rows = (tiles + 63) / 64;
  return (((size_t) rows * 64 * 16 + (size_t) (rows + 1) * 32) + 255) & ~(size_t) 255;
}

int kway_grow (gt4hip_context *ctx, void **p, size_t *have, size_t need)
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

}  // namespace

/* N-way union of 3..8 non-empty lists in one pass.  *used = 0 when the call must take the pairwise
 * tree instead (a tile would not fit LDS: adversarial key distributions; or the single-pass chain
 * gave up on a shared device).  `out`: capacity >= sum of the lists (unless count_only). */
int gt4hip_kway_union (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t k, uint32_t rule, uint32_t cutoff, uint32_t ovr,
                       uint32_t filter, bool count_only, gt4hip_list *out, uint64_t *n_words, uint64_t *total_count, double *device_ms,
                       int *used)
{
  *used = 0;
  if (k < 3 || k > KWAY_MAX) return GT4HIP_OK;
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
      for (int i = 0; i < KWAY_MAX; i++)
        if (lv.owned[i]) gt4hip_list_free (lv.owned[i]);
  };
  HIPCHK (ctx, hipEventRecord (ctx->ev[0], st));
  /* sample levels until one fits a single tile */
  while (levels.back ().total > (u64) KWAY_CAP) {
    const Level &lo = levels.back ();
    Level up;
    memset (&up, 0, sizeof up);
    up.p.k = k;
    for (uint32_t i = 0; i < k && !rc; i++) {
      const u64 m = lo.p.n[i] / KWAY_SAMPLE;
      rc = gt4hip_list_new (ctx, m ? m : 1, lists[0]->word_length, &up.owned[i]);
      if (rc) break;
      up.p.list[i] = (const u32 *) up.owned[i]->dev;
      up.p.n[i] = m;
      up.total += m;
      if (m) {
        u64 g = (m + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL (k_kway_sample, dim3 ((unsigned) g), dim3 (256), 0, st, lo.p.list[i], m, (u32 *) up.owned[i]->dev);
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
  const double margin = 5.0 * KWAY_SAMPLE * sqrt ((double) k / 6.0);
  u32 G = (u32) (((double) KWAY_CAP - margin) / KWAY_SAMPLE);
  if (G < 1) G = 1;
  if (ctx->kway_g > 0) G = (u32) ctx->kway_g;
  for (int l = (int) levels.size () - 1; l >= 0 && !rc; l--) {
    Level &lv = levels[l];
    const u64 m_total = merged ? merged->n_words : 0;
    const u64 tiles = m_total ? m_total / G + 2 : 1;
    if (tiles >= 0xfffffff0ull) {
      rc = gt4hip_fail (ctx, GT4HIP_EINVAL, "lists too long: %llu tiles", (unsigned long long) tiles);
      break;
    }
    lv.p.num_tiles = (u32) tiles;
    lv.p.rule = rule;
    lv.p.cutoff = cutoff;
    lv.p.count_override = ovr;
    lv.p.filter = filter;
    lv.p.spin_limit = ctx->spin_limit;
    if ((rc = kway_grow (ctx, (void **) &ctx->kway_part, &ctx->kway_part_bytes, (size_t) (tiles + 1) * KWAY_MAX * 8))) break;
    {
      const u64 threads = (tiles + 1) * KWAY_MAX;
      hipLaunchKernelGGL (k_kway_partition, dim3 ((unsigned) ((threads + 255) / 256)), dim3 (256), 0, st, lv.p, merged ? (const u32 *) merged->dev : NULL,
                          m_total, G, (u64 *) ctx->kway_part);
      hipMemsetAsync (ctx->scratch, 0, 64, st);
      hipLaunchKernelGGL (k_kway_check, dim3 ((unsigned) ((tiles + 255) / 256)), dim3 (256), 0, st, (const u64 *) ctx->kway_part, (u32) tiles, (u32) KWAY_CAP,
                          (u32 *) ctx->scratch);
      hipError_t e = hipMemcpyAsync (ctx->scratch_host, ctx->scratch, 8, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize (st);
      if (e != hipSuccess) {
        rc = gt4hip_fail (ctx, GT4HIP_EHIP, "k-way partition failed: %s", hipGetErrorString (e));
        break;
      }
      if (ctx->scratch_host[0] & 0xffffffffu) { /* a tile would overflow LDS: the caller takes the pairwise tree */
        if (merged) gt4hip_list_free (merged);
        cleanup ();
        ctx->kway_overflows++;
        return GT4HIP_OK;
      }
    }
    if (merged) {
      gt4hip_list_free (merged);
      merged = NULL;
    }
    const int mode = l > 0 ? KWAY_DUPS : (count_only ? KWAY_COUNT : KWAY_UNION);
    u32 *dst = NULL;
    if (l > 0) {
      if ((rc = gt4hip_list_new (ctx, lv.total ? lv.total : 1, lists[0]->word_length, &merged))) break;
      merged->n_words = lv.total;
      dst = (u32 *) merged->dev;
    } else if (!count_only) {
      dst = (u32 *) out->dev;
    }
    int grid = ctx->n_cus;
    if (ctx->grid_override > 0) grid = (int) ctx->grid_override;
    if (mode == KWAY_UNION) {
      if ((rc = kway_grow (ctx, (void **) &ctx->desc, &ctx->desc_bytes, kway_desc_bytes (tiles)))) break;
      hipMemsetAsync (ctx->desc, 0, kway_desc_bytes (tiles), st);
      if ((u64) grid > tiles + 1) grid = (int) tiles + 1;
    } else if ((u64) grid > tiles) {
      grid = (int) tiles;
    }
    hipMemsetAsync (ctx->ctl, 0, sizeof (PairControl), st);
    hipError_t e = launch_kway_mode (st, mode, grid, lv.p, (const u64 *) ctx->kway_part, dst, (u64 *) ctx->desc, ctx->ctl);
    if (e != hipSuccess) {
      rc = gt4hip_fail (ctx, GT4HIP_EHIP, "k-way merge launch failed: %s", hipGetErrorString (e));
      break;
    }
    if (l == 0) {
      HIPCHK (ctx, hipEventRecord (ctx->ev[3], st));
      e = hipMemcpyAsync (ctx->ctl_host, ctx->ctl, sizeof (PairControl), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize (st);
      if (e != hipSuccess) {
        rc = gt4hip_fail (ctx, GT4HIP_EHIP, "k-way merge failed: %s", hipGetErrorString (e));
        break;
      }
      if (ctx->ctl_host->error) {
        /* a bounded wait gave up (shared device) or a consistency check tripped: the tree redoes the call */
        ctx->single_pass_fallbacks++;
        cleanup ();
        return GT4HIP_OK;
      }
      *n_words = ctx->ctl_host->n_words[0];
      *total_count = ctx->ctl_host->total_count[0];
      float ms = 0;
      if (hipEventElapsedTime (&ms, ctx->ev[0], ctx->ev[3]) == hipSuccess) *device_ms = ms;
      *used = 1;
    }
  }
  if (merged) gt4hip_list_free (merged);
  cleanup ();
  return rc;
}
