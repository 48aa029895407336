/* gt4hip_nway_body.h -- the N-way tile kernel, its partition kernels and the host orchestration of one call, compiled
 * TWICE by gt4hip_nway.hip: GT4_KM = 8 (up to eight lists per launch: 64-record slots, samples every 128 records -- the
 * geometry of rounds 3 and 4, unchanged) and GT4_KM = 32 (nine to thirty-two lists in ONE pass, round 5: 32-record
 * half-slots, samples every 64 records -- glistmaker's collation width, reference src/glistmaker.c:787-835).
 * No include guard on purpose.  See gt4hip_nway.hip for what the code restates. */
namespace gt4 {

namespace {

namespace GT4_KM_NS {

constexpr int NWAY_MAX = GT4_KM;  /* lists per launch */
#ifndef GT4_NWAY_SAMPLE
#define GT4_NWAY_SAMPLE 128
#endif
/* The geometry of many lists (GT4_KM = 32).  With runs rounded up to 64-record slots a tile of k runs loses half a
 * slot per run, and 5 sigma = 5 S sqrt (k / 6) records to the lists' offsets against their sample grids: thirty-two
 * lists at S = 128 would leave 1594 of 4096 positions to records (39 %; 32-record half-slots at S = 64: 69 % on paper,
 * 58 % measured -- lists on a regular grid all round up at once -- and 63 ms against 67 for levels of eight-way merges).
 * So the many-list kernel does not round at all: the runs lie END TO END in the tile's position space (position p of
 * the tile is record p - P_r of run r), a lane finds its run by a popcount over a mask of run starts (one 64-bit LDS
 * read per record slot) and loads through a per-lane address; S = 64 then leaves 3357 of 4096 positions (82 %). */
constexpr int NWAY_SAMPLE = GT4_KM == 8 ? GT4_NWAY_SAMPLE : 64; /* S: one sample per S records */
constexpr int NWAY_HS = GT4_KM == 8 ? WAVE : 1;                 /* a run is rounded up to a multiple of this many positions */
constexpr int NWAY_PSTRIDE = NWAY_MAX + 2;  /* u64 per tile boundary in the partition table: the lists' cuts, the tile's smallest possible key, interpolation constants */
#ifndef GT4_NWAY_MARGIN
#define GT4_NWAY_MARGIN 5.0 /* standard deviations of a tile's size kept free at the first try.  2.5 (27 samples per tile instead of 24, 7 % of the tiles cut in two) measured 36.4 ms against 30.1: fuller tiles give the service wavefront records of its own */
#endif
#ifndef GT4_NWAY_LIMIT
#define GT4_NWAY_LIMIT 48
#endif
#ifndef GT4_NWAY_TRY0
#define GT4_NWAY_TRY0 32
#endif
constexpr int NWAY_LIMIT = GT4_NWAY_LIMIT;    /* keys per bucket the bucket walks handle */
constexpr int NWAY_TRY0 = GT4_NWAY_TRY0;     /* ... that the interpolation's buckets may hold before the tile is bucketed by a pivot run instead */

enum : int { NWAY_COUNT = 0, NWAY_UNION = 1, NWAY_DUPS = 2, NWAY_TABLE = 3, NWAY_PROBE = 4 };
__host__ __device__ constexpr bool nway_staged (int mode) { return mode == NWAY_UNION || mode == NWAY_DUPS; } /* kept records leave through the staging area */
/* LEAD (NWAY_UNION, NWAY_COUNT): no ordered copy of the tile.  The first record to set its position's bit in a bitmap
 * is the position's LEADER; the counts are folded by LDS atomics as before; behind the barrier the leader reads
 * the folded count, applies the cutoff (a leader that is not kept clears its bit again) and, behind one more barrier,
 * finds its output slot as the number of bits below its own -- a popcount prefix every wavefront works out for itself.
 * No key array, no live bytes, no pass over the positions (a quarter of them empty): the keys' 34 KB go to the grouped
 * keys (which the pivot keys of a clustered tile then share: refilled behind the searches). */
#ifndef GT4_NWAY_LEAD
#define GT4_NWAY_LEAD 1
#endif
#ifndef GT4_NWAY_LEAD_BITS
#define GT4_NWAY_LEAD_BITS 16
#endif
__host__ __device__ constexpr bool nway_lead (int mode) { return GT4_NWAY_LEAD && (mode == NWAY_UNION || mode == NWAY_COUNT); }

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
  u32 force_fallback;  /* tests: 1 every tile takes the search path, 2 every tile is bucketed by its pivot run */
  u32 scan_group;      /* the scanner workgroup as summers + chainer (launches with very many rows) */
  /* NWAY_TABLE (the count table of glistquery's multi-list dump, src/set-operations.c:131-183): a tile's j-th
   * distinct key is row (records in front of the tile) + j of the ragged table; list i's count of the key goes to
   * table_counts[row * table_cols + table_col[i]]; every tile's number of distinct keys -> tile_totals[tile]
   * (the index gt4hip_table_download gathers by).
   * NWAY_PROBE (the table restricted to the keys of list 0: gt4_is_union, search_lists_multi; src/set-operations.c:
   * 185-228, src/glistquery.c:776-812): row r is record r of list 0 -- no counting launch, no ordered pass: list 0's
   * records leave their index at their position, every record of the same key finds it there.  rule NUMBER:
   * count_override instead of the count (membership). */
  u32 *tile_totals;
  u64 *table_keys;
  u32 *table_counts;
  u32 table_cols;
  u32 table_col[NWAY_MAX];
};

/* ------------------------------------------------------------------ K5 / K6: samples and tile boundaries */

/* every S-th key of every list of `lo` (the last key of every full block of S records) -> the lists of `up` */
__global__ void k_nway_sample (NwayParams lo, NwayParams up)
{
  u64 total = 0;
  for (u32 i = 0; i < up.k; i++) total += up.n[i];
  /* four samples per thread and round: every one of them is a scattered 8-byte read (a memory round trip each), asked
   * for together (one per thread and round took 1.0 ms for the 3.1e7 samples of eight 5e8-record lists) */
  constexpr int U = 4;
  const u64 step = (u64) gridDim.x * blockDim.x;
  for (u64 g0 = (u64) blockIdx.x * blockDim.x + threadIdx.x; g0 < total; g0 += U * step) {
    u64 key[U], j[U];
    u32 i[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const u64 g = g0 + (u64) u * step;
      j[u] = g < total ? g : 0;
      i[u] = 0;
      if (g < total)
        while (j[u] >= up.n[i[u]]) {
          j[u] -= up.n[i[u]];
          i[u]++;
        }
      key[u] = g < total ? load_key (lo.list[i[u]], (j[u] + 1) * NWAY_SAMPLE - 1) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (g0 + (u64) u * step >= total) continue;
      u32 *__restrict__ out = const_cast<u32 *> (up.list[i[u]]);
      out[3 * j[u]] = (u32) key[u];
      out[3 * j[u] + 1] = (u32) (key[u] >> 32);
      out[3 * j[u] + 2] = 0;
    }
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
/* Two passes, as the pair kernel's partition: pass 0 searches every NWAY_COARSE-th boundary in the whole
 * lists, pass 1 the others between their coarse neighbours (the cuts are monotone in the boundary
 * key): half the dependent reads, and neighbouring threads probe the same few cache lines. */
__device__ __forceinline__ u64 nway_bucket_consts (u64 lo, u64 hi, u32 n_buckets);

constexpr u64 NWAY_COARSE = 64;

__global__ void k_nway_partition (NwayParams p, const u32 *__restrict__ merged, u64 m_total, u32 G, u32 n_buckets, u64 *__restrict__ part, int pass)
{
  const u64 id = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  const u64 t = id / NWAY_PSTRIDE;
  const u32 i = (u32) (id % NWAY_PSTRIDE);
  if (t > p.num_tiles) return;
  const bool coarse = t % NWAY_COARSE == 0 || t == p.num_tiles;
  if (i < NWAY_MAX ? (pass == 0) != coarse : pass != 0) return; /* (the tiles' key ranges need no search: pass 0) */
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
      if (!coarse) {
        const u64 t0 = t - t % NWAY_COARSE, t1 = t0 + NWAY_COARSE < (u64) p.num_tiles ? t0 + NWAY_COARSE : (u64) p.num_tiles;
        lo = part[t0 * NWAY_PSTRIDE + i];
        hi = part[t1 * NWAY_PSTRIDE + i];
      }
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
      const bool direct = vmax < n_buckets;
      const u32 mul = direct ? 0u : (u32) (((u64) n_buckets << 32) / ((u64) vmax + 1ull));
      v = (u64) sh | (direct ? 1ull << 8 : (u64) mul << 32);
      /* Will the interpolation work?  The tile's own samples tell: G keys spread over thousands of
       * buckets share hardly any when the keys are spread evenly; stretches of adjacent keys between
       * wide gaps put most of them into a few.  Such a tile is bucketed by its pivot run at once. */
      if (merged && t + 1 < p.num_tiles) {
        const u64 first = t * (u64) G, last = (t + 1) * (u64) G < m_total ? (t + 1) * (u64) G : m_total;
        u32 prev = 0xffffffffu, same = 0, cnt = 0;
        u64 prev_key = 0;
        bool have_prev = false;
        for (u64 j = first; j < last; j++) {
          const u64 x = load_key (merged, j);
          if (x < lo || x > hi || (have_prev && x == prev_key)) continue; /* (equal keys of different lists share a bucket by right) */
          prev_key = x;
          have_prev = true;
          const u32 vv = (u32) ((x - lo) >> sh);
          const u32 b = direct ? vv : __umulhi (vv, mul);
          same += b == prev ? 1u : 0u;
          prev = b;
          cnt++;
        }
        if (cnt >= 8 && 2 * same > cnt) v |= 1ull << 9;
      }
    }
  }
  part[t * NWAY_PSTRIDE + i] = v;
}

/* ---- The same table from SAMPLE COUNTS (every level but the topmost).  The merged samples carry the
 * list they came from (NWAY_DUPS stores it in the count word), so the number c of list i's samples in
 * front of a boundary is a prefix count -- and the boundary's cut in list i lies in the S records
 * behind sample c (or, when a sample EQUAL to the boundary key was merged behind it, in the next S):
 * seven probes inside one 1.5 KB stretch instead of a binary search over the whole bracket.
 *   k_nway_sample_counts   per bracket of 64 tiles: samples of every list
 *   k_nway_bracket_bases   exclusive prefix over the brackets (one wavefront per list)
 *   k_nway_partition_rows  one wavefront per bracket, one lane per tile: counts of the tile's own samples
 *                          (and whether they are clustered: see k_nway_partition), prefix over the lanes,
 *                          eight short searches, the tile's key range and bucket constants */
constexpr u32 NWAY_BRACKET = 64;

#if GT4_KM > 8
/* (more than eight lists: the counts of a bracket's samples per list by LDS atomics; the prefix kernel's wavefronts
 * take several lists each; the rows kernel goes through the lists in groups of eight) */
__global__ __launch_bounds__ (256) void k_nway_sample_counts (const u32 *__restrict__ merged, u64 m_total, u32 G, u64 n_brackets, u32 *__restrict__ cnt)
{
  __shared__ u32 c_s[4][NWAY_MAX];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const u64 br = (u64) blockIdx.x * 4 + w;
  if (lane < NWAY_MAX) c_s[w][lane] = 0;
  __syncthreads ();
  if (br < n_brackets) {
    const u64 first = br * NWAY_BRACKET * G, end = first + (u64) NWAY_BRACKET * G < m_total ? first + (u64) NWAY_BRACKET * G : m_total;
    for (u64 j = first + lane; j < end; j += WAVE) {
      const u32 id = merged[3 * j + 2];
      if (id < (u32) NWAY_MAX) atomicAdd (&c_s[w][id], 1u);
    }
  }
  __syncthreads ();
  if (br < n_brackets && lane < NWAY_MAX) cnt[br * NWAY_MAX + lane] = c_s[w][lane];
}

__global__ __launch_bounds__ (1024) void k_nway_bracket_bases (u32 *__restrict__ cnt, u64 n_brackets)
{
  const int lane = threadIdx.x & 63;
  for (int list = threadIdx.x >> 6; list < NWAY_MAX; list += 16) {
    u64 carry = 0;
    constexpr int U = 4;
    for (u64 b0 = 0; b0 < n_brackets; b0 += U * WAVE) {
      u64 v[U], sum = 0;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const u64 b = b0 + (u64) (U * lane + u);
        v[u] = b < n_brackets ? cnt[b * NWAY_MAX + list] : 0u;
      }
#pragma unroll
      for (int u = 0; u < U; u++) sum += v[u];
      const u64 incl = wave_inclusive_scan (sum, lane);
      u64 before = carry + incl - sum;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const u64 b = b0 + (u64) (U * lane + u);
        if (b < n_brackets) cnt[b * NWAY_MAX + list] = (u32) before;
        before += v[u];
      }
      carry += (u64) (u32) __builtin_amdgcn_readlane ((int) (u32) incl, WAVE - 1) | ((u64) (u32) __builtin_amdgcn_readlane ((int) (u32) (incl >> 32), WAVE - 1) << 32);
    }
  }
}

constexpr u32 NWAY_G_MAX = 64; /* samples per tile the bracket's LDS copy has room for */

/* (many lists: only the samples' LIST NUMBERS are staged -- one byte each, 4 KB per bracket instead of 48 KB of whole
 * samples, which held the kernel to three wavefronts per CU: 8.8 ms of a 48 ms union of 32 lists.  The tiles' own
 * samples are not tested for clustering here; the tile kernel finds clustered tiles by their longest bucket.) */
__global__ __launch_bounds__ (64, 5) void k_nway_partition_rows (NwayParams p, const u32 *__restrict__ merged, u64 m_total, u32 G, u32 n_buckets, const u32 *__restrict__ bases,
                                                           u64 *__restrict__ part)
{
  __shared__ unsigned char sid_s[NWAY_BRACKET * NWAY_G_MAX];
  const int lane = threadIdx.x;
  const u64 br = blockIdx.x;
  const u64 t = br * NWAY_BRACKET + lane;
  {
    const u64 f = br * NWAY_BRACKET * G;
    const u64 cnt = f >= m_total ? 0 : (m_total - f < (u64) NWAY_BRACKET * G ? m_total - f : (u64) NWAY_BRACKET * G);
    for (u32 i = lane; i < (u32) cnt; i += WAVE) sid_s[i] = (unsigned char) merged[3 * (f + i) + 2];
    __syncthreads ();
  }
  const u64 nt = p.num_tiles;
  const bool row = t <= nt;
  const bool has_x = row && t > 0 && t < nt, has_y = row && t + 1 < nt;
  const u64 x = has_x ? nway_boundary_key (merged, m_total, G, p.num_tiles, t) : 0ull;
  const u64 y = has_y ? nway_boundary_key (merged, m_total, G, p.num_tiles, t + 1) : 0ull;
  const u32 x_list = has_x ? merged[3 * (t == nt - 1 ? m_total - 1 : t * (u64) G - 1) + 2] : 0xffffffffu;
  const bool tile = row && t < nt;
  const u64 first = t * (u64) G, end = !tile || first >= m_total ? first : (first + G < m_total ? first + G : m_total);
  /* the tile's key range and bucket function (as k_nway_partition) */
  if (tile) {
    u64 lo, hi;
    if (t == 0) {
      lo = ~0ull;
      for (u32 j = 0; j < p.k; j++)
        if (p.n[j]) {
          const u64 f = load_key (p.list[j], 0);
          lo = f < lo ? f : lo;
        }
    } else {
      lo = x + 1ull;
    }
    if (t + 1 == nt) {
      hi = 0;
      for (u32 j = 0; j < p.k; j++)
        if (p.n[j]) {
          const u64 l = load_key (p.list[j], p.n[j] - 1);
          hi = l > hi ? l : hi;
        }
    } else {
      hi = y;
    }
    part[t * NWAY_PSTRIDE + NWAY_MAX] = lo;
    part[t * NWAY_PSTRIDE + NWAY_MAX + 1] = nway_bucket_consts (lo, hi, n_buckets);
  } else if (row) {
    part[t * NWAY_PSTRIDE + NWAY_MAX] = 0;
    part[t * NWAY_PSTRIDE + NWAY_MAX + 1] = 0;
  }
  /* the cuts, eight lists at a time */
  for (u32 g0 = 0; g0 < (u32) NWAY_MAX; g0 += 8) {
    u64 c0 = 0, c1 = 0; /* the tile's own samples of lists g0 .. g0 + 7: 16-bit fields */
    if (tile) {
      for (u64 j0 = first; j0 < end; j0++) {
        const u32 j = (u32) (j0 - br * NWAY_BRACKET * G);
        const u32 sid = (u32) sid_s[j] - g0;
        const u64 one = 1ull << (16 * (sid & 3u));
        c0 += sid < 4u ? one : 0ull;
        c1 += (sid >= 4u && sid < 8u) ? one : 0ull;
      }
    }
    c0 = wave_inclusive_scan (c0, lane) - c0; /* (every lane takes part) */
    c1 = wave_inclusive_scan (c1, lane) - c1;
    if (tile) {
      u64 a[8], h[8];
      bool need[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const u32 li = g0 + (u32) i;
        const u64 c = li < p.k && t > 0 ? (u64) bases[br * NWAY_MAX + li] + (((i < 4 ? c0 : c1) >> (16 * (i & 3))) & 0xffffu) : 0ull;
        a[i] = h[i] = c * NWAY_SAMPLE;
        need[i] = li < p.k && t > 0 && li != x_list;
      }
      for (int round = 0; round < 3; round++) {
        u64 e[8], kk[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const u64 ni = p.n[g0 + i];
          e[i] = a[i] + NWAY_SAMPLE < ni ? a[i] + NWAY_SAMPLE : ni;
          kk[i] = need[i] && e[i] > a[i] ? load_key (p.list[g0 + i], e[i] - 1) : 0ull;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if (!need[i]) continue;
          if (e[i] == a[i]) {
            h[i] = a[i];
            need[i] = false;
          } else if (kk[i] <= x) {
            a[i] = h[i] = e[i];
          } else {
            h[i] = e[i] - 1;
            need[i] = false;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 8; i++)
        if (need[i]) h[i] = p.n[g0 + i];
      for (;;) {
        bool any = false;
        u64 km[8];
#pragma unroll
        for (int i = 0; i < 8; i++) km[i] = a[i] < h[i] ? load_key (p.list[g0 + i], (a[i] + h[i]) >> 1) : 0ull;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if (a[i] >= h[i]) continue;
          const u64 mid = (a[i] + h[i]) >> 1;
          if (km[i] <= x) a[i] = mid + 1;
          else h[i] = mid;
          any |= a[i] < h[i];
        }
        if (!any) break;
      }
#pragma unroll
      for (int i = 0; i < 8; i++) part[t * NWAY_PSTRIDE + g0 + i] = g0 + (u32) i < p.k && t > 0 ? a[i] : 0ull;
    } else if (row) { /* t == num_tiles: the lists' ends */
      for (u32 i = 0; i < 8; i++) part[t * NWAY_PSTRIDE + g0 + i] = g0 + i < p.k ? p.n[g0 + i] : 0ull;
    }
  }
}
#else
__global__ __launch_bounds__ (256) void k_nway_sample_counts (const u32 *__restrict__ merged, u64 m_total, u32 G, u64 n_brackets, u32 *__restrict__ cnt)
{
  const int lane = threadIdx.x & 63;
  const u64 br = (u64) blockIdx.x * 4 + (threadIdx.x >> 6);
  if (br >= n_brackets) return;
  const u64 first = br * NWAY_BRACKET * G, end = first + (u64) NWAY_BRACKET * G < m_total ? first + (u64) NWAY_BRACKET * G : m_total;
  u64 c0 = 0, c1 = 0; /* 16-bit fields: lists 0..3, 4..7 (a bracket has at most 64 * 32 samples) */
  for (u64 j = first + lane; j < end; j += WAVE) {
    const u32 id = merged[3 * j + 2];
    const u64 one = 1ull << (16 * (id & 3u));
    c0 += id < 4u ? one : 0ull;
    c1 += id < 4u ? 0ull : one;
  }
  c0 = wave_sum (c0);
  c1 = wave_sum (c1);
  if (lane < NWAY_MAX) cnt[br * NWAY_MAX + lane] = (u32) (((lane < 4 ? c0 : c1) >> (16 * (lane & 3))) & 0xffffu);
}

__global__ __launch_bounds__ (64 * NWAY_MAX) void k_nway_bracket_bases (u32 *__restrict__ cnt, u64 n_brackets)
{
  const int lane = threadIdx.x & 63, list = threadIdx.x >> 6;
  u64 carry = 0;
  constexpr int U = 4; /* brackets per lane and round: the loads of a round are asked for together (one per lane and round was a memory round trip per 64 brackets: 0.18 ms for 2e4 brackets) */
  for (u64 b0 = 0; b0 < n_brackets; b0 += U * WAVE) {
    u64 v[U], sum = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const u64 b = b0 + (u64) (U * lane + u);
      v[u] = b < n_brackets ? cnt[b * NWAY_MAX + list] : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; u++) sum += v[u];
    const u64 incl = wave_inclusive_scan (sum, lane);
    u64 before = carry + incl - sum;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const u64 b = b0 + (u64) (U * lane + u);
      if (b < n_brackets) cnt[b * NWAY_MAX + list] = (u32) before;
      before += v[u];
    }
    carry += (u64) (u32) __builtin_amdgcn_readlane ((int) (u32) incl, WAVE - 1) | ((u64) (u32) __builtin_amdgcn_readlane ((int) (u32) (incl >> 32), WAVE - 1) << 32);
  }
}

constexpr u32 NWAY_G_MAX = 32; /* samples per tile the bracket's LDS copy has room for */

__global__ __launch_bounds__ (64) void k_nway_partition_rows (NwayParams p, const u32 *__restrict__ merged, u64 m_total, u32 G, u32 n_buckets, const u32 *__restrict__ bases,
                                                           u64 *__restrict__ part)
{
  __shared__ u32x4 smp4[NWAY_BRACKET * NWAY_G_MAX * 3 / 4]; /* the bracket's samples: read once, 16 bytes per lane and instruction */
  const u32 *const smp = reinterpret_cast<const u32 *> (smp4);
  const int lane = threadIdx.x;
  const u64 br = blockIdx.x;
  const u64 t = br * NWAY_BRACKET + lane;
  {
    const u64 f = br * NWAY_BRACKET * G;
    const u64 cnt = f >= m_total ? 0 : (m_total - f < (u64) NWAY_BRACKET * G ? m_total - f : (u64) NWAY_BRACKET * G);
    const u32 quads = (u32) ((3 * cnt + 3) / 4); /* (the list's allocation is a multiple of 16 bytes and f * 12 is one too) */
    const u32x4 *src = reinterpret_cast<const u32x4 *> (merged + 3 * f);
    for (u32 i = lane; i < quads; i += WAVE) smp4[i] = src[i];
    __syncthreads ();
  }
  const u64 nt = p.num_tiles;
  const bool row = t <= nt;
  /* boundary keys in front of this tile and behind it */
  const bool has_x = row && t > 0 && t < nt, has_y = row && t + 1 < nt;
  const u64 x = has_x ? nway_boundary_key (merged, m_total, G, p.num_tiles, t) : 0ull;
  const u64 y = has_y ? nway_boundary_key (merged, m_total, G, p.num_tiles, t + 1) : 0ull;
  /* the list the boundary sample came from: its cut is behind that very sample, no search */
  const u32 x_list = has_x ? merged[3 * (t == nt - 1 ? m_total - 1 : t * (u64) G - 1) + 2] : 0xffffffffu;
  /* the tile's key range and bucket function (as k_nway_partition) */
  u64 lo_key = 0, bk = 0;
  u32 sh = 0, mul = 0;
  bool direct = false;
  const bool tile = row && t < nt;
  u64 c0 = 0, c1 = 0; /* the tile's own samples per list: 16-bit fields, lists 0..3 and 4..7 */
  if (tile) {
    u64 lo, hi;
    if (t == 0) {
      lo = ~0ull;
      for (u32 j = 0; j < p.k; j++)
        if (p.n[j]) {
          const u64 f = load_key (p.list[j], 0);
          lo = f < lo ? f : lo;
        }
    } else {
      lo = x + 1ull;
    }
    if (t + 1 == nt) {
      hi = 0;
      for (u32 j = 0; j < p.k; j++)
        if (p.n[j]) {
          const u64 l = load_key (p.list[j], p.n[j] - 1);
          hi = l > hi ? l : hi;
        }
    } else {
      hi = y;
    }
    const u64 D = hi >= lo ? hi - lo : 0ull;
    const u32 bl = D ? 64u - (u32) __builtin_clzll (D) : 0u;
    sh = bl > 32u ? bl - 32u : 0u;
    const u32 vmax = (u32) (D >> sh);
    direct = vmax < n_buckets;
    mul = direct ? 0u : (u32) (((u64) n_buckets << 32) / ((u64) vmax + 1ull));
    lo_key = lo;
    bk = (u64) sh | (direct ? 1ull << 8 : (u64) mul << 32);
    /* the tile's own samples: counts per list, and whether the interpolation will work on them */
    const u64 first = t * (u64) G, end = first >= m_total ? first : (first + G < m_total ? first + G : m_total);
    u32 prev = 0xffffffffu, same = 0, cnt = 0;
    u64 prev_key = 0;
    bool have_prev = false;
    for (u64 j0 = first; j0 < end; j0 += 8) { /* (eight samples asked for at once) */
      u64 sk[8];
      u32 sid[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const u32 j = (u32) ((j0 + u < end ? j0 + u : end - 1) - br * NWAY_BRACKET * G);
        sk[u] = (u64) smp[3 * j] | ((u64) smp[3 * j + 1] << 32);
        sid[u] = smp[3 * j + 2];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (j0 + u >= end) continue;
        const u64 s = sk[u];
        const u64 one = 1ull << (16 * (sid[u] & 3u));
        c0 += sid[u] < 4u ? one : 0ull;
        c1 += sid[u] < 4u ? 0ull : one;
        if (s < lo || s > hi || (have_prev && s == prev_key)) continue; /* (equal keys of different lists share a bucket by right) */
        prev_key = s;
        have_prev = true;
        const u32 vv = (u32) ((s - lo) >> sh);
        const u32 b = direct ? vv : __umulhi (vv, mul);
        same += b == prev ? 1u : 0u;
        prev = b;
        cnt++;
      }
    }
    if (t + 1 < nt && cnt >= 8 && 2 * same > cnt) bk |= 1ull << 9;
  }
  /* samples in front of the tile = the bracket's base + the earlier lanes' (every lane takes part) */
  c0 = wave_inclusive_scan (c0, lane) - c0;
  c1 = wave_inclusive_scan (c1, lane) - c1;
  if (tile) {
    /* the eight searches in step: every round asks for one key of every list */
    u64 a[NWAY_MAX], h[NWAY_MAX];
    bool need[NWAY_MAX]; /* the stretch that holds the cut is not found yet */
#pragma unroll
    for (int i = 0; i < NWAY_MAX; i++) {
      const u64 c = (u32) i < p.k && t > 0 ? (u64) bases[br * NWAY_MAX + i] + (((i < 4 ? c0 : c1) >> (16 * (i & 3))) & 0xffffu) : 0ull;
      a[i] = h[i] = c * NWAY_SAMPLE;
      need[i] = (u32) i < p.k && t > 0 && (u32) i != x_list; /* (c counts the boundary sample itself: c * S is one behind it) */
    }
    for (int round = 0; round < 3; round++) { /* (a sample equal to the boundary key merged behind it: one stretch further; the list's tail: one more) */
      u64 e[NWAY_MAX], kk[NWAY_MAX];
#pragma unroll
      for (int i = 0; i < NWAY_MAX; i++) {
        e[i] = a[i] + NWAY_SAMPLE < p.n[i] ? a[i] + NWAY_SAMPLE : p.n[i];
        kk[i] = need[i] && e[i] > a[i] ? load_key (p.list[i], e[i] - 1) : 0ull;
      }
#pragma unroll
      for (int i = 0; i < NWAY_MAX; i++) {
        if (!need[i]) continue;
        if (e[i] == a[i]) { /* the list ends here */
          h[i] = a[i];
          need[i] = false;
        } else if (kk[i] <= x) { /* the whole stretch belongs to earlier tiles */
          a[i] = h[i] = e[i];
        } else {
          h[i] = e[i] - 1;
          need[i] = false;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NWAY_MAX; i++)
      if (need[i]) h[i] = p.n[i]; /* (cannot happen: keys are unique inside a list; searched in full all the same) */
    for (;;) {
      bool any = false;
      u64 km[NWAY_MAX];
#pragma unroll
      for (int i = 0; i < NWAY_MAX; i++) km[i] = a[i] < h[i] ? load_key (p.list[i], (a[i] + h[i]) >> 1) : 0ull;
#pragma unroll
      for (int i = 0; i < NWAY_MAX; i++) {
        if (a[i] >= h[i]) continue;
        const u64 mid = (a[i] + h[i]) >> 1;
        if (km[i] <= x) a[i] = mid + 1;
        else h[i] = mid;
        any |= a[i] < h[i];
      }
      if (!any) break;
    }
#pragma unroll
    for (int i = 0; i < NWAY_MAX; i++) part[t * NWAY_PSTRIDE + i] = (u32) i < p.k && t > 0 ? a[i] : 0ull;
    part[t * NWAY_PSTRIDE + NWAY_MAX] = lo_key;
    part[t * NWAY_PSTRIDE + NWAY_MAX + 1] = bk;
  } else if (row) { /* t == num_tiles: the lists' ends */
    for (u32 i = 0; i < NWAY_MAX; i++) part[t * NWAY_PSTRIDE + i] = i < p.k ? p.n[i] : 0ull;
    part[t * NWAY_PSTRIDE + NWAY_MAX] = 0;
    part[t * NWAY_PSTRIDE + NWAY_MAX + 1] = 0;
  }
}
#endif /* GT4_KM > 8 */

/* ---- Will the interpolation work on these keys?  A probe of the longest list in front of everything else:
 * every workgroup takes a window of NWAY_PROBE_KEYS consecutive records -- about the key range of one tile --
 * and counts the keys that fall into the bucket of their predecessor under the tile kernel's own bucket
 * function over the window's key range.  Evenly spread keys share hardly any of the window's buckets (a
 * quarter to a third of them do); stretches of adjacent keys between wide gaps put nearly all of them into a
 * few.  flagged[0] += 1 per window in which more than 60 % do: such tiles take the pivot-run buckets or the
 * search path (two to three times the time of a tile), and beyond a fifth of the tiles the pairwise tree of
 * the pair kernel is the faster union (profiles/round4: 27 ms against 52 ms on clustered lists of 2.5e8). */
constexpr u32 NWAY_PROBE_KEYS = 3072;
constexpr u32 NWAY_PROBE_WINDOWS = 1024;

__global__ __launch_bounds__ (256) void k_nway_probe (const u32 *__restrict__ list, u64 n, u32 windows, u32 n_buckets, u32 *__restrict__ flagged)
{
  __shared__ u32 same_s;
  if (threadIdx.x == 0) same_s = 0;
  __syncthreads ();
  const u64 first = (u64) (((unsigned __int128) (n - NWAY_PROBE_KEYS) * blockIdx.x) / (windows > 1 ? windows - 1 : 1));
  const u64 lo = load_key (list, first), hi = load_key (list, first + NWAY_PROBE_KEYS - 1);
  const u64 D = hi - lo;
  const u32 bl = D ? 64u - (u32) __builtin_clzll (D) : 0u;
  const u32 sh = bl > 32u ? bl - 32u : 0u;
  const u32 vmax = (u32) (D >> sh);
  const bool direct = vmax < n_buckets;
  const u32 mul = direct ? 0u : (u32) (((u64) n_buckets << 32) / ((u64) vmax + 1ull));
  u32 same = 0;
  for (u32 i = 1 + threadIdx.x; i < NWAY_PROBE_KEYS; i += blockDim.x) {
    const u32 v0 = (u32) ((load_key (list, first + i - 1) - lo) >> sh), v1 = (u32) ((load_key (list, first + i) - lo) >> sh);
    const u32 b0 = direct ? v0 : __umulhi (v0, mul), b1 = direct ? v1 : __umulhi (v1, mul);
    same += b0 == b1 ? 1u : 0u;
  }
  same = dpp_wave_sum_u32 (same);
  if ((threadIdx.x & 63) == 0) atomicAdd (&same_s, same);
  __syncthreads ();
  if (threadIdx.x == 0 && 10u * same_s > 6u * NWAY_PROBE_KEYS) atomicAdd (flagged, 1u);
}

/* ---- Tiles that would not fit LDS are cut in two (round 4).  A tile holds G merged samples' worth of records
 * plus what the lists' offsets against their sample grids add (sigma = S sqrt (k / 6) records); G sits five sigma
 * below the capacity.  Round 3 repeated the whole partition with fewer samples per tile when any tile overflowed
 * all the same (lists of very different density); now such a tile is cut at the middle key of its longest run and
 * only a tile that needs more than two pieces sends the call back.  (Fuller tiles -- G two and a half sigma below,
 * one tile in fourteen cut -- were the reason to build this and measured SLOWER: GT4_NWAY_MARGIN.)
 *   k_nway_need    per nominal tile: 1, or 2 when its wave slots exceed the capacity (more than two: the old retry);
 *                  sums per block of NWAY_SPLIT_BLOCK tiles
 *   k_nway_need_scan  exclusive prefix over the blocks (one workgroup)
 *   k_nway_emit    the final table: row base + prefix inside the block; the second half's cuts by eight
 *                  upper bounds of the pivot key inside the tile's runs, key ranges and bucket constants per half */
constexpr u32 NWAY_SPLIT_BLOCK = 1024;

__device__ __forceinline__ u64 nway_bucket_consts (u64 lo, u64 hi, u32 n_buckets)
{
  const u64 D = hi >= lo ? hi - lo : 0ull;
  const u32 bl = D ? 64u - (u32) __builtin_clzll (D) : 0u;
  const u32 sh = bl > 32u ? bl - 32u : 0u;
  const u32 vmax = (u32) (D >> sh);
  const bool direct = vmax < n_buckets;
  const u32 mul = direct ? 0u : (u32) (((u64) n_buckets << 32) / ((u64) vmax + 1ull));
  return (u64) sh | (direct ? 1ull << 8 : (u64) mul << 32);
}

__device__ __forceinline__ u32 nway_tile_slots (const u64 *__restrict__ part, u64 t, bool *mono, u32 *records)
{
  u64 slots = 0, recs = 0;
  for (int i = 0; i < NWAY_MAX; i++) {
    const u64 a = part[t * NWAY_PSTRIDE + i], b = part[(t + 1) * NWAY_PSTRIDE + i];
    *mono &= b >= a;
    slots += (b - a + NWAY_HS - 1) / NWAY_HS;
    recs += b >= a ? b - a : 0;
  }
  *records = recs > 0xffffffffull ? 0xffffffffu : (u32) recs;
  return slots > 0xffffffffull ? 0xffffffffu : (u32) slots;
}

/* flag[0]: a tile needs more than two pieces (or the table is not monotone); flag[1]: tiles whose samples look
 * clustered; flag[2]: tiles cut in two */
#if GT4_KM > 8
/* (many lists: a partition row is 34 entries -- one HALF-wavefront per tile reads its two rows side by side, 256 bytes
 * per load; a thread per tile read them 272 bytes apart: 0.63 ms per launch, 3.2 ms of a 46 ms union of 32 lists) */
__global__ __launch_bounds__ (NWAY_SPLIT_BLOCK) void k_nway_need (const u64 *__restrict__ part, u32 num_tiles, u32 nch, u32 max_rec, u32 *__restrict__ need, u32 *__restrict__ block_sums, u32 *flag)
{
  static_assert (NWAY_MAX == 32 && NWAY_SPLIT_BLOCK == 1024, "a half-wavefront per tile, sixteen wavefronts per block of 1024 tiles");
  __shared__ u32 ws[NWAY_SPLIT_BLOCK / WAVE];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, half = lane >> 5, li = lane & 31;
  u32 vsum = 0;
  for (int it = 0; it < 32; it++) {
    const u64 t = (u64) blockIdx.x * NWAY_SPLIT_BLOCK + (u64) wid * 64 + (u64) (2 * it + half);
    const bool in = t < num_tiles;
    const u64 a = in ? part[t * NWAY_PSTRIDE + li] : 0ull, b = in ? part[(t + 1) * NWAY_PSTRIDE + li] : 0ull;
    const bool mono = b >= a;
    const u64 len = mono ? b - a : 0ull;
    const u32 r = len > 0xffffffull ? 0xffffffu : (u32) len;               /* (a tile of more than 2^24 records of one list: refused anyway) */
    const u32 sl = (r + (u32) NWAY_HS - 1u) / (u32) NWAY_HS;
    const u32 ri = dpp_inclusive_scan_u32 (r), si = dpp_inclusive_scan_u32 (sl);
    const u32 r0 = (u32) __builtin_amdgcn_readlane ((int) ri, 31), r1 = (u32) __builtin_amdgcn_readlane ((int) ri, 63) - r0;
    const u32 s0 = (u32) __builtin_amdgcn_readlane ((int) si, 31), s1 = (u32) __builtin_amdgcn_readlane ((int) si, 63) - s0;
    const u64 bad = __builtin_amdgcn_ballot_w64 (!mono);
    const bool bad_h = half ? (bad >> 32) != 0 : (u32) bad != 0u;
    const u32 recs = half ? r1 : r0, slots = half ? s1 : s0;
    u32 v = 0;
    if (in && li == 0) {
      v = slots <= nch && recs <= max_rec ? 1u : 2u;
      if (bad_h || slots > 2 * nch - 2 * NWAY_MAX || recs / 2 > max_rec) atomicOr (flag, 1u);
      if ((part[t * NWAY_PSTRIDE + NWAY_MAX + 1] >> 9) & 1ull) atomicAdd (flag + 1, 1u);
      if (v == 2u) atomicAdd (flag + 2, 1u);
      need[t] = v;
    }
    vsum += v;
  }
  vsum = dpp_wave_sum_u32 (vsum);
  if (lane == 0) ws[wid] = vsum;
  __syncthreads ();
  if (threadIdx.x == 0) {
    u32 sum = 0;
    for (u32 w = 0; w < NWAY_SPLIT_BLOCK / WAVE; w++) sum += ws[w];
    block_sums[blockIdx.x] = sum;
  }
}
#else
__global__ __launch_bounds__ (NWAY_SPLIT_BLOCK) void k_nway_need (const u64 *__restrict__ part, u32 num_tiles, u32 nch, u32 max_rec, u32 *__restrict__ need, u32 *__restrict__ block_sums, u32 *flag)
{
  __shared__ u32 ws[NWAY_SPLIT_BLOCK / WAVE];
  const u64 t = (u64) blockIdx.x * NWAY_SPLIT_BLOCK + threadIdx.x;
  u32 v = 0;
  if (t < num_tiles) {
    bool mono = true;
    u32 recs;
    const u32 slots = nway_tile_slots (part, t, &mono, &recs);
    v = slots <= nch && recs <= max_rec ? 1u : 2u; /* (max_rec: the records the kernel's wavefronts take between them; k_nway_sub) */
    if (!mono || slots > 2 * nch - 2 * NWAY_MAX || recs / 2 > max_rec) atomicOr (flag, 1u); /* (each half rounds every run up once more) */
    if ((part[t * NWAY_PSTRIDE + NWAY_MAX + 1] >> 9) & 1ull) atomicAdd (flag + 1, 1u);
    if (v == 2u) atomicAdd (flag + 2, 1u);
    need[t] = v;
  }
  v = dpp_wave_sum_u32 (v);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x / WAVE] = v;
  __syncthreads ();
  if (threadIdx.x == 0) {
    u32 sum = 0;
    for (u32 w = 0; w < NWAY_SPLIT_BLOCK / WAVE; w++) sum += ws[w];
    block_sums[blockIdx.x] = sum;
  }
}

#endif /* GT4_KM > 8 */

__global__ __launch_bounds__ (1024) void k_nway_need_scan (u32 *__restrict__ block_sums, u32 n_blocks, u32 *__restrict__ total)
{
  __shared__ u32 wsum[16];
  __shared__ u32 carry_s;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads ();
  for (u32 b0 = 0; b0 < n_blocks; b0 += 1024) {
    const u32 i = b0 + threadIdx.x;
    const u32 v = i < n_blocks ? block_sums[i] : 0u;
    const u32 incl = dpp_inclusive_scan_u32 (v);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads ();
    u32 before = 0, all = 0;
    for (int w = 0; w < 16; w++) {
      const u32 x = wsum[w];
      before += w < wid ? x : 0u;
      all += x;
    }
    const u32 c = carry_s;
    if (i < n_blocks) block_sums[i] = c + before + incl - v;
    __syncthreads ();
    if (threadIdx.x == 0) carry_s = c + all;
    __syncthreads ();
  }
  if (threadIdx.x == 0) *total = carry_s;
}

__global__ __launch_bounds__ (NWAY_SPLIT_BLOCK) void k_nway_emit (NwayParams p, const u64 *__restrict__ part, u32 num_tiles, const u32 *__restrict__ need, const u32 *__restrict__ block_base,
                                                                 u32 n_buckets, u32 nch, u32 max_rec, u64 *__restrict__ out, u32 *flag)
{
  __shared__ u32 ws[NWAY_SPLIT_BLOCK / WAVE];
  const u64 t = (u64) blockIdx.x * NWAY_SPLIT_BLOCK + threadIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const u32 v = t < num_tiles ? need[t] : 0u;
  const u32 incl = dpp_inclusive_scan_u32 (v);
  if (lane == 63) ws[wid] = incl;
  __syncthreads ();
  u32 before = block_base[blockIdx.x];
  for (int w = 0; w < wid; w++) before += ws[w];
  const u64 at = (u64) before + incl - v; /* the tile's (first) row in the final table */
  if (t == num_tiles) { /* the lists' ends */
    for (int i = 0; i < NWAY_PSTRIDE; i++) out[at * NWAY_PSTRIDE + i] = part[t * NWAY_PSTRIDE + i];
    return;
  }
  if (t > num_tiles) return;
  if (v == 1u) {
    for (int i = 0; i < NWAY_PSTRIDE; i++) out[at * NWAY_PSTRIDE + i] = part[t * NWAY_PSTRIDE + i];
    return;
  }
  const u64 row_lo = part[t * NWAY_PSTRIDE + NWAY_MAX], row_bk = part[t * NWAY_PSTRIDE + NWAY_MAX + 1];
  /* two pieces: the pivot is the middle key of the tile's longest run; keys <= pivot go left (equal keys of different
   * lists stay together).  (No per-list arrays: thirty-two lists' worth would not stay in registers.) */
  u32 longest = 0;
  u64 best = 0;
  for (int i = 0; i < NWAY_MAX; i++) {
    const u64 len_i = part[(t + 1) * NWAY_PSTRIDE + i] - part[t * NWAY_PSTRIDE + i];
    if ((u32) i < p.k && len_i > best) {
      best = len_i;
      longest = (u32) i;
    }
  }
  const u64 pivot = load_key (p.list[longest], part[t * NWAY_PSTRIDE + longest] + (best - 1) / 2);
  /* the tile's largest possible key: the next tile's smallest minus one; the last tile ends at the lists' largest key */
  u64 hi_key;
  if (t + 1 < num_tiles) {
    hi_key = part[(t + 1) * NWAY_PSTRIDE + NWAY_MAX] - 1ull;
  } else {
    hi_key = 0;
    for (u32 j = 0; j < p.k; j++)
      if (p.n[j]) {
        const u64 l = load_key (p.list[j], p.n[j] - 1);
        hi_key = l > hi_key ? l : hi_key;
      }
  }
  const u64 clustered = row_bk & (1ull << 9);
  /* a piece that still does not fit (runs of very different length: the pivot halves the longest only) sends the
   * call back to fewer samples per tile */
  u64 s0 = 0, s1 = 0, r0 = 0, r1 = 0;
  for (u32 i = 0; i < (u32) NWAY_MAX; i++) {
    const u64 begin_i = part[t * NWAY_PSTRIDE + i], end_i = part[(t + 1) * NWAY_PSTRIDE + i];
    u64 lo = begin_i, hi = end_i;
    if (i >= p.k) hi = lo;
    while (lo < hi) {
      const u64 m = (lo + hi) >> 1;
      if (load_key (p.list[i], m) <= pivot) lo = m + 1;
      else hi = m;
    }
    out[at * NWAY_PSTRIDE + i] = begin_i;
    out[(at + 1) * NWAY_PSTRIDE + i] = lo;
    s0 += (lo - begin_i + NWAY_HS - 1) / NWAY_HS;
    s1 += (end_i - lo + NWAY_HS - 1) / NWAY_HS;
    r0 += lo - begin_i;
    r1 += end_i - lo;
  }
  out[at * NWAY_PSTRIDE + NWAY_MAX] = row_lo;
  out[at * NWAY_PSTRIDE + NWAY_MAX + 1] = nway_bucket_consts (row_lo, pivot, n_buckets) | clustered;
  out[(at + 1) * NWAY_PSTRIDE + NWAY_MAX] = pivot + 1ull; /* (pivot < hi_key: the right piece holds a larger key) */
  out[(at + 1) * NWAY_PSTRIDE + NWAY_MAX + 1] = nway_bucket_consts (pivot + 1ull, hi_key, n_buckets) | clustered;
  if (s0 > nch || s1 > nch || r0 > max_rec || r1 > max_rec) atomicOr (flag, 1u);
}

/* where every tile's rows start in the ragged table = the records in front of the tile (the sum of its cuts) */
__global__ void k_nway_padded_bases (const u64 *__restrict__ part, u64 tiles, u32 k, u64 *__restrict__ padded)
{
  const u64 t = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (t > tiles) return;
  u64 sum = 0;
  for (u32 i = 0; i < k; i++) sum += part[t * NWAY_PSTRIDE + i];
  padded[t] = sum;
}

/* rows before every tile = exclusive prefix of the tiles' distinct keys, in three small launches (one workgroup
 * walking 2e5 tiles took 0.28 ms of a 7 ms table): sums per block of 1024 tiles, their prefix, the tiles' own */
__global__ __launch_bounds__ (1024) void k_nway_base_sums (const u32 *__restrict__ totals, u64 tiles, u64 *__restrict__ block_sums)
{
  __shared__ u64 ws[16];
  const u64 i = (u64) blockIdx.x * 1024 + threadIdx.x;
  const u64 v = wave_sum (i < tiles ? (u64) totals[i] : 0ull);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads ();
  if (threadIdx.x == 0) {
    u64 sum = 0;
    for (int w = 0; w < 16; w++) sum += ws[w];
    block_sums[blockIdx.x] = sum;
  }
}

__global__ __launch_bounds__ (1024) void k_nway_base_scan (u64 *__restrict__ block_sums, u64 n_blocks)
{
  __shared__ u64 wsum[16];
  __shared__ u64 carry_s;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads ();
  for (u64 b0 = 0; b0 < n_blocks; b0 += 1024) {
    const u64 i = b0 + threadIdx.x;
    const u64 v = i < n_blocks ? block_sums[i] : 0ull;
    const u64 incl = wave_inclusive_scan (v, lane);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads ();
    u64 before = 0, all = 0;
    for (int w = 0; w < 16; w++) {
      const u64 x = wsum[w];
      before += w < wid ? x : 0;
      all += x;
    }
    const u64 c = carry_s;
    if (i < n_blocks) block_sums[i] = c + before + incl - v;
    __syncthreads ();
    if (threadIdx.x == 0) carry_s = c + all;
    __syncthreads ();
  }
}

/* bases[t] for t <= tiles (bases[tiles] = the total) */
__global__ __launch_bounds__ (1024) void k_nway_tile_bases (const u32 *__restrict__ totals, u64 tiles, const u64 *__restrict__ block_base, u64 *__restrict__ bases)
{
  __shared__ u64 wsum[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const u64 i = (u64) blockIdx.x * 1024 + threadIdx.x;
  const u64 v = i < tiles ? (u64) totals[i] : 0ull;
  const u64 incl = wave_inclusive_scan (v, lane);
  if (lane == 63) wsum[wid] = incl;
  __syncthreads ();
  u64 before = block_base[blockIdx.x];
  for (int w = 0; w < wid; w++) before += wsum[w];
  if (i <= tiles) bases[i] = before + incl - v;
}

/* ------------------------------------------------------------------ K7: the tile kernel */

#ifndef GT4_NWAY_NT
#define GT4_NWAY_NT 1024
#endif
#ifndef GT4_NWAY_RPT
#define GT4_NWAY_RPT 4
#endif
#ifndef GT4_NWAY_NBF
#define GT4_NWAY_NBF 1
#endif
#ifndef GT4_NWAY_SVPRIO
#define GT4_NWAY_SVPRIO 3
#endif
#ifndef GT4_TABLE_STORE_AUX
#define GT4_TABLE_STORE_AUX 0 /* cache policy of the count tables' row stores (see gt4hip_device.h) */
#endif
#ifndef GT4_NWAY_ROWW
#define GT4_NWAY_ROWW 14336 /* words of the count tables' row area in LDS (56 KB: 159 of 160 KB with it; 12288: 1 - 2 % slower, 10240: 2 - 3 %) */
#endif
/* Round 4: the per-tile work that does not depend on the number of records is 58 % of a tile (time per tile against
 * samples per tile: 14.1 ns + 0.42 ns x G on 256 CUs, profiles/round4/r4_nway_experiments.log), part of it
 * instructions every one of the sixteen wavefronts executes.  Three cuts, each A/B-measured (31.4 -> 30.1 ms together;
 * GT4_NWAY_FILL / _SCAN4 / _LEAN = 0 restore the old forms):
 *   FILL   the grouped-key area is filled with all-ones once per tile (two 16-byte stores per thread, behind the
 *          walks of the previous tile) instead of every thread working out which skewed slots its buckets leave free
 *          (-1.0 ms);
 *   SCAN4  the bucket counters are scanned by four wavefronts (one per SIMD, eight words = sixteen counters per lane,
 *          16-byte LDS accesses) instead of sixteen (two words per lane): twelve wavefronts skip two DPP scans, two
 *          DPP maxima and their LDS traffic (-0.3 ms with FILL; +1.5 ms without it: sixteen slot tests per lane);
 *   LEAN   one DPP scan behind B6 instead of a scan and a sum. */
#ifndef GT4_NWAY_FILL
#define GT4_NWAY_FILL 1
#endif
#ifndef GT4_NWAY_SCAN4
#define GT4_NWAY_SCAN4 1
#endif
#ifndef GT4_NWAY_LEAN
#define GT4_NWAY_LEAN 1
#endif

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

/* buckets of a tile: about one per position, a multiple of 2048 (the counters are scanned 16 bytes at a time by 256 lanes) */
__host__ __device__ constexpr int nway_buckets (int positions) { return (positions + 2047) / 2048 * 2048; }

template <int NT, int RPT, int NBF, int MODE>
struct NwayShared {
  static constexpr int CAP = NT * RPT;     /* positions = records a tile may hold (runs rounded up to 64) */
  static constexpr int NCH = CAP / WAVE;   /* wave slots */
  static constexpr int NB = nway_buckets (NBF * CAP); /* buckets */
  static constexpr int NW = NT / WAVE;
  /* the tile in key order: key and folded count per position -- or, in the fallback, the records
   * as sorted runs (packed 12 bytes at their positions) */
  /* Records of one list in consecutive lanes lie about as many positions apart in the ordered tile as
   * there are lists: a power-of-two stride puts 32 lanes on 4 LDS banks.  Grouped keys and ordered tile
   * are therefore SKEWED: index i lives at i + i / 32 (nway_skew), which spreads every power-of-two
   * stride over all banks (measured before: 60 % of all LDS cycles were bank conflicts). */
  static constexpr int CAPS = CAP + CAP / 32;                 /* skewed positions */
  static constexpr int GSZ = (CAPS + NWAY_LIMIT + 5) & ~1;    /* grouped keys: + the longest bucket walk behind the last key */
  static constexpr bool LEAD = nway_lead (MODE);
  /* LEAD: the bitmap holds GT4_NWAY_LEAD_BITS positions per 32-bit word.  The 64 records of a wave-instruction come
   * from one sorted list and lie about as many positions apart as there are lists, so with 32 positions per word four
   * lanes claim bits of the SAME word in one atomic instruction, which the LDS serialises; fewer positions per word
   * make the claims cheaper and the scan behind B6 longer.  Measured (8 x 5e8 stride lists, tile kernel ms): 32 bits
   * 28.72, 16 bits 28.59, 8 bits 28.84, 4 bits 29.98; independent / genomic keys gain 1 - 2 % from 8 against 32. */
  static constexpr int LBP = GT4_NWAY_LEAD_BITS;              /* positions per bitmap word (a power of two) */
  static constexpr int LWL = (CAP / LBP + WAVE - 1) / WAVE;   /* bitmap words per lane of a scanning wavefront */
  static constexpr int LW = LWL * WAVE;                       /* ... bitmap words (padded) */
  union {
    struct {
      alignas (16) u64 skey[LEAD ? GSZ + WAVE : CAPS];        /* LEAD: the grouped keys live here (+ a row nobody reads: see the trash rows) */
      u32 scnt[CAPS];
    } s;
    u32 raw[3 * CAP];
  };
  alignas (16) u64 g_own[LEAD ? 2 : GSZ + WAVE]; /* keys grouped by bucket; all-ones wherever no key is */
  __device__ __forceinline__ u64 *g () { return LEAD ? s.skey : g_own; }
  /* TRASH ROWS: the per-record steps are straight-line code -- every LDS read of a thread's RPT records is issued
   * before the first one is waited for, no exec-mask bookkeeping, no branch between them -- so a lane whose record
   * is not there (or is not kept) does its store or atomic too, into a row of WAVE words / keys / records behind the
   * array, one per lane, that nobody reads */
  alignas (16) u32 cnt[NB / 2 + 4 + WAVE]; /* 16-bit bucket counters, then bucket starts, in pairs (+ the total) (+ a trash row) */
  alignas (16) u32 live[LEAD ? 4 : (CAPS + 3) / 4]; /* one byte per position: a key was stored there */
  alignas (16) u32 lead[2][LEAD ? LW : 4]; /* LEAD: the positions that have a (kept) leader, tiles alternating */
  alignas (16) unsigned short wpre[LEAD && nway_staged (MODE) ? NT / WAVE : 1][LEAD && nway_staged (MODE) ? LW : 4]; /* LEAD: kept leaders in front of every bitmap word, per wavefront */
  /* the kept records, packed, written out during the NEXT tile (+ a trash row); the count tables: ROWW words of the
   * tile's rows at a time (see table_rows) */
  static constexpr int ROWW = GT4_NWAY_ROWW;
  static constexpr int ROW_COLS_MAX = 384;  /* wider tables: rows straight to global memory, as before round 5 */
  alignas (16) u32 stage[nway_staged (MODE) ? 3 * CAP + 4 + 3 * WAVE + 8 : ((MODE == NWAY_TABLE || MODE == NWAY_PROBE) ? ROWW + 4 : 4)];
  alignas (16) u32 wtot[NW], wmax[NW], wkept[NW];
  /* the tiles of this iteration, the next one (being fetched) and the one after (being described),
   * three deep: one 64-record wave slot per wave-instruction */
#if GT4_KM > 8
  static constexpr int NSL = 4;            /* (many lists: no slot table) */
  static constexpr int NHM = 2 * NCH;      /* stretches of 32 positions */
  u64 hmask[3][NHM];                       /* per stretch: bit i = a run starts at its position i | runs that start in front of the stretch << 32 */
  u64 rtab[3][NWAY_MAX];                   /* per NON-EMPTY run, in order: address of list record (tile position 0 - first position of the run) */
  u32 rlist[3][NWAY_MAX];                  /* ... the list it is a run of */
#else
  static constexpr int NSL = NCH;
#endif
  u64 slot_addr[3][NSL];
  alignas (16) u32 slot_cnt[3][NSL];
  u32 slot_run[3][NSL];                    /* (NWAY_TABLE, NWAY_PROBE, NWAY_DUPS) the list a slot's records come from */
  u32 tab_pbase[3][NWAY_MAX];              /* first position of each run */
  u32 tab_len[3][NWAY_MAX];
  /* tile number (0xffffffff: none), records, wave slots, shift | direct << 8, multiplier, smallest
   * possible key (2), NWAY_DUPS: where the tile's output starts (2) */
  alignas (16) u32 hdr[3][12];
  u64 excl;
  u32 tick;
};

#ifndef GT4_NWAY_WAVES
#define GT4_NWAY_WAVES 4
#endif
__host__ __device__ constexpr int nway_waves_per_simd (int nt) { return GT4_NWAY_WAVES; }

__device__ __forceinline__ u32 nway_skew (u32 i) { return i + (i >> 5); }
/* a where the mask is all ones, b where it is zero -- one bit-field insert; `c ? a : b` on a per-lane condition became
 * exec-mask bookkeeping (four scalar instructions each on the CU's one scalar unit) */
__device__ __forceinline__ u32 nway_pick (u32 mask, u32 a, u32 b) { return (a & mask) | (b & ~mask); }
__device__ __forceinline__ u32 nway_valid_mask (u32 ba) { return (u32) ((int) ba >> 31); }

/* LDS accesses by byte offset through address-space-3 pointers: the compiler keeps generic pointers for
 * loop-invariant per-thread addresses otherwise (flat loads, two registers per address) */
typedef __attribute__ ((address_space (3))) u32 lds_u32;
typedef __attribute__ ((address_space (3))) u64 lds_u64;
typedef __attribute__ ((address_space (3))) unsigned char lds_u8;
template <class T> __device__ __forceinline__ u32 lds_offset (T *p) { return (u32) (uintptr_t) p; }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
template <class T> __device__ __forceinline__ T lds_load (u32 byte_offset) { return *(__attribute__ ((address_space (3))) T *) byte_offset; }
#pragma clang diagnostic pop

/* Two steps of four bucket walks: eight INDEPENDENT 8-byte reads, one wait, eight compares.  Inline
 * assembly because the compiler merges two reads of one walk into a ds_read2_b64 (twice the LDS cycles
 * of two ds_read_b64: MI355X_MICROARCH.md, LDS table) or, told not to (volatile), waits for every
 * single read.  The wait is part of the statement: the outputs are valid behind it. */
template <int J>
__device__ __forceinline__ void nway_rank_pair (u32 a0, u32 a1, u32 a2, u32 a3, const u64 (&key)[4], u32 (&lt)[4])
{
  u64 r0, r1, r2, r3, r4, r5, r6, r7;
  asm volatile ("ds_read_b64 %0, %8 offset:%12\n\t"
                "ds_read_b64 %1, %9 offset:%12\n\t"
                "ds_read_b64 %2, %10 offset:%12\n\t"
                "ds_read_b64 %3, %11 offset:%12\n\t"
                "ds_read_b64 %4, %8 offset:%13\n\t"
                "ds_read_b64 %5, %9 offset:%13\n\t"
                "ds_read_b64 %6, %10 offset:%13\n\t"
                "ds_read_b64 %7, %11 offset:%13\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(8 * J), "n"(8 * J + 8)
                : "memory");
  lt[0] += (r0 < key[0] ? 1u : 0u) + (r4 < key[0] ? 1u : 0u);
  lt[1] += (r1 < key[1] ? 1u : 0u) + (r5 < key[1] ? 1u : 0u);
  lt[2] += (r2 < key[2] ? 1u : 0u) + (r6 < key[2] ? 1u : 0u);
  lt[3] += (r3 < key[3] ? 1u : 0u) + (r7 < key[3] ? 1u : 0u);
}

template <int J>
__device__ __forceinline__ void nway_rank_steps (u32 mx, u32 a0, u32 a1, u32 a2, u32 a3, const u64 (&key)[4], u32 (&lt)[4])
{
  if ((u32) J >= mx) return; /* uniform */
  nway_rank_pair<J> (a0, a1, a2, a3, key, lt);
  if constexpr (J + 2 < NWAY_LIMIT) nway_rank_steps<J + 2> (mx, a0, a1, a2, a3, key, lt);
}

__device__ __forceinline__ u64 readlane_u64 (u64 v, int l)
{
  return (u64) (u32) __builtin_amdgcn_readlane ((int) (u32) v, l) | ((u64) (u32) __builtin_amdgcn_readlane ((int) (u32) (v >> 32), l) << 32);
}

/* (diagnostics, off by default: GT4_NWAY_INJ_{VALU,SALU,LDS} dummy instructions per wavefront and tile at point
 * GT4_NWAY_INJ_AT -- what a saturated unit charges for them is how the tile's time is told apart) */
#ifndef GT4_NWAY_INJ_VALU
#define GT4_NWAY_INJ_VALU 0
#endif
/* (diagnostics, results INVALID: bit mask of LDS access classes that go to lane-linear, conflict-free addresses instead of
 * their own -- 1 bucket-count atomics, 2 grouped-key stores, 4 bucket walks, 8 fold atomics, 16 bitmap claims, 32 staging
 * stores, 64 leaders' count reads -- to tell which of them the LDS bank conflicts belong to) */
#ifndef GT4_NWAY_NOCONF
#define GT4_NWAY_NOCONF 0
#endif
#ifndef GT4_NWAY_INJ_SALU
#define GT4_NWAY_INJ_SALU 0
#endif
#ifndef GT4_NWAY_INJ_LDS
#define GT4_NWAY_INJ_LDS 0
#endif
#ifndef GT4_NWAY_INJ_AT
#define GT4_NWAY_INJ_AT 0
#endif
template <int AT>
__device__ __forceinline__ void nway_inject (u32 lds_addr)
{
  if constexpr (AT == GT4_NWAY_INJ_AT) {
    if constexpr (GT4_NWAY_INJ_VALU > 0) {
      u32 a = 0, b = 1, c = 2, d = 3;
#pragma unroll
      for (int i = 0; i < GT4_NWAY_INJ_VALU / 4; i++)
        asm volatile ("v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 1\n\tv_add_u32 %2, %2, 1\n\tv_add_u32 %3, %3, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    if constexpr (GT4_NWAY_INJ_SALU > 0) {
      u32 a = 0, b = 1;
#pragma unroll
      for (int i = 0; i < GT4_NWAY_INJ_SALU / 2; i++) asm volatile ("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1" : "+s"(a), "+s"(b) : : "scc");
    }
    if constexpr (GT4_NWAY_INJ_LDS > 0) {
      u32 r;
#pragma unroll
      for (int i = 0; i < GT4_NWAY_INJ_LDS; i++) asm volatile ("ds_read_b32 %0, %1" : "=v"(r) : "v"(lds_addr) : "memory");
      asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

template <int NT, int RPT, int NBF, int MODE>
__global__ __launch_bounds__ (NT, nway_waves_per_simd (NT)) void
k_nway_merge (NwayParams p, const u64 *__restrict__ part, u32 *__restrict__ out, u64 *desc, PairControl *ctl)
{
  typedef NwayShared<NT, RPT, NBF, MODE> Shared;
  constexpr int CAP = Shared::CAP, NW = Shared::NW, NCH = Shared::NCH, NB = Shared::NB;
  constexpr int NWORDS = NB / 2, WPT = NWORDS / NT;
  constexpr int CAPS = Shared::CAPS;
  constexpr bool LEAD = Shared::LEAD;
  static_assert (NCH <= 2 * WAVE, "one lane per wave slot builds the slot table, in two rounds at most");
  static_assert (WPT * NT == NWORDS && WPT >= 1, "every thread scans the same number of counter words");
  static_assert (NW <= 16 && NW >= 2, "wave totals are reduced by one DPP row");
  static_assert (NWAY_LIMIT % 2 == 0 && NWAY_TRY0 <= NWAY_LIMIT, "bucket walks go two steps at a time");
  static_assert (CAP <= 32767 && NB <= 65536, "16-bit bucket counters and starts; bucket, arrival number and a flag share a dword");
  static_assert (NWAY_MAX > 8 || 2 * NWAY_PSTRIDE <= WAVE, "one lane per partition entry of a tile");
  static_assert (NWAY_PSTRIDE <= WAVE, "many lists: one lane per entry of ONE partition row");
  __shared__ Shared sh;
  int tid = threadIdx.x, lane = tid & (WAVE - 1); /* (not const: see the top of the tile loop) */
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
      /* one wavefront sums and chains the rows while the launch has few enough of them (one hop less
       * between a tile's total and its successors' offsets: with a single staging area the chain's
       * latency bounds the time per tile); summers + chainer beyond that */
      const u32 n_sub = p.scan_group ? (NW < 8 ? (u32) NW : 8u) : 1u;
      if ((u32) wid < n_sub) scanner_part (agg, carry + 4 * (n_rows + 1), carry, p.num_tiles, ctl, lane, spin_limit, (u32) wid, n_sub);
      return;
    }
  }
  const u32 n_workers = MODE == NWAY_UNION ? gridDim.x - 1 : gridDim.x;
  const u32 wk = MODE == NWAY_UNION ? role - 1 : blockIdx.x;
  const u32 ntl = p.num_tiles;
  /* The LAST wavefront is the service wavefront.  Wave slots are dealt to the wavefronts in order (RPT
   * consecutive slots each), a tile fills 85 % of them on average, so the last wavefront usually has
   * no records and does what must not sit in front of everybody's barrier: tile numbers (by ticket or
   * round-robin), partition entries, the slot table two tiles ahead (from registers: no LDS round
   * trips), the chain words of the tile being written out, the publication of the tile total. */
  const bool service = wid == NW - 1;
  /* the SIMD issues oldest-first and the last wavefront is the youngest of its SIMD: without a raised
   * priority its few instructions crawl behind three ranking wavefronts (measured: 5.7 k cycles for the
   * slot table alone) and everybody waits for it at the next barrier */
  if (service) __builtin_amdgcn_s_setprio (GT4_NWAY_SVPRIO);
  auto deal = [&] (int j) -> u32 { /* lane 0 of the service wavefront */
    if (p.dynamic) {
      const u32 t = atomicAdd (&ctl->ticket, 1u);
      return t < ntl ? t : 0xffffffffu;
    }
    const u64 t = (u64) wk + (u64) j * n_workers;
    return t < (u64) ntl ? (u32) t : 0xffffffffu;
  };
#if GT4_KM > 8
  /* (many lists: a partition row is KM + 2 entries -- one lane per entry of the tile's START row; the END row is the
   * next tile's start row, asked for with it) */
  struct RowPair { u64 a, b; };
  auto load_row = [&] (u32 tile) -> RowPair {
    RowPair v = { 0, 0 };
    if (tile < ntl && lane < NWAY_PSTRIDE) {
      v.a = __hip_atomic_load (&part[(u64) tile * NWAY_PSTRIDE + (u64) lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v.b = __hip_atomic_load (&part[((u64) tile + 1) * NWAY_PSTRIDE + (u64) lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return v;
  };
  /* the table of `tile`: the runs end to end in the position space, a mask of run starts per stretch of 32 positions */
  auto build_table = [&] (RowPair row, u32 tile, int tb) {
    if (tile >= ntl) {
      if (lane == 0) sh.hdr[tb][0] = 0xffffffffu;
      return;
    }
    const u64 lbv = (u32) lane < p.k ? (u64) p.list[lane < NWAY_MAX ? lane : 0] : 0ull; /* lane q: base address of list q */
    const u32 len = (u32) lane < p.k ? (u32) row.b - (u32) row.a : 0u;
    const u32 incl = dpp_inclusive_scan_u32 (len), excl = incl - len; /* excl: first position of the run */
    const u32 n = (u32) __builtin_amdgcn_readlane ((int) incl, WAVE - 1);
    if (lane < NWAY_MAX) {
      sh.tab_pbase[tb][lane] = excl;
      sh.tab_len[tb][lane] = len;
    }
    /* non-empty runs, in order */
    const u64 nz = __builtin_amdgcn_ballot_w64 (len != 0u);
    const u32 rank = __builtin_amdgcn_mbcnt_hi ((u32) (nz >> 32), __builtin_amdgcn_mbcnt_lo ((u32) nz, 0u));
    u32 *const hm32 = reinterpret_cast<u32 *> (&sh.hmask[tb][0]);
    hm32[2 * lane] = 0;                /* (the stretches' masks; 2 x 64 = NHM of them) */
    hm32[2 * (lane + WAVE)] = 0;
    asm volatile ("" ::: "memory");
    if (len) {
      sh.rtab[tb][rank] = lbv + 12ull * row.a - 12ull * (u64) excl;
      sh.rlist[tb][rank] = (u32) lane;
      atomicOr (&hm32[2 * (excl >> 5)], 1u << (excl & 31u));
    }
    asm volatile ("" ::: "memory");
    {
      const u32 m0 = hm32[2 * lane], m1 = hm32[2 * (lane + WAVE)];
      const u32 c0 = (u32) __popc (m0), c1 = (u32) __popc (m1);
      const u32 i0 = dpp_inclusive_scan_u32 (c0);
      const u32 t0 = (u32) __builtin_amdgcn_readlane ((int) i0, WAVE - 1);
      const u32 i1 = dpp_inclusive_scan_u32 (c1);
      hm32[2 * lane + 1] = i0 - c0;
      hm32[2 * (lane + WAVE) + 1] = t0 + i1 - c1;
    }
    u64 base = 0;
    /* the records in front of the tile: where a level of merged samples starts its output, and a count table the tile's rows */
    if (MODE == NWAY_DUPS || MODE == NWAY_TABLE) base = wave_sum ((u32) lane < p.k ? row.a : 0ull);
    if (MODE == NWAY_PROBE) base = readlane_u64 (row.a, 0); /* the tile's first record of list 0 = its first row */
    const u32 rlo = (u32) row.a, rhi = (u32) (row.a >> 32);
    const u32 lo_lo = (u32) __builtin_amdgcn_readlane ((int) rlo, NWAY_MAX), lo_hi = (u32) __builtin_amdgcn_readlane ((int) rhi, NWAY_MAX);
    const u32 bk_lo = (u32) __builtin_amdgcn_readlane ((int) rlo, NWAY_MAX + 1), bk_hi = (u32) __builtin_amdgcn_readlane ((int) rhi, NWAY_MAX + 1);
    /* the longest run (the lowest list among equals): one wave maximum over length << 6 | 63 - lane */
    const u32 best = dpp_wave_max_u32 ((len << 6) | (63u - (u32) lane));
    const u32 pl = 63u - (best & 63u);
    const u32 pv_len = best >> 6, pv_base = (u32) __builtin_amdgcn_readlane ((int) excl, (int) pl);
    u32 h = tile;
    h = lane == 1 ? n : h;
    h = lane == 2 ? (n + (u32) WAVE - 1u) / (u32) WAVE : h; /* wave slots */
    h = lane == 3 ? bk_lo : h;
    h = lane == 4 ? bk_hi : h;
    h = lane == 5 ? lo_lo : h;
    h = lane == 6 ? lo_hi : h;
    h = lane == 7 ? (u32) base : h;
    h = lane == 8 ? (u32) (base >> 32) : h;
    h = lane == 9 ? pv_base : h;
    h = lane == 10 ? pv_len : h;
    if (lane < 11) sh.hdr[tb][lane] = h;
  };
#else
  auto load_row = [&] (u32 tile) -> u64 { /* lane i: entry i of the tile's two partition rows (its start and its end) */
    u64 v = 0;
    if (tile < ntl && lane < 2 * NWAY_PSTRIDE)
      v = __hip_atomic_load (&part[(u64) tile * NWAY_PSTRIDE + (u64) lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
  };
  /* the slot table of `tile` (partition entries in `row`, one per lane) into table tb: branch-free
   * vector code -- run lengths by a lane shift, their prefix by a DPP scan, the run of a slot by
   * eight compares against broadcast prefixes, the run's data by lane permutes */
  auto build_table = [&] (u64 row, u32 tile, int tb) {
    if (tile >= ntl) {
      if (lane == 0) sh.hdr[tb][0] = 0xffffffffu;
      return;
    }
    u64 lbv = 0; /* lane q: base address of list q (rebuilt here: two registers less in every wavefront's loop) */
#pragma unroll
    for (int m = 0; m < NWAY_MAX; m++) lbv = lane == m ? (u64) p.list[m] : lbv;
    const u32 rlo = (u32) row, rhi = (u32) (row >> 32);
    const u32 elo = __shfl_down (rlo, NWAY_PSTRIDE, WAVE);
    const u32 len = (u32) lane < p.k ? elo - rlo : 0u; /* (p.k <= 8; a run is shorter than 2^32 records) */
    const u32 nw = (len + WAVE - 1) / WAVE;
    const u32 incl = dpp_inclusive_scan_u32 (nw), excl = incl - nw;
    const u32 total = (u32) __builtin_amdgcn_readlane ((int) incl, WAVE - 1);
    const u32 n = dpp_wave_sum_u32 (len);
    if (lane < NWAY_MAX) {
      sh.tab_pbase[tb][lane] = excl * WAVE;
      sh.tab_len[tb][lane] = len;
    }
#pragma unroll
    for (int sb = 0; sb < NCH; sb += WAVE) { /* one lane per wave slot (two rounds when a tile has more than 64) */
      const u32 slot = (u32) (sb + lane);
      u32 run = 0;
#pragma unroll
      for (int q = 0; q < NWAY_MAX - 1; q++) run += slot >= (u32) __builtin_amdgcn_readlane ((int) incl, q) ? 1u : 0u;
      const u32 len_r = __shfl (len, run, WAVE), excl_r = __shfl (excl, run, WAVE);
      const u64 s_r = (u64) __shfl (rlo, run, WAVE) | ((u64) __shfl (rhi, run, WAVE) << 32);
      const u64 lb_r = (u64) __shfl ((u32) lbv, run, WAVE) | ((u64) __shfl ((u32) (lbv >> 32), run, WAVE) << 32);
      const bool in = slot < total;
      const u32 first = in ? (slot - excl_r) * WAVE : 0u;
      if (slot < (u32) NCH) {
        sh.slot_cnt[tb][slot] = in ? (len_r - first < (u32) WAVE ? len_r - first : (u32) WAVE) : 0u;
        sh.slot_addr[tb][slot] = lb_r + 12ull * (s_r + first);
        if (MODE == NWAY_TABLE || MODE == NWAY_PROBE || MODE == NWAY_DUPS) sh.slot_run[tb][slot] = run;
      }
    }
    u64 base = 0;
    if (MODE == NWAY_DUPS || MODE == NWAY_TABLE) {
      /* the records in front of the tile: where a level of merged samples starts the tile's output -- and where
       * the count table starts the tile's rows (a tile has at most as many distinct keys as records: the table is
       * RAGGED, see gt4hip_count_table) */
#pragma unroll
      for (int q = 0; q < NWAY_MAX; q++) base += (u32) q < p.k ? readlane_u64 (row, q) : 0ull;
    }
    if (MODE == NWAY_PROBE) base = readlane_u64 (row, 0); /* the tile's first record of list 0 = its first row */
    const u32 lo_lo = (u32) __builtin_amdgcn_readlane ((int) rlo, NWAY_MAX), lo_hi = (u32) __builtin_amdgcn_readlane ((int) rhi, NWAY_MAX);
    const u32 bk_lo = (u32) __builtin_amdgcn_readlane ((int) rlo, NWAY_MAX + 1), bk_hi = (u32) __builtin_amdgcn_readlane ((int) rhi, NWAY_MAX + 1);
    u32 h = tile;
    h = lane == 1 ? n : h;
    h = lane == 2 ? total : h;
    h = lane == 3 ? bk_lo : h;
    h = lane == 4 ? bk_hi : h;
    h = lane == 5 ? lo_lo : h;
    h = lane == 6 ? lo_hi : h;
    h = lane == 7 ? (u32) base : h;
    h = lane == 8 ? (u32) (base >> 32) : h;
    /* the longest run: the pivot of the second bucketing attempt (first position, records) */
    u32 pv_len = 0, pv_base = 0;
#pragma unroll
    for (int q = 0; q < NWAY_MAX; q++) {
      const u32 lq = (u32) __builtin_amdgcn_readlane ((int) len, q), bq = (u32) __builtin_amdgcn_readlane ((int) excl, q) * WAVE;
      const bool better = lq > pv_len; /* uniform */
      pv_base = better ? bq : pv_base;
      pv_len = better ? lq : pv_len;
    }
    h = lane == 9 ? pv_base : h;
    h = lane == 10 ? pv_len : h;
    if (lane < 11) sh.hdr[tb][lane] = h;
  };

#endif /* GT4_KM > 8 */

  /* The tile's records, fetched one tile ahead into registers.  A wavefront fetches 64 consecutive
   * records of ONE run per instruction, so descriptor and addresses are scalar and the range-checked
   * descriptor zero-fills past the run's end: no per-lane bounds. */
  u32x3 pre[RPT];
#if GT4_KM > 8
  /* (many lists: position p = 64 x chunk + lane of the tile is a record of the run whose start is the last one at or
   * in front of p -- the stretch's mask of run starts and the number of starts in front of the stretch come in ONE
   * 64-bit LDS read, the run's address entry in another; positions behind the tile's last record are not loaded) */
  auto run_of = [&] (int tb, u32 pos) -> u32 { /* the non-empty run position `pos` belongs to, counted from 0 */
    const u64 hm = sh.hmask[tb][pos >> 5];
    return (u32) (hm >> 32) + (u32) __popc ((u32) hm & ((2u << (pos & 31u)) - 1u)) - 1u;
  };
  auto fetch = [&] (int tb) {
    const u32 n_t = uniform32 (sh.hdr[tb][1]);
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const u32 pos = (u32) (wid * RPT + k) * WAVE + (u32) lane;
      if (pos < n_t) {
        const u64 addr = sh.rtab[tb][run_of (tb, pos)] + 12ull * pos;
        pre[k] = *reinterpret_cast<const u32x3 *> (addr); /* (plain: non-temporal per-lane loads measured 0.4 % slower, r5_cache_policy.log) */
      }
    }
  };
#else
  auto fetch = [&] (int tb) {
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int chunk = wid * RPT + k;
      const u64 addr = uniform64 (sh.slot_addr[tb][chunk]);
      const u32 c = uniform32 (sh.slot_cnt[tb][chunk]);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc ((void *) addr, 0, (int) (12 * c), 0x00020000);
      /* (non-temporal where the union streams its records; the count tables' launches measured 9 % SLOWER with it -- 6.2
       * against 5.7 ms on six lists -- and keep plain loads: profiles/round5/r5_cache_policy.log) */
      pre[k] = __builtin_amdgcn_raw_buffer_load_b96 (rs, 12 * lane, 0, (MODE == NWAY_TABLE || MODE == NWAY_PROBE) ? 0 : GT4_LOAD_AUX);
    }
  };
#endif

  /* ---- prologue: tiles of iterations 0 .. 3, tables of the first two, entries of the third */
  u32 sv_t2 = 0xffffffffu; /* service wavefront: tile of iteration it + 2 (uniform) */
  u32 sv_tk = 0xffffffffu; /* ... of iteration it + 3, in lane 0 (a ticket drawn one iteration ago) */
#if GT4_KM > 8
  RowPair sv_row = { 0, 0 }; /* partition entries of tile sv_t2, asked for one iteration ago */
#else
  u64 sv_row = 0;          /* partition entries of tile sv_t2, asked for one iteration ago */
#endif
  if (service) {
    u32 d[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };
    if (lane == 0)
      for (int q = 0; q < 4; q++) d[q] = deal (q);
    const u32 d0 = uniform32 (d[0]), d1 = uniform32 (d[1]);
    sv_t2 = uniform32 (d[2]);
    sv_tk = d[3];
    const auto r0 = load_row (d0), r1 = load_row (d1);
    sv_row = load_row (sv_t2);
    build_table (r0, d0, 0);
    build_table (r1, d1, 1);
  }
#pragma unroll
  for (int i = 0; i < WPT; i++) sh.cnt[tid * WPT + i] = 0;
  auto fill_g = [&] () { /* all-ones wherever no key is: what a bucket walk meets behind its bucket must not be smaller than any key */
    static_assert (Shared::GSZ % 2 == 0, "the grouped keys are filled 16 bytes at a time");
#pragma unroll
    for (int r = 0; r < Shared::GSZ / 2 / NT; r++) *reinterpret_cast<u32x4 *> (&sh.g ()[2 * (r * NT + tid)]) = u32x4 { ~0u, ~0u, ~0u, ~0u };
    if (tid < Shared::GSZ / 2 - Shared::GSZ / 2 / NT * NT) *reinterpret_cast<u32x4 *> (&sh.g ()[2 * (Shared::GSZ / 2 / NT * NT + tid)]) = u32x4 { ~0u, ~0u, ~0u, ~0u };
  };
  if (GT4_NWAY_FILL) fill_g ();
  if (MODE == NWAY_TABLE || MODE == NWAY_PROBE) /* the row area of the count tables starts as zeros (see table_rows) */
    for (int c = 4 * tid; c < Shared::ROWW + 4; c += 4 * NT) *reinterpret_cast<u32x4 *> (&sh.stage[c]) = u32x4 { 0, 0, 0, 0 };
  __syncthreads ();
  if (uniform32 (sh.hdr[0][0]) < ntl && (u32) (wid * RPT) < uniform32 (sh.hdr[0][2])) fetch (0);

  u64 acc_sum = 0; /* per-thread sum of kept counts */
  u64 blk_cnt = 0; /* records kept (the same in every thread) */
  u32 pend_tot = 0, pend_tile = 0;
  u64 pend_base = 0;
  bool pend = false;
  int it = 0;
  int tb = 0, tb1 = 1, tb2 = 2; /* tables of this tile, the next, the one after */
  /* the list record k of this thread comes from (many lists: per lane; else the same for the whole wave slot) */
  auto list_of = [&] (int k) -> u32 {
#if GT4_KM > 8
    return sh.rlist[tb][run_of (tb, (u32) (wid * RPT + k) * WAVE + (u32) lane)];
#else
    return uniform32 (sh.slot_run[tb][wid * RPT + k]);
#endif
  };
  /* The count tables' rows leave through LDS (round 5): as many of the tile's rows as fit ROWW words are zeroed there, the
   * records drop their counts in, and the rows go out whole, 16 bytes per lane -- instead of zeros stored to global memory
   * and 4-byte stores scattered over them (32 columns: 67 GB written for 46 GB of table, half the rows went to HBM twice).
   * row[k]: the tile's row record k belongs in (anything >= n_rows: none). */
  auto table_rows = [&] (const u32 (&row)[RPT], const u32 (&val)[RPT], u32 n_rows, u64 first_row) {
    const u32 cols = p.table_cols;
    const u32 rb = (u32) Shared::ROWW / cols;
    u32 col[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) col[k] = p.table_col[list_of (k)];
    /* (the row area is all zeros here: zeroed once in front of the first tile, and every thread zeroes the 16 bytes it has
     * just sent out -- no zeroing pass and no barrier in front of the records' stores) */
    for (u32 r0 = 0; r0 < n_rows; r0 += rb) {
      const u32 nr = n_rows - r0 < rb ? n_rows - r0 : rb, words = nr * cols;
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const u32 rr = row[k] - r0;
        if (rr < nr) sh.stage[rr * cols + col[k]] = val[k];
      }
      __syncthreads ();
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc ((void *) (p.table_counts + (first_row + r0) * cols), 0, (int) (4 * words), 0x00020000);
      for (u32 c = (u32) tid; 4u * c < words; c += NT) {
        u32x4 *const q = reinterpret_cast<u32x4 *> (&sh.stage[4u * c]);
        __builtin_amdgcn_raw_buffer_store_b128 (*q, rs, 16 * c, 0, GT4_TABLE_STORE_AUX);
        *q = u32x4 { 0, 0, 0, 0 };
      }
      if (r0 + rb < n_rows) __syncthreads (); /* (the next rows' counts go where these lay) */
    }
  };
#ifdef GT4_PROFILE_PHASES
  u64 ph[24];
  for (int i = 0; i < 24; i++) ph[i] = 0;
  u64 t_last;
  asm volatile ("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last) :: "memory");
#endif

  for (;;) {
    /* the thread number is made opaque once per tile: addresses and masks derived from it are then
     * recomputed where they are used (a few VALU each) instead of living in ~25 registers across the
     * whole loop (hoisted by the compiler), which had the rank walk's registers spill */
    asm volatile ("" : "+v"(tid));
    lane = tid & (WAVE - 1);
    PHASE_STAMP (23); /* (diagnostics: the back edge) */
    u32 cur, n, slots, bk0, bk_mul;
    u64 key_lo, out_base;
    {
      const u32x4 h0 = *reinterpret_cast<const u32x4 *> (&sh.hdr[tb][0]), h1 = *reinterpret_cast<const u32x4 *> (&sh.hdr[tb][4]);
      const u32 h8 = sh.hdr[tb][8];
      cur = uniform32 (h0.x);
      n = uniform32 (h0.y);
      slots = uniform32 (h0.z);
      bk0 = uniform32 (h0.w);
      bk_mul = uniform32 (h1.x);
      key_lo = (u64) uniform32 (h1.y) | ((u64) uniform32 (h1.z) << 32);
      out_base = (u64) uniform32 (h1.w) | ((u64) uniform32 (h8) << 32);
    }
    if (cur >= ntl) break;
    if (n > (u32) CAP || slots > (u32) NCH) {
      if (tid == 0) atomicOr (&ctl->error, 2u);
      break;
    }
    const u32 bk_sh = bk0 & 0xffu;
    const bool bk_direct = (bk0 >> 8) & 1u;
    const bool has_rec = (u32) (wid * RPT) < slots;          /* this wavefront holds records of the tile */
    const bool has_pos = (u32) (wid * RPT * WAVE) < n;       /* ... positions of the ordered tile */

    /* ---- phase 0: the prefetched records leave the fetch registers */
    u64 key[RPT];
    u32 cnt[RPT], ba[RPT]; /* ba: bucket | arrival number << 16 | valid << 31 */
#pragma unroll
    for (int k = 0; k < RPT; k++) ba[k] = 0;
    /* (a wavefront without records of this tile keeps whatever the registers hold: its lanes are not valid) */
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      key[k] = (u64) pre[k].x | ((u64) pre[k].y << 32);
      cnt[k] = pre[k].z;
    }
    PHASE_STAMP (19); /* (diagnostics: the tile's header) */
    nway_inject<0> (lds_offset (&sh.hdr[0][0]) + 4u * (u32) lane);
#ifdef GT4_PROFILE_PHASES
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory"); /* (the diagnostics build takes the wait for the prefetched records here) */
#endif
    PHASE_STAMP (20); /* (diagnostics: the wait for the prefetched records) */
    u32 xagg = 0;
    u64 xcarry = 0;
    u32 st[RPT];
    u32 mx = 0;
    bool accepted = false; /* the buckets of the last attempt are walked (else: the search path) */
    /* one bucketing pass over bucket numbers bk[]: count (arrival numbers), scan, group the keys */
    auto count_pass = [&] (const u32 (&bk)[RPT]) {
      if (has_rec) {
        u32 c[RPT], old[RPT];
#if GT4_KM > 8
        const u32 lane_ = (u32) lane; /* (many lists: a position holds a record iff it lies in front of the tile's end) */
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          const u32 first = (u32) (wid * RPT + k) * WAVE;
          c[k] = n > first ? n - first : 0u;
        }
#else
        const u32 lane_ = (u32) lane;
        if constexpr (RPT == 4) { /* (the wavefront's four slots: one 16-byte read; compared per lane, no scalar copy) */
          const u32x4 c4 = *reinterpret_cast<const u32x4 *> (&sh.slot_cnt[tb][wid * RPT]);
          c[0] = c4.x, c[1] = c4.y, c[2] = c4.z, c[3] = c4.w;
        } else {
#pragma unroll
          for (int k = 0; k < RPT; k++) c[k] = sh.slot_cnt[tb][wid * RPT + k];
        }
#endif
#pragma unroll
        for (int k = 0; k < RPT; k++) { /* the atomics one behind the other: one wait for all of them */
          const u32 b = bk[k] < (u32) NB ? bk[k] : (u32) NB - 1u;
          const u32 vm = lane_ < c[k] ? ~0u : 0u;
          old[k] = atomicAdd (&sh.cnt[(GT4_NWAY_NOCONF & 1) ? (u32) (NB / 2 + 4) + (u32) lane : nway_pick (vm, b >> 1, (u32) (NB / 2 + 4) + (u32) lane)], 1u << ((b & 1u) * 16u));
        }
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          const u32 b = bk[k] < (u32) NB ? bk[k] : (u32) NB - 1u;
          ba[k] = lane_ < c[k] ? b | (((old[k] >> ((b & 1u) * 16u)) & 0x7fffu) << 16) | 0x80000000u : 0u;
        }
      }
    };
    auto bucket_pass = [&] (u32 limit) {
      PHASE_STAMP (0);
      __syncthreads (); /* B1: every record is counted */
      PHASE_STAMP (1);

      if constexpr (GT4_NWAY_SCAN4 && NWORDS % (4 * WAVE * 4) == 0 && NW >= 4) {
        /* ---- scan of the bucket counters by the first four wavefronts (one per SIMD: a workgroup's wavefronts go to
         * the SIMDs in turn): WPL words = 2 WPL counters per lane, read and written 16 bytes at a time */
        constexpr int WPL = NWORDS / (4 * WAVE);
        u32 ex[2 * WPL];
        u32 tsum = 0, incl = 0;
        const int sl = (wid & 3) * WAVE + lane; /* the lane's place among the 256 scanning lanes */
        if (wid < 4) {
          u32 tmax = 0;
          u32 w[WPL];
#pragma unroll
          for (int i = 0; i < WPL / 4; i++) {
            const u32x4 q = *reinterpret_cast<const u32x4 *> (&sh.cnt[sl * WPL + 4 * i]);
            w[4 * i] = q.x;
            w[4 * i + 1] = q.y;
            w[4 * i + 2] = q.z;
            w[4 * i + 3] = q.w;
          }
#pragma unroll
          for (int i = 0; i < WPL; i++) {
            const u32 a = w[i] & 0xffffu, b = w[i] >> 16;
            ex[2 * i] = tsum;
            tsum += a;
            ex[2 * i + 1] = tsum;
            tsum += b;
            tmax = a > tmax ? a : tmax;
            tmax = b > tmax ? b : tmax;
          }
          incl = dpp_inclusive_scan_u32 (tsum);
          const u32 wmx = dpp_wave_max_u32 (tmax);
          if (lane == WAVE - 1) {
            sh.wtot[wid] = incl;
            sh.wmax[wid] = wmx;
          }
        }
        PHASE_STAMP (2);
        __syncthreads (); /* B2: the four wavefronts' totals */
        PHASE_STAMP (3);
        if (wid < 4) {
          const u32x4 t4 = *reinterpret_cast<const u32x4 *> (&sh.wtot[0]);
          const u32 wbase = (wid > 0 ? t4.x : 0u) + (wid > 1 ? t4.y : 0u) + (wid > 2 ? t4.z : 0u); /* (wid is uniform: scalar selects) */
          const u32 tbase = wbase + incl - tsum;
#pragma unroll
          for (int i = 0; i < WPL / 4; i++) {
            u32x4 q;
            q.x = (tbase + ex[8 * i]) | ((tbase + ex[8 * i + 1]) << 16);
            q.y = (tbase + ex[8 * i + 2]) | ((tbase + ex[8 * i + 3]) << 16);
            q.z = (tbase + ex[8 * i + 4]) | ((tbase + ex[8 * i + 5]) << 16);
            q.w = (tbase + ex[8 * i + 6]) | ((tbase + ex[8 * i + 7]) << 16);
            *reinterpret_cast<u32x4 *> (&sh.cnt[sl * WPL + 4 * i]) = q;
          }
          if (sl == 4 * WAVE - 1) sh.cnt[NWORDS] = tbase + tsum; /* start of the bucket behind the last = the tile's records */
          if (!GT4_NWAY_FILL) {
#pragma unroll
            for (int j = 0; j < 2 * WPL; j++) {
              const u32 s0 = tbase + ex[j], e0 = tbase + (j + 1 < 2 * WPL ? ex[j + 1 < 2 * WPL ? j + 1 : 0] : tsum);
              if ((e0 >> 5) != (s0 >> 5)) {
                sh.g ()[e0 + (s0 >> 5)] = ~0ull;
                if ((e0 >> 5) - (s0 >> 5) > 1u) sh.g ()[e0 + (s0 >> 5) + 1u] = ~0ull;
              }
            }
          }
        }
        if (!GT4_NWAY_FILL && tid < NWAY_LIMIT + 2) sh.g ()[nway_skew (n) + (u32) tid] = ~0ull;
      } else {
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
      PHASE_STAMP (2);
      __syncthreads (); /* B2: wave totals */
      PHASE_STAMP (3);
      {
        const u32 x = lane < NW ? sh.wtot[lane] : 0u;
        const u32 y = lane < NW ? sh.wmax[lane] : 0u;
        const u32 wbase = dpp_wave_sum_u32 (lane < wid ? x : 0u);
        mx = dpp_wave_max_u32 (y);
        const u32 tbase = wbase + incl - tsum;
#pragma unroll
        for (int i = 0; i < WPT; i++) sh.cnt[tid * WPT + i] = (tbase + ex[2 * i]) | ((tbase + ex[2 * i + 1]) << 16);
        if (tid == NT - 1) sh.cnt[NWORDS] = tbase + tsum; /* start of the bucket behind the last = the tile's records */
        /* What a bucket walk meets behind its bucket must not be smaller than any key: the following
         * buckets' keys are not, and the skewed slots a bucket that crosses multiples of 32 leaves free
         * behind its last key (two at most: walked buckets hold no more than 48 keys) get all-ones here,
         * as do the slots behind the tile's last key */
        if (!GT4_NWAY_FILL) {
#pragma unroll
          for (int j = 0; j < 2 * WPT; j++) {
            const u32 s0 = tbase + ex[j], e0 = tbase + (j + 1 < 2 * WPT ? ex[j + 1 < 2 * WPT ? j + 1 : 0] : tsum);
            if ((e0 >> 5) != (s0 >> 5)) {
              sh.g ()[e0 + (s0 >> 5)] = ~0ull;
              if ((e0 >> 5) - (s0 >> 5) > 1u) sh.g ()[e0 + (s0 >> 5) + 1u] = ~0ull;
            }
          }
          if (tid < NWAY_LIMIT + 2) sh.g ()[nway_skew (n) + (u32) tid] = ~0ull; /* (+2: the walks read two steps ahead) */
        }
      }
      }
      PHASE_STAMP (4);
      __syncthreads (); /* B3: bucket starts */
      PHASE_STAMP (5);
      if constexpr (GT4_NWAY_SCAN4 && NWORDS % (4 * WAVE * 4) == 0 && NW >= 4) {
        const u32x4 m4 = *reinterpret_cast<const u32x4 *> (&sh.wmax[0]);
        const u32 m01 = m4.x > m4.y ? m4.x : m4.y, m23 = m4.z > m4.w ? m4.z : m4.w;
        mx = uniform32 (m01 > m23 ? m01 : m23);
      }

      /* ---- the keys grouped by bucket */
#pragma unroll
      for (int k = 0; k < RPT; k++) st[k] = 0;
      accepted = mx <= limit;
      if (has_rec && accepted) {
        u32 w0[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) w0[k] = sh.cnt[(ba[k] & 0xffffu) >> 1]; /* (word 0 where no record is) */
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          const u32 b = ba[k] & 0xffffu;
          const u32 s = (b & 1u) ? w0[k] >> 16 : w0[k]; /* start of the bucket */
          const u32 vm = nway_valid_mask (ba[k]);
          st[k] = s & 0xffffu & vm;
          sh.g ()[(GT4_NWAY_NOCONF & 2) ? (u32) Shared::GSZ + (u32) lane : nway_pick (vm, nway_skew (st[k]) + ((ba[k] >> 16) & 0x7fffu), (u32) Shared::GSZ + (u32) lane)] = key[k]; /* a bucket's keys stay together */
        }
      }
      PHASE_STAMP (6);
      __syncthreads (); /* B4: keys grouped */
      PHASE_STAMP (7);
#pragma unroll
      for (int i = 0; i < WPT; i++) sh.cnt[tid * WPT + i] = 0; /* the next pass's / the next tile's counters */
    };

    /* ---- buckets.  First by interpolation inside the tile's key range (no search at all).  If a bucket
     * then holds more than NWAY_TRY0 keys -- clustered keys: stretches of adjacent keys with wide gaps
     * put a whole stretch into one bucket -- the tile is bucketed again, by RANK IN ITS LONGEST RUN (one
     * binary search per record in that run's keys, copied to LDS): a bucket then holds what the other
     * runs have between two neighbours of the pivot run, whatever the keys' values.  Only a tile that
     * defeats that too (more than NWAY_LIMIT keys in a bucket) takes the full search path below. */
    const bool pivot_first = ((bk0 >> 9) & 1u) && p.force_fallback == 0; /* the partition found the tile's samples clustered */
    auto pivot_buckets = [&] (u32 (&bk)[RPT]) {
      const u32 pv_base = uniform32 (sh.hdr[tb][9]), pv_len = uniform32 (sh.hdr[tb][10]);
      /* the pivot run's keys to LDS, in order (its records sit at positions pv_base ..; sh.s.skey is free until the fold) */
      if (has_rec) {
#pragma unroll
        for (int k = 0; k < RPT; k++) {
#if GT4_KM > 8
          const bool here = (u32) (wid * RPT + k) * WAVE + (u32) lane < n;
#else
          const bool here = (u32) lane < uniform32 (sh.slot_cnt[tb][wid * RPT + k]);
#endif
          const u32 q = (u32) (wid * RPT + k) * WAVE + (u32) lane - pv_base;
          if (here && q < pv_len) sh.s.skey[q] = key[k];
        }
      }
      __syncthreads (); /* pivot keys complete (and the counters are zero) */
      /* sub-buckets per gap between two pivot keys, by interpolation inside the gap (what lies between
       * two neighbours of the longest run is spread evenly far more often than the tile as a whole) */
      u32 sub_bits = 0;
      while (sub_bits < 3 && ((pv_len + 1u) << (sub_bits + 1)) <= (u32) NB) sub_bits++;
      const float sub_n = (float) (1u << sub_bits);
      u32 lb[RPT]; /* lower bounds in the pivot run: the searches of a thread's records in step */
#pragma unroll
      for (int k = 0; k < RPT; k++) lb[k] = 0;
      for (u32 h = 1u << (31 - __builtin_clz (pv_len | 1u)); h; h >>= 1) { /* uniform trip count */
        u64 pk[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) pk[k] = sh.s.skey[(lb[k] + h <= pv_len ? lb[k] + h : 1u) - 1u];
#pragma unroll
        for (int k = 0; k < RPT; k++) lb[k] = (lb[k] + h <= pv_len && pk[k] < key[k]) ? lb[k] + h : lb[k];
      }
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const u32 b = lb[k];
        /* key in (pk[b-1], pk[b]]: its place in the gap, monotone in the key (float conversions and
         * products by positive constants are), the same for equal keys */
        u32 sub = 0;
        if (b > 0 && b < pv_len) {
          const u64 lo_k = sh.s.skey[b - 1u], d = sh.s.skey[b] - lo_k, x = key[k] - lo_k;
          const u32 shf = d >> 24 ? 40u - (u32) __builtin_clzll (d) : 0u; /* d >> shf below 2^24: exact in a float */
          const float q = (float) (u32) (x >> shf) * sub_n / (float) ((u32) (d >> shf) + 1u);
          sub = (u32) q;
          sub = sub < (1u << sub_bits) ? sub : (1u << sub_bits) - 1u;
        }
        bk[k] = (b << sub_bits) + sub;
      }
      if (LEAD) { /* the pivot keys lay where the grouped keys go */
        __syncthreads ();
        fill_g ();
      }
    };
    {
      u32 bk[RPT];
      if (__builtin_expect (pivot_first, 0)) {
        pivot_buckets (bk);
      } else {
        /* (a tile whose shifted key range is below the number of buckets: v itself, less one -- the same multiply, no branch) */
        const u32 mul = bk_direct ? 0xffffffffu : bk_mul;
#pragma unroll
        for (int k = 0; k < RPT; k++) bk[k] = __umulhi ((u32) ((key[k] - key_lo) >> bk_sh), mul);
      }
      count_pass (bk); /* (the atomics first: their round trip overlaps what follows) */
    }
    PHASE_STAMP (21); /* (bucket numbers, counting atomics returned) */
    /* the next tile's records: asked for as soon as this tile's have left the registers, a whole
     * iteration before they are looked at (its table was written during the previous iteration).  One
     * tile per workgroup is all that is in flight. */
    {
      const u32 nxt = uniform32 (sh.hdr[tb1][0]);
      if (nxt < ntl && (u32) (wid * RPT) < uniform32 (sh.hdr[tb1][2])) {
        fetch (tb1);
      } else {
        /* nothing to fetch: the fetch registers may hold anything (said so, or the compiler keeps their old values alive
         * through this arm and copies all twelve on both arms: 24 moves per wavefront and tile) */
#pragma unroll
        for (int k = 0; k < RPT; k++) asm volatile ("" : "=v"(pre[k].x), "=v"(pre[k].y), "=v"(pre[k].z));
      }
    }
    PHASE_STAMP (22); /* (the next tile's fetch issued) */
    /* the ordered tile: counts 0, nothing live */
    {
      static_assert (CAPS % 4 == 0 && CAPS >= 4 * NT, "the ordered tile's counts are zeroed 16 bytes at a time: whole rounds and a part of one");
#pragma unroll
      for (int r = 0; r < CAPS / (4 * NT); r++) *reinterpret_cast<u32x4 *> (&sh.s.scnt[4 * (r * NT + tid)]) = u32x4 { 0, 0, 0, 0 };
      if (tid < (CAPS - CAPS / (4 * NT) * (4 * NT)) / 4) *reinterpret_cast<u32x4 *> (&sh.s.scnt[CAPS / (4 * NT) * (4 * NT) + 4 * tid]) = u32x4 { 0, 0, 0, 0 };
    }
    if (!LEAD) for (int i = tid; i < (CAPS + 3) / 4; i += NT) sh.live[i] = 0;
    if (LEAD) { /* (last read two tiles ago) */
      static_assert (!LEAD || Shared::LW % 4 == 0, "the bitmap is zeroed 16 bytes at a time");
      for (int i = tid; i < Shared::LW / 4; i += NT) *reinterpret_cast<u32x4 *> (&sh.lead[it & 1][4 * i]) = u32x4 { 0, 0, 0, 0 };
    }
    /* service: the chain words of the tile staged one iteration ago are asked for; they are looked at
     * behind B4 at the earliest (the memory counter retires in order: a look waits for every older
     * operation of this wavefront, the previous write-out's stores included) */
    if (service && MODE == NWAY_UNION && pend) {
      const u64 prow = pend_tile / WAVE;
      if ((u32) lane < pend_tile % WAVE) xagg = peek_u32 (&agg[prow * WAVE + lane]);
      xcarry = peek_u64 (&carry[prow]);
    }

    /* ---- buckets.  Attempt 0: by interpolation inside the tile's key range (no search at all).  If a
     * bucket holds more than NWAY_TRY0 keys -- clustered keys: stretches of adjacent keys with wide
     * gaps put a whole stretch into one bucket -- attempt 1 buckets by RANK IN THE TILE'S LONGEST RUN
     * (one binary search per record in that run's keys, copied to LDS): buckets then hold what the other
     * runs have between two neighbours of the pivot run, whatever the keys' values.  Only a tile that
     * defeats that too (more than NWAY_LIMIT keys in a bucket) takes the full search path below. */
    bucket_pass (p.force_fallback ? 0u : (pivot_first ? (u32) NWAY_LIMIT : (u32) NWAY_TRY0));
    if (__builtin_expect (!accepted, 0)) { /* (cold: laid out behind the loop; ONE test on the usual path) */
      if (p.force_fallback && mx == 0) {
        accepted = true; /* (an empty tile) */
      } else if (!pivot_first && p.force_fallback != 1) {
        u32 bk[RPT];
        pivot_buckets (bk);
        count_pass (bk);
        bucket_pass ((u32) NWAY_LIMIT);
      }
    }

    /* ---- position of every record = number of smaller keys in the tile */
    u32 pos[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) pos[k] = 0;
    nway_inject<1> (lds_offset (&sh.hdr[0][0]) + 4u * (u32) lane);
    if (__builtin_expect (accepted, 1)) {
      if (has_rec) {
        u32 lt[RPT], ga[RPT];
        const u32 g0 = lds_offset (&sh.g ()[0]);
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          lt[k] = 0;
          ga[k] = (GT4_NWAY_NOCONF & 4) ? g0 + 8u * (u32) lane : g0 + 8u * nway_skew (st[k]);
        }
        /* every lane runs the longest bucket's length (rounded up to even): behind its own bucket a lane
         * meets larger keys or all-ones */
#pragma unroll
        for (int q = 0; q + 4 <= RPT; q += 4)
          nway_rank_steps<0> (mx, ga[q], ga[q + 1], ga[q + 2], ga[q + 3], *reinterpret_cast<const u64 (*)[4]> (&key[q]), *reinterpret_cast<u32 (*)[4]> (&lt[q]));
        if constexpr (RPT % 4 != 0) { /* (a fifth position per thread: two steps at a time as well) */
          for (u32 j = 0; j < mx; j += 2) {
#pragma unroll
            for (int k = RPT / 4 * 4; k < RPT; k++) {
              const u64 r0 = lds_load<u64> (ga[k] + 8u * j), r1 = lds_load<u64> (ga[k] + 8u * j + 8u);
              lt[k] += (r0 < key[k] ? 1u : 0u) + (r1 < key[k] ? 1u : 0u);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < RPT; k++) pos[k] = st[k] + lt[k];
      }
    } else {
      /* clustered keys: the records back to LDS as the sorted runs they came as, and every record adds
       * up its lower bounds in all the runs */
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const u32 q = (u32) (wid * RPT + k) * WAVE + (u32) lane;
        if (ba[k] >> 31) {
          sh.raw[3 * q] = (u32) key[k];
          sh.raw[3 * q + 1] = (u32) (key[k] >> 32);
          sh.raw[3 * q + 2] = cnt[k];
        }
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
#pragma unroll
      for (int k = 0; k < RPT; k++) pos[k] = (ba[k] >> 31) ? pos[k] : 0u; /* (lanes without a record searched with whatever their registers held) */
      __syncthreads ();
      for (int i = 4 * tid; i < CAPS; i += 4 * NT) *reinterpret_cast<u32x4 *> (&sh.s.scnt[i]) = u32x4 { 0, 0, 0, 0 }; /* (the runs lay over the counts) */
      __syncthreads ();
    }
    if (GT4_NWAY_NOCONF) { /* (diagnostics: whatever the misdirected accesses made of the positions stays inside the tile) */
#pragma unroll
      for (int k = 0; k < RPT; k++) pos[k] = pos[k] < (u32) CAP ? pos[k] : (u32) CAP - 1u;
    }
    PHASE_STAMP (8);

    /* ---- the key once per position, the counts folded by LDS atomics */
    u32 lead_bits = 0;     /* LEAD: record k is the first of its key to arrive at its position (and, behind B5, is kept) */
    u32 lead_before[RPT];  /* ... its bitmap word as the record found it */
    if (LEAD && has_rec) {
      /* straight-line: the folds, then the claims (a lane without a record adds 0 to a word of its own and claims nothing) */
      if (p.rule == 1u) {
#pragma unroll
        for (int k = 0; k < RPT; k++) atomicAdd (&sh.s.scnt[(GT4_NWAY_NOCONF & 8) ? (u32) lane : nway_pick (nway_valid_mask (ba[k]), nway_skew (pos[k]), (u32) lane)], cnt[k] & nway_valid_mask (ba[k]));
      } else if (p.rule == 4u) {
#pragma unroll
        for (int k = 0; k < RPT; k++) atomicMax (&sh.s.scnt[nway_pick (nway_valid_mask (ba[k]), nway_skew (pos[k]), (u32) lane)], cnt[k] & nway_valid_mask (ba[k]));
      }
#pragma unroll
      for (int k = 0; k < RPT; k++)
        lead_before[k] = atomicOr (&sh.lead[it & 1][(GT4_NWAY_NOCONF & 16) ? (u32) lane : nway_pick (nway_valid_mask (ba[k]), pos[k] / (u32) Shared::LBP, (u32) lane)], (1u << (pos[k] % (u32) Shared::LBP)) & nway_valid_mask (ba[k]));
    }
    if (!LEAD && has_rec) {
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        if (!(ba[k] >> 31)) continue;
        u32 q = nway_skew (pos[k]);
        if (MODE == NWAY_DUPS) q = nway_skew (pos[k] + atomicAdd (&sh.s.scnt[q], 1u)); /* equal sample keys: one position each */
        else if (MODE == NWAY_TABLE) { /* (counts go to the table, below) */ }
        else if (MODE == NWAY_PROBE) {
          /* a record of list 0 (the first run: its slots are the tile's first) leaves its index + 1 */
          const u32 idx = (u32) (wid * RPT + k) * WAVE + (u32) lane;
#if GT4_KM > 8
          if (idx < sh.tab_len[tb][0]) {
#else
          if (uniform32 (sh.slot_run[tb][wid * RPT + k]) == 0u) {
#endif
            sh.s.scnt[q] = idx + 1u;
            p.table_keys[out_base + idx] = key[k];
          }
          continue;
        }
        else if (p.rule == 1u) atomicAdd (&sh.s.scnt[q], cnt[k]);
        else if (p.rule == 4u) atomicMax (&sh.s.scnt[q], cnt[k]);
        sh.s.skey[q] = key[k];
        reinterpret_cast<unsigned char *> (sh.live)[q] = MODE == NWAY_DUPS ? (unsigned char) (1u + list_of (k)) : (unsigned char) 1;
      }
    }
    PHASE_STAMP (9);
    /* ---- service window (the other wavefronts are ranking): the table of the tile two iterations
     * ahead from the entries asked for one iteration ago, the ticket drawn then, new requests -- and
     * the previous tile leaves its staging area as soon as the chain has its offset: only this
     * wavefront ever waits for the chain, and not before everybody else stands at B6 */
    bool wo_done = !(nway_staged (MODE) && pend);
    auto write_out = [&] (u64 excl_bytes) {
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc ((void *) (reinterpret_cast<char *> (out) + excl_bytes), 0, (int) (12 * pend_tot), 0x00020000);
      const u32 chunks = (3 * pend_tot + 3) >> 2;
      for (u32 c0 = 0; c0 < chunks; c0 += 4 * WAVE) {
        u32x4 w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const u32 c = c0 + (u32) u * WAVE + (u32) lane;
          w[u] = *reinterpret_cast<const u32x4 *> (sh.stage + 4 * (c < chunks ? c : chunks - 1u));
        }
#pragma unroll
        for (int u = 0; u < 4; u++) __builtin_amdgcn_raw_buffer_store_b128 (w[u], r, 16 * (c0 + (u32) u * WAVE + (u32) lane), 0, GT4_STORE_AUX);
      }
      wo_done = true;
    };
    if (service) {
      PHASE_STAMP (16);
      build_table (sv_row, sv_t2, tb2);
      PHASE_STAMP (17);
      const u32 t3 = uniform32 (sv_tk);
      sv_t2 = t3;
      sv_row = load_row (t3);
      if (lane == 0) sv_tk = deal (it + 4);
      PHASE_STAMP (18);
      if (!wo_done) {
        if (MODE == NWAY_UNION) {
          const bool mine = (u32) lane < pend_tile % WAVE;
          if (__all (!mine || (xagg & AGG_READY) != 0) && (xcarry & CARRY_READY)) {
            write_out (12 * ((xcarry & ~CARRY_READY) + dpp_wave_sum_u32 (mine ? (xagg & ~AGG_READY) : 0u)));
          } else {
            /* not yet: ask again, look again behind B5 */
            const u64 prow = pend_tile / WAVE;
            if (mine && !(xagg & AGG_READY)) xagg = peek_u32 (&agg[prow * WAVE + lane]);
            if (!(xcarry & CARRY_READY)) xcarry = peek_u64 (&carry[prow]);
          }
        } else {
          write_out (12 * pend_base);
        }
      }
      PHASE_STAMP (19);
    }
    PHASE_STAMP (10);
    __syncthreads (); /* B5: the tile in key order */
    PHASE_STAMP (11);
    if (service && !wo_done) write_out (12 * resolve_offset (agg, carry, pend_tile, lane, xagg, xcarry, ctl, spin_limit));
    if (GT4_NWAY_FILL) fill_g (); /* every walk of this tile is behind B5: the grouped keys of the next tile start from all-ones */
    PHASE_STAMP (12);

    u32 tile_total;
    if constexpr (LEAD) {
      /* ---- the leaders look at the folded counts: cutoff, sum of the kept counts; a leader that is not kept gives
       * its bit back */
      u32 lf[RPT];
#pragma unroll
      for (int k = 0; k < RPT; k++) lf[k] = 0;
      if (has_rec) {
        /* (every lane reads: position 0 where no record is) */
        if (p.rule == 7u) {
#pragma unroll
          for (int k = 0; k < RPT; k++) lf[k] = p.count_override;
        } else {
#pragma unroll
          for (int k = 0; k < RPT; k++) lf[k] = lds_load<u32> (lds_offset (&sh.s.scnt[0]) + 4u * ((GT4_NWAY_NOCONF & 64) ? (u32) lane : nway_skew (pos[k])));
        }
        const u32 least = p.filter == FILTER_RAW ? 0u : p.cutoff; /* kept iff the folded count reaches it */
        u32 drop = 0;
#pragma unroll
        for (int k = 0; k < RPT; k++) { /* (masks, not conditions: see nway_pick) */
          const u32 leads = nway_valid_mask (ba[k]) & (((lead_before[k] >> (pos[k] % (u32) Shared::LBP)) & 1u) - 1u);
          const u32 enough = lf[k] >= least ? ~0u : 0u;
          acc_sum += lf[k] & leads & enough;
          lead_bits |= leads & enough & (1u << k);
          drop |= leads & ~enough & (1u << k);
        }
        if (drop) { /* (rare: a cutoff above the counts) */
#pragma unroll
          for (int k = 0; k < RPT; k++)
            if ((drop >> k) & 1u) atomicAnd (&sh.lead[it & 1][pos[k] / (u32) Shared::LBP], ~(1u << (pos[k] % (u32) Shared::LBP)));
        }
      }
      PHASE_STAMP (13);
      __syncthreads (); /* B6: the bitmap holds the kept leaders; the staging area is free */
      PHASE_STAMP (14);
      nway_inject<2> (lds_offset (&sh.hdr[0][0]) + 4u * (u32) lane);
      /* kept leaders in front of every bitmap word: every wavefront scans the bitmap itself (LWL words per lane) */
      constexpr int LWL = Shared::LWL;
      u32 w[LWL], c = 0;
#pragma unroll
      for (int j = 0; j < LWL; j++) w[j] = sh.lead[it & 1][lane * LWL + j]; /* (consecutive: 16-byte reads) */
#pragma unroll
      for (int j = 0; j < LWL; j++) c += (u32) __popc (w[j]);
      const u32 incl = dpp_inclusive_scan_u32 (c);
      tile_total = (u32) __builtin_amdgcn_readlane ((int) incl, WAVE - 1);
      blk_cnt += tile_total;
      if (MODE == NWAY_UNION && service) {
        if (lane == 0) publish_u32 (&agg[cur], AGG_READY | tile_total);
      }
      if (MODE == NWAY_COUNT && tid == 0 && p.tile_totals) p.tile_totals[cur] = tile_total;
      PHASE_STAMP (16); /* (diagnostics, wavefronts other than the service one: the bitmap scan) */
      if (nway_staged (MODE) && has_rec) {
        u32 before = incl - c;
        u32 pre16[LWL]; /* (every lane its words, whether it holds a leader or not) */
#pragma unroll
        for (int j = 0; j < LWL; j++) {
          pre16[j] = before;
          before += (u32) __popc (w[j]);
        }
        if constexpr (LWL == 8) { /* eight 16-bit prefixes: one 16-byte store */
          *reinterpret_cast<u32x4 *> (&sh.wpre[wid][lane * LWL]) = u32x4 { pre16[0] | (pre16[1] << 16), pre16[2] | (pre16[3] << 16), pre16[4] | (pre16[5] << 16), pre16[6] | (pre16[7] << 16) };
        } else {
#pragma unroll
          for (int j = 0; j < LWL; j++) sh.wpre[wid][lane * LWL + j] = (unsigned short) pre16[j];
        }
        /* (the table is this wavefront's own: LDS operations of one wavefront complete in order) */
        u32 pw[RPT], lw[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          pw[k] = sh.wpre[wid][pos[k] / (u32) Shared::LBP];
          lw[k] = sh.lead[it & 1][pos[k] / (u32) Shared::LBP];
        }
        PHASE_STAMP (17); /* (diagnostics: prefix table written, words asked for) */
#pragma unroll
        for (int k = 0; k < RPT; k++) { /* (what is not kept goes to the lane's trash record) */
          const u32 slot = (GT4_NWAY_NOCONF & 32) ? (u32) CAP + 2u + (u32) lane : nway_pick (0u - ((lead_bits >> k) & 1u), pw[k] + (u32) __popc (lw[k] & ((1u << (pos[k] % (u32) Shared::LBP)) - 1u)), (u32) CAP + 2u + (u32) lane);
          sh.stage[3 * slot] = (u32) key[k];
          sh.stage[3 * slot + 1] = (u32) (key[k] >> 32);
          sh.stage[3 * slot + 2] = lf[k];
        }
      }
    } else {
    /* ---- positions in order, one per lane (a wavefront walks its RPT chunks of 64): keep test, ballots */
    u64 okey[RPT];
    u32 ocnt[RPT];
    u32 keep_bits = 0, wave_kept = 0;
    u32 kpre[RPT]; /* kept in the wavefront's earlier chunks (uniform) */
#pragma unroll
    for (int i = 0; i < RPT; i++) {
      okey[i] = 0;
      ocnt[i] = 0;
      kpre[i] = 0;
    }
    if (MODE == NWAY_PROBE && p.table_cols <= (u32) Shared::ROW_COLS_MAX) {
      /* (behind B5: every record of list 0 has left its index + 1 where its key's records look) */
      u32 row[RPT], val[RPT];
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        row[k] = has_rec && (ba[k] >> 31) ? sh.s.scnt[nway_skew (pos[k])] - 1u : 0xffffffffu; /* (0: list 0 does not hold the key) */
        val[k] = p.rule == 7u ? p.count_override : cnt[k];
      }
      table_rows (row, val, uniform32 (sh.tab_len[tb][0]), out_base);
    } else if (MODE == NWAY_PROBE && has_rec) {
      /* (wide tables: the host has zeroed the table) */
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        if (!(ba[k] >> 31)) continue;
        const u32 r = sh.s.scnt[nway_skew (pos[k])];
        if (r) p.table_counts[(out_base + r - 1u) * p.table_cols + p.table_col[list_of (k)]] = p.rule == 7u ? p.count_override : cnt[k];
      }
    }
    if (MODE != NWAY_PROBE && has_pos) {
#pragma unroll
      for (int i = 0; i < RPT; i++) {
        const u32 q = nway_skew ((u32) (wid * RPT + i) * WAVE + (u32) lane);
        const u32 lv = lds_load<unsigned char> (lds_offset (&sh.live[0]) + q);
        const bool on = lv != 0;
        okey[i] = lds_load<u64> (lds_offset (&sh.s.skey[0]) + 8u * q);
        u32 f = lds_load<u32> (lds_offset (&sh.s.scnt[0]) + 4u * q);
        if (MODE == NWAY_DUPS) f = lv - 1u; /* a merged sample keeps the list it came from: the partition counts them */
        else if (MODE == NWAY_TABLE) f = 0;
        else if (p.rule == 7u) f = p.count_override;
        ocnt[i] = f;
        const bool keep = on & (MODE == NWAY_DUPS || MODE == NWAY_TABLE || p.filter == FILTER_RAW || f >= p.cutoff);
        keep_bits |= keep ? 1u << i : 0u;
        acc_sum += keep ? f : 0u;
        kpre[i] = wave_kept;
        wave_kept += (u32) __popcll (__builtin_amdgcn_ballot_w64 (keep));
      }
    }
    if (lane == 0) sh.wkept[wid] = wave_kept;
    PHASE_STAMP (13);
    __syncthreads (); /* B6: kept per wavefront; the staging area is free */
    PHASE_STAMP (14);
    {
      const u32 x = lane < NW ? sh.wkept[lane] : 0u;
      const u32 incl2 = dpp_inclusive_scan_u32 (x);
      tile_total = (u32) __builtin_amdgcn_readlane ((int) incl2, WAVE - 1);
      const u32 wbase = GT4_NWAY_LEAN ? (wid ? (u32) __builtin_amdgcn_readlane ((int) incl2, wid - 1) : 0u) : dpp_wave_sum_u32 (lane < wid ? x : 0u);
      blk_cnt += tile_total;
      if (MODE == NWAY_UNION && service) {
        if (lane == 0) publish_u32 (&agg[cur], AGG_READY | tile_total);
      }
      if ((MODE == NWAY_COUNT || MODE == NWAY_TABLE) && tid == 0 && p.tile_totals) p.tile_totals[cur] = tile_total;
      if (nway_staged (MODE) && wave_kept) {
#pragma unroll
        for (int i = 0; i < RPT; i++) {
          const bool keep = (keep_bits >> i) & 1u;
          const u64 m = __builtin_amdgcn_ballot_w64 (keep);
          const u32 slot = wbase + kpre[i] + __builtin_amdgcn_mbcnt_hi ((u32) (m >> 32), __builtin_amdgcn_mbcnt_lo ((u32) m, 0u));
          if (keep) {
            sh.stage[3 * slot] = (u32) okey[i];
            sh.stage[3 * slot + 1] = (u32) (okey[i] >> 32);
            sh.stage[3 * slot + 2] = ocnt[i];
          }
        }
      }
      if (MODE == NWAY_TABLE) {
        const bool via_lds = p.table_cols <= (u32) Shared::ROW_COLS_MAX; /* uniform */
        /* Wide tables only: the tile's rows of the count matrix start as zeros, written here, 16 bytes per lane (round 4:
         * instead of a memset of the whole matrix in front of the launch); the records' own stores follow behind a wait
         * for these and the barrier below, so they land on the zeros. */
        if (!via_lds) {
          const u64 words = (u64) tile_total * p.table_cols;
          const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc ((void *) (p.table_counts + out_base * p.table_cols), 0, (int) (4 * words), 0x00020000);
          for (u32 c = (u32) tid; 4ull * c < words; c += NT) __builtin_amdgcn_raw_buffer_store_b128 (u32x4 { 0, 0, 0, 0 }, zr, 16 * c, 0, 0);
          asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
        }
        /* the key column, and every position's row (over the ordered tile's counts, which this mode
         * does not fold) for the records to find */
        if (wave_kept) {
#pragma unroll
          for (int i = 0; i < RPT; i++) {
            const bool keep = (keep_bits >> i) & 1u;
            const u64 m = __builtin_amdgcn_ballot_w64 (keep);
            const u32 slot = wbase + kpre[i] + __builtin_amdgcn_mbcnt_hi ((u32) (m >> 32), __builtin_amdgcn_mbcnt_lo ((u32) m, 0u));
            if (keep) {
              p.table_keys[out_base + slot] = okey[i];
              sh.s.scnt[nway_skew ((u32) (wid * RPT + i) * WAVE + (u32) lane)] = slot;
            }
          }
        }
        __syncthreads ();
        if (via_lds) {
          u32 row[RPT];
#pragma unroll
          for (int k = 0; k < RPT; k++) row[k] = has_rec && (ba[k] >> 31) ? sh.s.scnt[nway_skew (pos[k])] : 0xffffffffu;
          table_rows (row, cnt, tile_total, out_base); /* (its barrier stands behind every thread's look at its rows) */
        } else {
          if (has_rec) {
#pragma unroll
            for (int k = 0; k < RPT; k++) {
              if (!(ba[k] >> 31)) continue;
              const u32 col = p.table_col[list_of (k)];
              const u64 row = out_base + sh.s.scnt[nway_skew (pos[k])];
              p.table_counts[row * p.table_cols + col] = cnt[k];
            }
          }
          __syncthreads (); /* (the rows lie where the next tile's counts are zeroed) */
        }
      }
    }
    } /* (!LEAD) */
    pend = nway_staged (MODE);
    pend_tot = tile_total;
    pend_tile = cur;
    pend_base = out_base;
    PHASE_STAMP (15);
    it++;
    {
      const int t0 = tb;
      tb = tb1;
      tb1 = tb2;
      tb2 = t0;
    }
  }
#ifdef GT4_PROFILE_PHASES
  if (tid == GT4_STAMP_TID)
    for (int i = 0; i < 24; i++) atomicAdd (&ctl->phase_cycles[i], ph[i]);
#endif
  /* drain: the last tile is still staged */
  if (nway_staged (MODE) && pend) {
    __syncthreads ();
    if (MODE == NWAY_UNION && wid == 0) {
      const u64 x = resolve_offset (agg, carry, pend_tile, lane, 0, 0, ctl, spin_limit);
      if (lane == 0) sh.excl = x;
    }
    __syncthreads ();
    write_out_tile<NT> (out, MODE == NWAY_UNION ? uniform64 (sh.excl) : pend_base, pend_tot, sh.stage, tid);
  }
  if (MODE != NWAY_DUPS && MODE != NWAY_TABLE && MODE != NWAY_PROBE) {
    const u64 v = wave_sum (acc_sum);
    if (lane == 0 && v) atomicAdd (&ctl->total_count[0], v);
    if (tid == 0 && blk_cnt) atomicAdd (&ctl->n_words[0], blk_cnt);
  }
  if (MODE == NWAY_TABLE && tid == 0 && blk_cnt) atomicAdd (&ctl->n_words[0], blk_cnt); /* the table's rows */
}

#if GT4_KM == 8
#include "gt4hip_nsub.h"
#endif

constexpr int NWAY_NT = GT4_NWAY_NT;
constexpr int NWAY_NBF = GT4_NWAY_NBF;
/* positions per thread: the modes that keep no ordered copy of the tile (GT4_NWAY_LEAD) have LDS for one more */
#ifndef GT4_NWAY_RPT_LEAD
#define GT4_NWAY_RPT_LEAD GT4_NWAY_RPT
#endif
constexpr int nway_rpt (int mode) { return nway_lead (mode) ? GT4_NWAY_RPT_LEAD : GT4_NWAY_RPT; }
constexpr int nway_cap (int mode) { return NWAY_NT * nway_rpt (mode); }
constexpr int NWAY_CAP_MIN = NWAY_NT * (GT4_NWAY_RPT_LEAD < GT4_NWAY_RPT ? GT4_NWAY_RPT_LEAD : GT4_NWAY_RPT);

template <int MODE>
hipError_t launch_nway (hipStream_t s, int grid, const NwayParams &p, const u64 *part, u32 *out, u64 *desc, PairControl *ctl)
{
  hipLaunchKernelGGL ((k_nway_merge<NWAY_NT, nway_rpt (MODE), NWAY_NBF, MODE>), dim3 (grid), dim3 (NWAY_NT), 0, s, p, part, out, desc, ctl);
  return hipGetLastError ();
}

hipError_t launch_nway_mode (hipStream_t s, int mode, int grid, const NwayParams &p, const u64 *part, u32 *out, u64 *desc, PairControl *ctl, bool sub)
{
#if GT4_KM == 8
  if (sub && mode == NWAY_UNION) {
    hipLaunchKernelGGL ((k_nway_sub<NWAY_UNION>), dim3 (grid), dim3 (SUB_NT), 0, s, p, part, out, desc, ctl);
    return hipGetLastError ();
  }
  if (sub && mode == NWAY_COUNT) {
    hipLaunchKernelGGL ((k_nway_sub<NWAY_COUNT>), dim3 (grid), dim3 (SUB_NT), 0, s, p, part, out, desc, ctl);
    return hipGetLastError ();
  }
#endif
  if (mode == NWAY_DUPS) return launch_nway<NWAY_DUPS> (s, grid, p, part, out, desc, ctl);
  if (mode == NWAY_COUNT) return launch_nway<NWAY_COUNT> (s, grid, p, part, out, desc, ctl);
  if (mode == NWAY_TABLE) return launch_nway<NWAY_TABLE> (s, grid, p, part, out, desc, ctl);
  if (mode == NWAY_PROBE) return launch_nway<NWAY_PROBE> (s, grid, p, part, out, desc, ctl);
  return launch_nway<NWAY_UNION> (s, grid, p, part, out, desc, ctl);
}

int nway_blocks_per_cu (int mode)
{
  static int cache[5] = { 0, 0, 0, 0, 0 };
  if (!cache[mode]) {
    int n = 0;
    hipError_t e;
    if (mode == NWAY_DUPS) e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_nway_merge<NWAY_NT, nway_rpt (NWAY_DUPS), NWAY_NBF, NWAY_DUPS>, NWAY_NT, 0);
    else if (mode == NWAY_COUNT) e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_nway_merge<NWAY_NT, nway_rpt (NWAY_COUNT), NWAY_NBF, NWAY_COUNT>, NWAY_NT, 0);
    else if (mode == NWAY_TABLE) e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_nway_merge<NWAY_NT, nway_rpt (NWAY_TABLE), NWAY_NBF, NWAY_TABLE>, NWAY_NT, 0);
    else if (mode == NWAY_PROBE) e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_nway_merge<NWAY_NT, nway_rpt (NWAY_PROBE), NWAY_NBF, NWAY_PROBE>, NWAY_NT, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_nway_merge<NWAY_NT, nway_rpt (NWAY_UNION), NWAY_NBF, NWAY_UNION>, NWAY_NT, 0);
    if (e != hipSuccess || n < 1) n = 1;
    const int by_regs = nway_waves_per_simd (NWAY_NT) * 4 / (NWAY_NT / 64);
    if (by_regs >= 1 && n > by_regs) n = by_regs;
    cache[mode] = n;
  }
  return cache[mode];
}


/* ------------------------------------------------------------------ host orchestration */


/* ------------------------------------------------------------------ host orchestration */

namespace host_part {

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
 * tile (G * S records) plus GT4_NWAY_MARGIN (five) standard deviations of the lists' offsets against their sample grids; tiles beyond the
 * capacity are cut in two (k_nway_emit). */
void nway_samples_per_tile (u32 k, int positions, long max_records, u32 *first_try, u32 *sure)
{
  const double cap = (double) positions - 0.5 * NWAY_HS * k; /* half a slot of padding per run, on average */
  const double margin = GT4_NWAY_MARGIN * NWAY_SAMPLE * sqrt ((double) k / 6.0);
  long g1 = (long) ((cap - margin) / NWAY_SAMPLE);
  long g0 = ((long) positions - (long) NWAY_HS * k) / NWAY_SAMPLE - (2L * k - 1);
  if (max_records > 0) { /* k_nway_sub: the records its workers take between them */
    const long r1 = (long) (((double) max_records - margin) / NWAY_SAMPLE), r0 = max_records / NWAY_SAMPLE - (2L * k - 1);
    if (r1 < g1) g1 = r1;
    if (r0 < g0) g0 = r0;
  }
  if (g0 < 1) g0 = 1;
  if (g1 < g0) g1 = g0;
  *first_try = (u32) g1;
  *sure = (u32) g0;
}

}  // namespace host_part
using namespace host_part;

int nway_run (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t k, uint32_t rule, uint32_t cutoff, uint32_t ovr,
                     uint32_t filter, bool count_only, gt4hip_list *out, uint64_t *n_words, uint64_t *total_count, double *device_ms,
                     int *used, gt4hip_count_table *table, const uint32_t *cols, bool probe)
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
  {
    const hipError_t e0 = hipEventRecord (ctx->ev[0], st);
    if (e0 != hipSuccess) return gt4hip_fail (ctx, GT4HIP_EHIP, "hipEventRecord failed: %s", hipGetErrorString (e0)); /* (nothing is owned yet) */
  }
  /* option "kway" = 1 (the default) lets the call decline clustered keys: *used = 0, the caller takes the tree */
  /* Host read-backs (round 5: nine per call of three levels -> four; each is a drained stream plus 20 - 30 us, 0.3 ms of
   * a 4.7 ms call on an eighth of the bench's lists, i.e. of one GPU's shard at 8 GPUs).  A top level of one tile reads
   * nothing back; the sample levels' control blocks are not read back (their error word stays set through the later
   * launches and is seen with the last one). */
  const bool may_decline = ctx->kway_enabled == 1 && !table && ctx->kway_vt == 0;
  u32 probe_windows = 0;
  bool probe_pending = false;
  hipMemsetAsync (ctx->ctl, 0, sizeof (PairControl), st);
  if (may_decline) {
    uint32_t longest = 0;
    for (uint32_t i = 1; i < k; i++)
      if (lists[i]->n_words > lists[longest]->n_words) longest = i;
    const u64 nl = lists[longest]->n_words;
    if (nl >= 16ull * NWAY_PROBE_KEYS) {
      const u32 windows = (u32) (nl / (4 * NWAY_PROBE_KEYS) < NWAY_PROBE_WINDOWS ? nl / (4 * NWAY_PROBE_KEYS) : NWAY_PROBE_WINDOWS);
      const hipError_t e = hipMemsetAsync ((char *) ctx->scratch + 32, 0, 8, st);
      if (e != hipSuccess) return gt4hip_fail (ctx, GT4HIP_EHIP, "N-way key probe failed: %s", hipGetErrorString (e));
      hipLaunchKernelGGL (k_nway_probe, dim3 (windows), dim3 (256), 0, st, (const u32 *) lists[longest]->dev, nl, windows, (u32) nway_buckets (NWAY_NBF * nway_cap (NWAY_UNION)), (u32 *) ctx->scratch + 8);
      probe_windows = windows;
      probe_pending = true;
      /* (read at once after all: read with the first partition read-back, a call that declines had sampled, merged
       * samples and partitioned for nothing -- 1.3 ms of a 53 ms tree on the clustered bench lists) */
      hipError_t e2 = hipMemcpyAsync (ctx->scratch_host + 4, (char *) ctx->scratch + 32, 8, hipMemcpyDeviceToHost, st);
      if (e2 == hipSuccess) e2 = hipStreamSynchronize (st);
      if (e2 != hipSuccess) return gt4hip_fail (ctx, GT4HIP_EHIP, "N-way key probe failed: %s", hipGetErrorString (e2));
      probe_pending = false;
      if (5ull * (u32) ctx->scratch_host[4] > probe_windows) {
        ctx->kway_declined++;
        return GT4HIP_OK; /* *used = 0 */
      }
    }
  }
  /* sample levels until one fits a single tile */
  const u64 one_tile = (u64) NWAY_CAP_MIN - (u64) NWAY_HS * k;
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
    }
    if (!rc && up.total) {
      u64 g = (up.total + 255) / 256;
      if (g > 16384) g = 16384;
      hipLaunchKernelGGL (k_nway_sample, dim3 ((unsigned) g), dim3 (256), 0, st, lo.p, up.p);
    }
    levels.push_back (up);
    if (rc) {
      cleanup ();
      return rc;
    }
  }
  /* top-down: the merged samples of level l+1 cut level l into tiles */
  gt4hip_list *merged = NULL; /* merged sample records of the level above */
  for (int l = (int) levels.size () - 1; l >= 0 && !rc; l--) {
    Level &lv = levels[l];
    const u64 m_total = merged ? merged->n_words : 0;
    const int mode = l > 0 ? NWAY_DUPS : (table ? (probe ? NWAY_PROBE : NWAY_TABLE) : (count_only ? NWAY_COUNT : NWAY_UNION));
    const int cap = nway_cap (mode); /* positions of a tile */
    const u32 n_buckets = (u32) nway_buckets (NWAY_NBF * cap);
    u32 g_try, g_sure;
#if GT4_KM == 8
    const bool sub = ctx->kway_sub != 0 && (mode == NWAY_UNION || mode == NWAY_COUNT); /* option "kway_sub" = 0: k_nway_merge for every mode */
    const u32 max_rec = sub ? (u32) SUB_MAXREC : 0xffffffffu;
    nway_samples_per_tile (k, cap, sub ? (long) SUB_MAXREC : 0L, &g_try, &g_sure);
#else
    const bool sub = false;
    const u32 max_rec = 0xffffffffu;
    nway_samples_per_tile (k, cap, 0L, &g_try, &g_sure);
#endif
    if (ctx->kway_g > 0) g_try = (u32) ctx->kway_g;
    u32 G = g_try;
    u64 tiles = 1;
    const u64 *part_final = NULL; /* the table the tile kernel reads: the partition's, or the one with the split tiles */
    for (;;) {
      tiles = m_total ? m_total / G + 2 : 1;
      if (tiles >= 0xfffffff0ull) {
        rc = gt4hip_fail (ctx, GT4HIP_EINVAL, "lists too long: %llu tiles", (unsigned long long) tiles);
        break;
      }
      lv.p.num_tiles = (u32) tiles;
      if ((rc = nway_grow (ctx, (void **) &ctx->kway_part, &ctx->kway_part_bytes, (size_t) (tiles + 1) * NWAY_PSTRIDE * 8))) break;
      const u64 threads = (tiles + 1) * NWAY_PSTRIDE;
      if (merged && ctx->kway_vt != 97 && G <= NWAY_G_MAX) {
        /* from the merged samples' list numbers (option "kway_vt" = 97 keeps the searches over whole brackets: tests) */
        const u64 n_br = (tiles + 1 + NWAY_BRACKET - 1) / NWAY_BRACKET;
        if ((rc = nway_grow (ctx, (void **) &ctx->kway_cnt, &ctx->kway_cnt_bytes, (size_t) n_br * NWAY_MAX * 4))) break;
        hipLaunchKernelGGL (k_nway_sample_counts, dim3 ((unsigned) ((n_br + 3) / 4)), dim3 (256), 0, st, (const u32 *) merged->dev, m_total, G, n_br, (u32 *) ctx->kway_cnt);
        hipLaunchKernelGGL (k_nway_bracket_bases, dim3 (1), dim3 (NWAY_MAX > 8 ? 1024 : 64 * NWAY_MAX), 0, st, (u32 *) ctx->kway_cnt, n_br);
        hipLaunchKernelGGL (k_nway_partition_rows, dim3 ((unsigned) n_br), dim3 (64), 0, st, lv.p, (const u32 *) merged->dev, m_total, G, n_buckets,
                            (const u32 *) ctx->kway_cnt, (u64 *) ctx->kway_part);
      } else {
        for (int pass = 0; pass < 2; pass++)
          hipLaunchKernelGGL (k_nway_partition, dim3 ((unsigned) ((threads + 255) / 256)), dim3 (256), 0, st, lv.p, merged ? (const u32 *) merged->dev : NULL,
                              m_total, G, n_buckets, (u64 *) ctx->kway_part, pass);
      }
      /* tiles that do not fit are cut in two (k_nway_need / _scan / _emit): flags = { more than two pieces,
       * clustered tiles, tiles cut, tiles of the final table } */
      const u64 n_blocks = (tiles + 1 + NWAY_SPLIT_BLOCK - 1) / NWAY_SPLIT_BLOCK;
      if ((rc = nway_grow (ctx, (void **) &ctx->kway_need, &ctx->kway_need_bytes, (size_t) (tiles + 1 + n_blocks + 4) * 4))) break;
      u32 *const need = (u32 *) ctx->kway_need, *const block_sums = need + tiles + 1;
      if (tiles == 1 && !merged && levels[l].total <= one_tile) {
        /* the top level: one tile that fits by construction -- nothing to cut, nothing to read back */
        part_final = (const u64 *) ctx->kway_part;
        if (l == 0) ctx->kway_splits = 0;
        break;
      }
      hipMemsetAsync (ctx->scratch, 0, 32, st);
      hipLaunchKernelGGL (k_nway_need, dim3 ((unsigned) n_blocks), dim3 (NWAY_SPLIT_BLOCK), 0, st, (const u64 *) ctx->kway_part, (u32) tiles, (u32) (cap / NWAY_HS), max_rec, need, block_sums,
                          (u32 *) ctx->scratch);
      hipLaunchKernelGGL (k_nway_need_scan, dim3 (1), dim3 (1024), 0, st, block_sums, (u32) n_blocks, (u32 *) ctx->scratch + 3);
      hipError_t e = hipMemcpyAsync (ctx->scratch_host, ctx->scratch, 40, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize (st);
      if (e != hipSuccess) {
        rc = gt4hip_fail (ctx, GT4HIP_EHIP, "N-way partition failed: %s", hipGetErrorString (e));
        break;
      }
      const u32 *const fl = (const u32 *) ctx->scratch_host;
      if (probe_pending) { /* (the probe ran in front of everything else on this stream) */
        probe_pending = false;
        if (5ull * fl[8] > probe_windows) {
          ctx->kway_declined++;
          if (merged) gt4hip_list_free (merged);
          cleanup ();
          return GT4HIP_OK; /* *used = 0 */
        }
      }
      bool overflow = fl[0] != 0;
      part_final = (const u64 *) ctx->kway_part;
      if (!overflow && fl[2]) {
        const u64 tiles2 = fl[3];
        if ((rc = nway_grow (ctx, (void **) &ctx->kway_part2, &ctx->kway_part2_bytes, (size_t) (tiles2 + 1) * NWAY_PSTRIDE * 8))) break;
        hipLaunchKernelGGL (k_nway_emit, dim3 ((unsigned) n_blocks), dim3 (NWAY_SPLIT_BLOCK), 0, st, lv.p, (const u64 *) ctx->kway_part, (u32) tiles, need, block_sums,
                            n_buckets, (u32) (cap / NWAY_HS), max_rec, (u64 *) ctx->kway_part2, (u32 *) ctx->scratch);
        e = hipMemcpyAsync (ctx->scratch_host + 4, ctx->scratch, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize (st);
        if (e != hipSuccess) {
          rc = gt4hip_fail (ctx, GT4HIP_EHIP, "N-way partition failed: %s", hipGetErrorString (e));
          break;
        }
        overflow = (u32) ctx->scratch_host[4] != 0;
        if (!overflow) {
          if (l == 0) ctx->kway_splits = fl[2];
          part_final = (const u64 *) ctx->kway_part2;
          tiles = tiles2;
          lv.p.num_tiles = (u32) tiles;
        }
      } else if (!overflow && l == 0) {
        ctx->kway_splits = 0;
      }
      if (!overflow) {
        if (l == 0 && may_decline && tiles >= 64 && 5ull * fl[1] > tiles) {
          /* the probe of the longest list did not see it, the tiles' own samples do: clustered keys */
          ctx->kway_declined++;
          if (merged) gt4hip_list_free (merged);
          cleanup ();
          return GT4HIP_OK; /* *used = 0 */
        }
        break;
      }
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
    lv.p.force_fallback = ctx->kway_vt == 99 ? 1u : (ctx->kway_vt == 98 ? 2u : 0u); /* option "kway_vt" = 99 / 98: every tile takes the search path / the pivot-run buckets (tests) */
    lv.p.scan_group = ctx->scan_group > 0 ? 1u : (ctx->scan_group < 0 ? 0u : (tiles > (48000ull << 6) ? 1u : 0u));
    lv.p.dynamic = ctx->dynamic > 0 ? 1u : (ctx->dynamic < 0 ? 0u : (mode == NWAY_UNION ? 1u : 0u));
    u32 *dst = NULL;
    if (l > 0) {
      if ((rc = gt4hip_list_new (ctx, lv.total ? lv.total : 1, lists[0]->word_length, &merged))) break;
      merged->n_words = lv.total;
      dst = (u32 *) merged->dev;
    } else if (!count_only) {
      dst = (u32 *) out->dev;
    }
    int grid = ctx->n_cus * (sub ? 1 : nway_blocks_per_cu (mode));
    if (ctx->grid_override > 0) grid = (int) ctx->grid_override;
    if (mode == NWAY_UNION) {
      if ((rc = nway_grow (ctx, (void **) &ctx->desc, &ctx->desc_bytes, nway_desc_bytes (tiles)))) break;
      hipMemsetAsync (ctx->desc, 0, nway_desc_bytes (tiles), st);
      if ((u64) grid > tiles + 1) grid = (int) tiles + 1;
    } else if ((u64) grid > tiles) {
      grid = (int) tiles;
    }
    if (l == 0 && table && probe) { /* rows = the records of list 0: the table exists before the launch */
      if ((rc = gt4hip_table_alloc (ctx, table, lists[0]->n_words, table->n_lists))) break;
      table->n_keys = lists[0]->n_words;
      /* (up to ROW_COLS_MAX columns every row leaves the kernel whole, zeros included: no memset -- round 5) */
      if (table->n_lists > (uint32_t) NwayShared<NWAY_NT, nway_rpt (NWAY_PROBE), NWAY_NBF, NWAY_PROBE>::ROW_COLS_MAX)
        hipMemsetAsync (table->device_counts, 0, (size_t) table->n_keys * table->n_lists * 4, st);
      lv.p.table_keys = (u64 *) table->device_keys;
      lv.p.table_counts = (u32 *) table->device_counts;
      lv.p.table_cols = table->n_lists;
      for (uint32_t i = 0; i < k; i++) lv.p.table_col[i] = cols[i];
    } else if (l == 0 && table) {
      /* ONE launch (round 4): every tile writes its rows where its records start -- a tile has at most as many distinct
       * keys as records, so the table is allocated for the records and stays RAGGED (unused rows behind every tile's;
       * gt4hip_table_download and gt4hip_table_compact know, see gt4hip_count_table).  Round 3 counted every tile's
       * distinct keys in a launch of their own first: the records were read twice. */
      if ((rc = nway_grow (ctx, (void **) &ctx->desc, &ctx->desc_bytes, (size_t) tiles * 4 + 32 + (size_t) ((tiles + 1 + 1023) / 1024) * 8))) break; /* the tiles' totals, then their sums per block of 1024 */
      if ((rc = gt4hip_table_alloc (ctx, table, lv.total, table->n_lists))) break;
      lv.p.tile_totals = (u32 *) ctx->desc;
      lv.p.table_keys = (u64 *) table->device_keys;
      lv.p.table_counts = (u32 *) table->device_counts;
      lv.p.table_cols = table->n_lists;
      for (uint32_t i = 0; i < k; i++) lv.p.table_col[i] = cols[i];
    }
    hipMemsetAsync (ctx->ctl, 0, offsetof (PairControl, error), st); /* (totals, ticket; the error word stays) */
    hipMemsetAsync (&ctx->ctl->role, 0, sizeof (PairControl) - offsetof (PairControl, role), st);
    if (l == 0) hipEventRecord (ctx->ev[1], st);
    hipError_t e = launch_nway_mode (st, mode, grid, lv.p, part_final, dst, (u64 *) ctx->desc, ctx->ctl, sub);
    if (e != hipSuccess) {
      rc = gt4hip_fail (ctx, GT4HIP_EHIP, "N-way merge launch failed: %s", hipGetErrorString (e));
      break;
    }
    if (l == 0) hipEventRecord (ctx->ev[2], st);
    /* every level reads its control block back: a refused tile or a wait that gave up must not go unseen */
    e = l == 0 ? hipEventRecord (ctx->ev[3], st) : hipSuccess;
    if (e != hipSuccess) {
      rc = gt4hip_fail (ctx, GT4HIP_EHIP, "hipEventRecord failed: %s", hipGetErrorString (e));
      break;
    }
    if (l > 0) continue; /* (a sample level: its error word, if any, is still there behind the last launch) */
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
        static const char *sub_names[24] = { "w:wait cuts", "w:gather", "w:buckets+count", "w:scan", "w:group+walk", "w:wait gathered", "w:raw store", "w:wait table+fetch", "w:zero+fold", "w:wait offset", "w:stores", "w:ordered", "w:fill+back edge", "s:table+ticket", "s:wait end", "s:totals", "s:wait gathered", "s:raw store+fetch", "s:offset (ready)", "s:wait raw", "s:cuts", "s:offset (waited)", "-", "s:back edge" };
        static const char *old_names[24] = { "p0: zeroing", "B1", "scan1", "B2", "scan2", "B3", "group", "B4", "rank", "fold", "service", "B5", "writeout+fill", "order", "B6", "stage+publish", "sv:-", "sv:table", "sv:ticket+row", "sv:try writeout | header", "p0: wait for records", "p0: buckets+atomics", "p0: fetch issue", "back edge" };
        unsigned long long tot = 0;
        for (int i = 0; i < 24; i++) tot += ctx->ctl_host->phase_cycles[i];
        fprintf (stderr, "[nway phases] tiles %llu:", (unsigned long long) tiles);
        const char **names = sub ? sub_names : old_names;
        for (int i = 0; i < 24; i++) fprintf (stderr, " %s %.1f%%", names[i], tot ? 100.0 * ctx->ctl_host->phase_cycles[i] / tot : 0.0);
        fprintf (stderr, " | avg cycles/tile %.0f\n", tiles ? (double) tot / tiles : 0.0);
      }
#endif
      *n_words = ctx->ctl_host->n_words[0];
      *total_count = ctx->ctl_host->total_count[0];
      if (table && !probe) {
        /* the ragged table's index: rows before every tile (compact) and where the tile's rows lie (padded) */
        table->n_keys = *n_words;
        if ((rc = gt4hip_table_set_ragged (ctx, table, tiles))) break;
        {
          const u64 nb = (tiles + 1 + 1023) / 1024;
          u64 *const bsum = (u64 *) ((char *) ctx->desc + (((size_t) tiles * 4 + 15) & ~(size_t) 15)); /* (behind the tiles' totals) */
          hipLaunchKernelGGL (k_nway_base_sums, dim3 ((unsigned) nb), dim3 (1024), 0, st, (const u32 *) ctx->desc, tiles, bsum);
          hipLaunchKernelGGL (k_nway_base_scan, dim3 (1), dim3 (1024), 0, st, bsum, nb);
          hipLaunchKernelGGL (k_nway_tile_bases, dim3 ((unsigned) nb), dim3 (1024), 0, st, (const u32 *) ctx->desc, tiles, (const u64 *) bsum, (u64 *) gt4hip_table_compact_bases (table));
        }
        hipLaunchKernelGGL (k_nway_padded_bases, dim3 ((unsigned) ((tiles + 256) / 256)), dim3 (256), 0, st, part_final, tiles, k, (u64 *) gt4hip_table_padded_bases (table));
        e = hipStreamSynchronize (st);
        if (e != hipSuccess) {
          rc = gt4hip_fail (ctx, GT4HIP_EHIP, "count table index failed: %s", hipGetErrorString (e));
          break;
        }
      }
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


}  // namespace GT4_KM_NS

}  // namespace

}  // namespace gt4
