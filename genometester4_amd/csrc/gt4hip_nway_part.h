/* gt4hip_nway_part.h -- N-way: constants, key samples, tile boundaries (partition), tiles cut in two, the ragged count
 * table's index kernels.  Included by gt4hip_nway_body.h inside namespace gt4::<anon>::km8 / km32; no include guard. */

constexpr int NWAY_MAX = GT4_KM;  /* lists per launch */
#define GT4_NWAY_SAMPLE 128
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
#define GT4_NWAY_MARGIN 5.0 /* standard deviations of a tile's size kept free at the first try.  2.5 (27 samples per tile instead of 24, 7 % of the tiles cut in two) measured 36.4 ms against 30.1: fuller tiles give the service wavefront records of its own */
#define GT4_NWAY_LIMIT 48
#define GT4_NWAY_TRY0 32
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
#define GT4_NWAY_LEAD 1
#define GT4_NWAY_LEAD_BITS 16
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

constexpr u32 NWAY_SPLIT_BLOCK = 1024; /* tiles per block of k_nway_need / _emit */

/* k_nway_sample_counts, k_nway_bracket_bases, k_nway_partition_rows, k_nway_need for this number of lists per launch */
#include GT4_KM_ROWS


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
                                                                 u32 n_buckets, u32 nch, u64 *__restrict__ out, u32 *flag)
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
  u64 s0 = 0, s1 = 0;
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
  }
  out[at * NWAY_PSTRIDE + NWAY_MAX] = row_lo;
  out[at * NWAY_PSTRIDE + NWAY_MAX + 1] = nway_bucket_consts (row_lo, pivot, n_buckets) | clustered;
  out[(at + 1) * NWAY_PSTRIDE + NWAY_MAX] = pivot + 1ull; /* (pivot < hi_key: the right piece holds a larger key) */
  out[(at + 1) * NWAY_PSTRIDE + NWAY_MAX + 1] = nway_bucket_consts (pivot + 1ull, hi_key, n_buckets) | clustered;
  if (s0 > nch || s1 > nch) atomicOr (flag, 1u);
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
