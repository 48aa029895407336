/* gt4hip_nway_rows8.h -- up to EIGHT lists per launch: the partition's row kernels and k_nway_need (a partition row is
 * ten entries: a lane per entry, a thread per tile).  Included by gt4hip_nway_part.h (GT4_KM_ROWS). */
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

__device__ __forceinline__ u32 nway_tile_slots (const u64 *__restrict__ part, u64 t, bool *mono)
{
  u64 slots = 0;
  for (int i = 0; i < NWAY_MAX; i++) {
    const u64 a = part[t * NWAY_PSTRIDE + i], b = part[(t + 1) * NWAY_PSTRIDE + i];
    *mono &= b >= a;
    slots += (b - a + NWAY_HS - 1) / NWAY_HS;
  }
  return slots > 0xffffffffull ? 0xffffffffu : (u32) slots;
}

/* flag[0]: a tile needs more than two pieces (or the table is not monotone); flag[1]: tiles whose samples look
 * clustered; flag[2]: tiles cut in two */
__global__ __launch_bounds__ (NWAY_SPLIT_BLOCK) void k_nway_need (const u64 *__restrict__ part, u32 num_tiles, u32 nch, u32 *__restrict__ need, u32 *__restrict__ block_sums, u32 *flag)
{
  __shared__ u32 ws[NWAY_SPLIT_BLOCK / WAVE];
  const u64 t = (u64) blockIdx.x * NWAY_SPLIT_BLOCK + threadIdx.x;
  u32 v = 0;
  if (t < num_tiles) {
    bool mono = true;
    const u32 slots = nway_tile_slots (part, t, &mono);
    v = slots <= nch ? 1u : 2u;
    if (!mono || slots > 2 * nch - 2 * NWAY_MAX) atomicOr (flag, 1u); /* (each half rounds every run up once more) */
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
