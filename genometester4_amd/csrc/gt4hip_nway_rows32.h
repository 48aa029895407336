/* gt4hip_nway_rows32.h -- nine to THIRTY-TWO lists per launch: the partition's row kernels and k_nway_need (a partition
 * row is 34 entries: lists in groups of eight, a half-wavefront per tile).  Included by gt4hip_nway_part.h (GT4_KM_ROWS). */
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

/* flag[0]: a tile needs more than two pieces (or the table is not monotone); flag[1]: tiles whose samples look
 * clustered; flag[2]: tiles cut in two */
/* (many lists: a partition row is 34 entries -- one HALF-wavefront per tile reads its two rows side by side, 256 bytes
 * per load; a thread per tile read them 272 bytes apart: 0.63 ms per launch, 3.2 ms of a 46 ms union of 32 lists) */
__global__ __launch_bounds__ (NWAY_SPLIT_BLOCK) void k_nway_need (const u64 *__restrict__ part, u32 num_tiles, u32 nch, u32 *__restrict__ need, u32 *__restrict__ block_sums, u32 *flag)
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
    const u32 si = dpp_inclusive_scan_u32 (sl);
    const u32 s0 = (u32) __builtin_amdgcn_readlane ((int) si, 31), s1 = (u32) __builtin_amdgcn_readlane ((int) si, 63) - s0;
    const u64 bad = __builtin_amdgcn_ballot_w64 (!mono);
    const bool bad_h = half ? (bad >> 32) != 0 : (u32) bad != 0u;
    const u32 slots = half ? s1 : s0;
    u32 v = 0;
    if (in && li == 0) {
      v = slots <= nch ? 1u : 2u;
      if (bad_h || slots > 2 * nch - 2 * NWAY_MAX) atomicOr (flag, 1u);
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
