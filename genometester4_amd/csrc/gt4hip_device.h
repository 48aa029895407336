/* gt4hip_device.h -- device helpers shared by the kernel files (gt4hip_kernels.hip: pair merge;
 * gt4hip_kway.hip: N-way tile merge): record access, wave scans, the chained scan of tile totals
 * (scanner wavefront + offset resolve), tile write-out.  gfx950 only, wave64. */
#ifndef GT4HIP_DEVICE_H
#define GT4HIP_DEVICE_H

#include "gt4hip_internal.h"

namespace gt4 {
namespace {

constexpr int WAVE = 64;

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__ ((ext_vector_type (4)));

/* ------------------------------------------------------------------ record access */

/* record i of a packed list viewed as dwords: key = words 3i, 3i+1; count = word 3i+2
 * (reference src/word-map.h:89-99: u64 at +0, u32 at +8, stride 12) */
__device__ __forceinline__ u64 load_key (const u32 *__restrict__ rec, u64 i)
{
  const u32 *p = rec + 3 * i;
  return (u64) p[0] | ((u64) p[1] << 32);
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

/* Inclusive prefix sum over the 64 lanes with DPP (no LDS crossbar, no dependent ds_bpermute
 * chain: ~12 VALU instructions instead of ~1000 cycles of shuffles).  gfx9 data-parallel
 * primitives: row_shr:n inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 carry the
 * row totals into the following rows; lanes without a source contribute 0.  Lane 63 ends up
 * with the wave total. */
__device__ __forceinline__ u32 dpp_inclusive_scan_u32 (u32 v)
{
  v += (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x111, 0xf, 0xf, false); /* row_shr:1 */
  v += (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x112, 0xf, 0xf, false); /* row_shr:2 */
  v += (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x114, 0xf, 0xf, false); /* row_shr:4 */
  v += (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x118, 0xf, 0xf, false); /* row_shr:8 */
  v += (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x142, 0xa, 0xf, false); /* row_bcast:15 -> rows 1, 3 */
  v += (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x143, 0xc, 0xf, false); /* row_bcast:31 -> rows 2, 3 */
  return v;
}

__device__ __forceinline__ u32 dpp_wave_sum_u32 (u32 v)
{
  return (u32) __builtin_amdgcn_readlane ((int) dpp_inclusive_scan_u32 (v), WAVE - 1);
}

/* Exclusive prefix of the chunk ballots' popcounts (up to 128 chunks: two per lane, packed into
 * the halves of one dword for a single scan) and the empty sentinel chunk behind them; returns the
 * tile total.  Every lane of the calling wavefront takes part. */
template <int NCH>
__device__ __forceinline__ u32 chunk_scan (u64 *km, u32 *cp, int lane)
{
  static_assert (NCH <= 2 * WAVE, "two chunks per lane");
  const u32 v0 = lane < NCH ? (u32) __popcll (km[lane]) : 0u;
  const u32 v1 = (NCH > WAVE && lane + WAVE < NCH) ? (u32) __popcll (km[NCH > WAVE ? lane + WAVE : 0]) : 0u;
  const u32 incl = dpp_inclusive_scan_u32 (v0 | (v1 << 16));
  const u32 last = (u32) __builtin_amdgcn_readlane ((int) incl, WAVE - 1);
  const u32 t0 = last & 0xffffu, total = t0 + (last >> 16);
  if (lane < NCH) cp[lane] = (incl & 0xffffu) - v0;
  if (NCH > WAVE && lane + WAVE < NCH) cp[lane + WAVE] = t0 + (incl >> 16) - v1;
  if (lane == 0) {
    cp[NCH] = total;
    km[NCH] = 0;
  }
  return total;
}

/* ------------------------------------------------------------------ tile descriptors (chained scan) */

/* Two-level chained scan of the tiles' output counts.  Per output stream s, zeroed before every
 * launch:
 *   agg[s][t]      u32, one per tile: bit 31 = published, low bits = records tile t keeps
 *   carry[s][r]    u64, one per row of 64 tiles: bit 63 = published, low bits = records kept by
 *                  all tiles of rows 0..r-1
 * Workers publish agg.  ONE scanner wavefront per stream walks the rows in order, sums each
 * complete row and publishes the running carry.  A tile's global output offset is carry[its row]
 * plus the counts of the tiles before it in its own row -- one coalesced 64-word load, summed by
 * the worker itself.  Every word is written once by a single relaxed agent-scope store and read
 * by relaxed agent-scope loads: value and flag travel in the same naturally aligned word, so no
 * fence is needed (cdna_hip_programming.md Guideline 16, form R2). */
constexpr u32 AGG_READY = 1u << 31;
constexpr u64 CARRY_READY = 1ull << 63;

__device__ __forceinline__ void publish_u32 (u32 *p, u32 v) { __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void publish_u64 (u64 *p, u64 v) { __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 peek_u32 (u32 *p) { return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 peek_u64 (u64 *p) { return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

/* (diagnostics build, `make prof`: PROF (...) and GT4_PROFILE_PHASES are in gt4hip_internal.h)
 * one asm statement with its own wait, fenced from the scheduler (cdna_hip_programming.md section 7, In-kernel stamps) */
#define GT4_STAMP_TID GT4_PROFILE_PHASES
#define PHASE_STAMP(i) PROF (do { __builtin_amdgcn_sched_barrier (0); u64 t_; asm volatile ("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier (0); if (tid == GT4_STAMP_TID) { ph[i] += t_ - t_last; } t_last = t_; } while (0))

constexpr u32 SPIN_LIMIT = 1u << 22; /* bounded: ~seconds; sets ctl->error instead of hanging */
#define GT4_SCAN_ROWS 16
constexpr int SCAN_ROWS = GT4_SCAN_ROWS;        /* rows of 64 tiles a scanner wavefront keeps in flight */

/* The scanner: one wavefront per stream.  Loads SCAN_ROWS x 64 tile counts at once (so that its
 * rate is set by L2 bandwidth, not by one round trip per row), waits for each row to be complete,
 * publishes the carry into the next row. */
__device__ void scanner_wave (u32 *agg, u64 *carry_out, u64 num_tiles, PairControl *ctl, int lane, u32 spin_limit)
{
  /* the scanner shares its SIMD with worker wavefronts and is the one serial resource of the
   * kernel: it must win the instruction arbitration */
  __builtin_amdgcn_s_setprio (3);
  u64 carry = 0;
  const u64 rows = (num_tiles + WAVE - 1) / WAVE;
  if (lane == 0) publish_u64 (&carry_out[0], CARRY_READY);
  /* two batches of SCAN_ROWS rows in flight: while one is summed and published, the loads of the
   * next are already under way (a batch costs one memory round trip otherwise) */
  u32 v[SCAN_ROWS], w[SCAN_ROWS];
  auto load_batch = [&] (u32 (&dst)[SCAN_ROWS], u64 r0) {
#pragma unroll
    for (int j = 0; j < SCAN_ROWS; j++) {
      const u64 idx = (r0 + j) * WAVE + lane;
      dst[j] = (r0 + j < rows && idx < num_tiles) ? peek_u32 (&agg[idx]) : AGG_READY;
    }
  };
  load_batch (v, 0);
  for (u64 r0 = 0; r0 < rows; r0 += SCAN_ROWS) {
    const int n = rows - r0 < (u64) SCAN_ROWS ? (int) (rows - r0) : SCAN_ROWS;
    load_batch (w, r0 + SCAN_ROWS);
    int done = 0;
    u32 spins = 0;
PROF (
    u64 st_rounds = 0, st_first = 0;
)
    for (;;) {
PROF (
      const int done_before = done;
)
      /* retire, in order, every row that is complete */
#pragma unroll
      for (int j = 0; j < SCAN_ROWS; j++) {
        if (j == done && j < n && __all ((v[j] & AGG_READY) != 0)) {
          carry += dpp_wave_sum_u32 (v[j] & ~AGG_READY);
          if (lane == 0) publish_u64 (&carry_out[r0 + j + 1], CARRY_READY | carry);
          done++;
        }
      }
PROF (
      st_rounds++;
      if (spins == 0) st_first += (u64) (done - done_before);
      if (done >= n && lane == 0) {
        atomicAdd (&ctl->resolve_stats[5], st_rounds);
        atomicAdd (&ctl->resolve_stats[6], st_first);
        atomicAdd (&ctl->resolve_stats[7], (u64) n);
      }
)
      if (done >= n) break;
      /* somebody else gave up (the host reruns the call on the two-pass path): give up at this look too */
      if ((spins & 63u) == 63u && peek_u32 (&ctl->error)) spins = spin_limit;
      if (++spins > spin_limit) {
        if (lane == 0) atomicOr (&ctl->error, 4u);
        return;
      }
      /* one round trip re-reads the missing words of ALL pending rows: at the frontier the
       * scanner must advance several rows per round trip to keep up with the workers */
#pragma unroll
      for (int j = 0; j < SCAN_ROWS; j++) {
        const u64 idx = (r0 + j) * WAVE + lane;
        if (j >= done && j < n && !(v[j] & AGG_READY)) v[j] = peek_u32 (&agg[idx]);
      }
    }
#pragma unroll
    for (int j = 0; j < SCAN_ROWS; j++) v[j] = w[j];
  }
}

/* The scanner split over several wavefronts of the scanner workgroup, for launches with more rows
 * than one wavefront can both sum and chain (2e6 tiles of the small geometry: 3e4 rows in ~12 ms).
 * SUMMER wavefronts (any number) take the rows round-robin, sum each complete row and publish
 * rowsum[row] (64-bit, flag in bit 63); ONE chainer wavefront reads 64 row sums per load, turns every
 * leading stretch of published ones into carries with a wave scan and publishes them -- the serial
 * chain advances up to 64 rows per memory round trip instead of one row per DPP reduction. */
__device__ void summer_wave (u32 *agg, u64 *rowsum, u64 num_tiles, PairControl *ctl, int lane, u32 spin_limit, u32 first, u32 stride)
{
  __builtin_amdgcn_s_setprio (2);
  const u64 rows = (num_tiles + WAVE - 1) / WAVE;
  if (first >= rows) return;
  const u64 mine = (rows - first + stride - 1) / stride; /* rows first, first + stride, ... */
  u32 v[SCAN_ROWS], w[SCAN_ROWS];
  auto load_batch = [&] (u32 (&dst)[SCAN_ROWS], u64 i0) {
#pragma unroll
    for (int j = 0; j < SCAN_ROWS; j++) {
      const u64 r = first + (i0 + j) * stride, idx = r * WAVE + lane;
      dst[j] = (i0 + j < mine && idx < num_tiles) ? peek_u32 (&agg[idx]) : AGG_READY;
    }
  };
  load_batch (v, 0);
  for (u64 i0 = 0; i0 < mine; i0 += SCAN_ROWS) {
    const int n = mine - i0 < (u64) SCAN_ROWS ? (int) (mine - i0) : SCAN_ROWS;
    load_batch (w, i0 + SCAN_ROWS);
    u32 pending = n == 32 ? 0xffffffffu : ((1u << n) - 1u); /* rows of the batch not yet published (any order) */
    u32 spins = 0;
    for (;;) {
#pragma unroll
      for (int j = 0; j < SCAN_ROWS; j++) {
        if (((pending >> j) & 1u) && __all ((v[j] & AGG_READY) != 0)) {
          const u32 sum = dpp_wave_sum_u32 (v[j] & ~AGG_READY);
          if (lane == 0) publish_u64 (&rowsum[first + (i0 + j) * stride], CARRY_READY | sum);
          pending &= ~(1u << j);
        }
      }
      if (!pending) break;
      if ((spins & 63u) == 63u && peek_u32 (&ctl->error)) spins = spin_limit;
      if (++spins > spin_limit) {
        if (lane == 0) atomicOr (&ctl->error, 4u);
        return;
      }
#pragma unroll
      for (int j = 0; j < SCAN_ROWS; j++) {
        const u64 idx = (first + (i0 + j) * stride) * WAVE + lane;
        if (((pending >> j) & 1u) && !(v[j] & AGG_READY)) v[j] = peek_u32 (&agg[idx]);
      }
    }
#pragma unroll
    for (int j = 0; j < SCAN_ROWS; j++) v[j] = w[j];
  }
}

__device__ void chainer_wave (u64 *rowsum, u64 *carry_out, u64 num_tiles, PairControl *ctl, int lane, u32 spin_limit)
{
  __builtin_amdgcn_s_setprio (3);
  const u64 rows = (num_tiles + WAVE - 1) / WAVE;
  if (lane == 0) publish_u64 (&carry_out[0], CARRY_READY);
  u64 base = 0;
  u64 nv = lane < (int) (rows < (u64) WAVE ? rows : WAVE) ? peek_u64 (&rowsum[lane]) : CARRY_READY; /* next block, loaded ahead */
  for (u64 b0 = 0; b0 < rows; b0 += WAVE) {
    const u64 r = b0 + lane;
    const bool in = r < rows;
    u64 v = nv;
    {
      const u64 rn = r + WAVE;
      nv = rn < rows ? peek_u64 (&rowsum[rn]) : CARRY_READY;
    }
    int done = 0;
    u32 spins = 0;
    for (;;) {
      const u64 ready = __builtin_amdgcn_ballot_w64 ((v & CARRY_READY) != 0);
      const int lead = ~ready ? (int) __builtin_ctzll (~ready) : WAVE; /* rows of the block whose sums are all in, from its start */
      if (lead > done) {
        const u64 inc = wave_inclusive_scan (lane < lead ? (v & ~CARRY_READY) : 0ull, lane);
        if (lane >= done && lane < lead && in) publish_u64 (&carry_out[r + 1], CARRY_READY | (base + inc));
        done = lead;
        if (done >= WAVE) {
          base += (u64) __shfl ((u32) inc, WAVE - 1, WAVE) | ((u64) __shfl ((u32) (inc >> 32), WAVE - 1, WAVE) << 32);
          break;
        }
      }
      if ((spins & 63u) == 63u && peek_u32 (&ctl->error)) spins = spin_limit;
      if (++spins > spin_limit) {
        if (lane == 0) atomicOr (&ctl->error, 4u);
        return;
      }
      if (in && !(v & CARRY_READY)) v = peek_u64 (&rowsum[r]);
    }
  }
}

/* The scanner workgroup's wavefront `sub` of `n_sub` working for one output stream: alone it sums
 * and chains (scanner_wave); with company, wavefront 0 chains and the others sum. */
__device__ __forceinline__ void scanner_part (u32 *agg, u64 *rowsum, u64 *carry_out, u64 num_tiles, PairControl *ctl, int lane, u32 spin_limit,
                                              u32 sub, u32 n_sub)
{
  if (n_sub <= 1) scanner_wave (agg, carry_out, num_tiles, ctl, lane, spin_limit);
  else if (sub == 0) chainer_wave (rowsum, carry_out, num_tiles, ctl, lane, spin_limit);
  else summer_wave (agg, rowsum, num_tiles, ctl, lane, spin_limit, sub - 1, n_sub - 1);
}

/* A tile's global output offset: carry of its row + counts of the tiles before it in the row.
 * `a` and `c` are the values of an earlier, speculative load of the same words (or 0).
 * LOOK-BACK (round 3): when the row's carry is not published yet the wavefront does not wait for the
 * scanner but goes back up to RESOLVE_LOOKBACK rows itself: carry[row - r] plus the complete rows in
 * between (64 counts per load, one DPP sum each) is the same number.  The scanner may then lag that
 * many rows behind the workers without anybody waiting for it -- with one staging area (the
 * any-combination pair kernel, the N-way kernel) the chain's round trip otherwise bounds the time per
 * tile (measured: the N-way kernel stripped of all its ranking and output still took 21.6 of 30 ms). */
/* GT4_RESOLVE_LOOKBACK: defined by the including kernel file -- the pair kernels take 3 (-u -d -c 3: 24.14 -> 23.48 ms); the
 * N-way kernel's service wavefront loses 9 % to the extra loads and takes 0 */
constexpr int RESOLVE_LOOKBACK = GT4_RESOLVE_LOOKBACK;

__device__ __forceinline__ u64 resolve_offset (u32 *agg, u64 *carry, u64 tile, int lane, u32 a, u64 c, PairControl *ctl, u32 spin_limit)
{
  const u64 row = tile / WAVE;
  const u32 pos = (u32) (tile % WAVE);
  u32 *const wa = &agg[row * WAVE + lane];
  u64 *const wc = &carry[row];
  const bool mine = (u32) lane < pos;
  u32 spins = 0;
PROF (
  const bool agg_ok0 = __all (!mine || (a & AGG_READY) != 0), carry_ok0 = (c & CARRY_READY) != 0;
)
  u64 back = 0; /* counts of the complete rows between the carry used and this row */
  for (;;) {
    const bool own_ok = __all (!mine || (a & AGG_READY) != 0);
    if (own_ok && (c & CARRY_READY)) break;
    if (own_ok && RESOLVE_LOOKBACK > 0) {
      /* the rows before this one, nearest first: all their counts and one carry, asked for together */
      u32 ar[RESOLVE_LOOKBACK > 0 ? RESOLVE_LOOKBACK : 1];
      u64 cr[RESOLVE_LOOKBACK > 0 ? RESOLVE_LOOKBACK : 1];
#pragma unroll
      for (int r = 0; r < RESOLVE_LOOKBACK; r++) {
        const bool has = row > (u64) r;
        ar[r] = has ? peek_u32 (&agg[(row - 1 - r) * WAVE + lane]) : 0u;
        cr[r] = has ? peek_u64 (&carry[row - 1 - r]) : 0ull;
      }
      u64 sum = 0;
      bool found = false, rows_ok = true;
#pragma unroll
      for (int r = 0; r < RESOLVE_LOOKBACK; r++) {
        if (found || !rows_ok || row <= (u64) r) continue; /* uniform */
        rows_ok = __all ((ar[r] & AGG_READY) != 0);
        if (!rows_ok) continue;
        sum += dpp_wave_sum_u32 (ar[r] & ~AGG_READY);
        if (cr[r] & CARRY_READY) {
          found = true;
          c = cr[r];
          back = sum;
        }
      }
      if (found) break;
    }
    if (++spins > spin_limit) {
      if (lane == 0) atomicOr (&ctl->error, 1u);
      break;
    }
    /* once any wait of the launch has given up (the scanner stops publishing then) every other wait
     * ends at its next look instead of spinning to its own bound, so the rest of the launch runs at
     * its normal pace: the offset returned is not above the true one (stores stay inside the output)
     * and is not used -- the host discards the outputs and reruns the call on the two-pass path */
    if ((spins & 15u) == 0 && peek_u32 (&ctl->error)) break;
    if (spins > 1) __builtin_amdgcn_s_sleep (1);
    if (mine && !(a & AGG_READY)) a = peek_u32 (wa);
    if (!(c & CARRY_READY)) c = peek_u64 (wc);
  }
PROF (
  if (lane == 0 && (tile & 63) == 17) { /* sample 1 in 64 so that the statistics do not perturb the run */
    atomicAdd (&ctl->resolve_stats[0], 1ull);
    atomicAdd (&ctl->resolve_stats[1], (u64) spins);
    atomicAdd (&ctl->resolve_stats[3], agg_ok0 ? 0ull : 1ull);
    atomicAdd (&ctl->resolve_stats[4], carry_ok0 ? 0ull : 1ull);
  }
)
  return (c & ~CARRY_READY) + back + dpp_wave_sum_u32 (mine ? (a & ~AGG_READY) : 0u);
}

/* Cache-policy bits of the merge kernels' streaming buffer loads and stores (1 sc0, 2 nt, 16 sc1): every record is
 * read once and written once, and says so -- NON-TEMPORAL both ways.  Measured (profiles/round5/r5_cache_policy.log,
 * three alternating runs each): the headline intersection 10.93 -> 10.81 ms per launch, the 8-way union 28.23 -> 27.85,
 * config 2 23.34 -> 23.18; sc1 (write-through, line dropped from L2) stores cost config 2 10 %. */
#define GT4_STORE_AUX 2
#define GT4_LOAD_AUX 2

/* Tile write-out: `tot` packed records from an LDS staging slot (16-byte aligned) to the output
 * list at record offset `excl`, as 16-byte buffer stores (dword alignment suffices; the
 * range-checked descriptor drops the dwords past the last record of the partial last chunk). */
template <int NT>
__device__ __forceinline__ void write_out_tile (u32 *out_rec, u64 excl, u32 tot, const u32 *slot, int tid)
{
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc ((void *) (out_rec + 3 * excl), 0, (int) (12 * tot), 0x00020000);
  const u32 chunks = (3 * tot + 3) >> 2;
  for (u32 c = (u32) tid; c < chunks; c += NT) {
    const u32x4 w = *reinterpret_cast<const u32x4 *> (slot + 4 * c);
    __builtin_amdgcn_raw_buffer_store_b128 (w, r, 16 * c, 0, GT4_STORE_AUX);
  }
}


/* The same with the offset in BYTES (the one wavefront that resolves it multiplies, not all sixteen)
 * and without a single per-lane branch: K guarded-by-a-scalar steps, the range-checked descriptor
 * drops what lies past the last record, the LDS read is clamped into the staged records.  The
 * scalar unit is shared by the CU's wavefronts: every scalar instruction here is paid sixteen times. */
template <int NT, int K>
__device__ __forceinline__ void write_out_fixed (u32 *out_rec, u64 excl_bytes, u32 tot, const u32 *slot, int tid)
{
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc ((void *) (reinterpret_cast<char *> (out_rec) + excl_bytes), 0, (int) (12 * tot), 0x00020000);
  const u32 chunks = (3 * tot + 3) >> 2;
#pragma unroll
  for (int k = 0; k < K; k++) {
    if ((u32) k * NT >= chunks) break; /* wave-uniform */
    const u32 c = (u32) k * NT + (u32) tid;
    const u32 cr = c < chunks ? c : chunks - 1u;
    const u32x4 w = *reinterpret_cast<const u32x4 *> (slot + 4 * cr);
    __builtin_amdgcn_raw_buffer_store_b128 (w, r, 16 * c, 0, GT4_STORE_AUX);
  }
}


/* values read back from LDS are the same in every lane; say so, so that addresses, descriptors and
 * loop bounds derived from them live in SGPRs (no waterfall loops around the buffer loads) */
__device__ __forceinline__ u32 uniform32 (u32 v) { return __builtin_amdgcn_readfirstlane (v); }
__device__ __forceinline__ u64 uniform64 (u64 v)
{
  /* the builtin returns int: go through u32 or a low word >= 2^31 sign-extends into the high half */
  return (u64) uniform32 ((u32) v) | ((u64) uniform32 ((u32) (v >> 32)) << 32);
}

}  // namespace
}  // namespace gt4

#endif
