/*
 * gt4hip_sort.hip -- glistmaker's table step on the device: packed k-mer words are
 * sorted and equal words folded into (word, number of occurrences) records, i.e. a sorted list as
 * every set operation of this library consumes and gt4_write_union collates.
 *
 * What it restates: wordtable_sort + wordtable_find_frequencies (reference src/word-table.c:217-260)
 * on top of hybridInPlaceRadixSort256 (src/utils.c:127-198: in-place MSD radix sort, 8-bit digits,
 * insertion sort below 32 words; only the digits below 2 * wordlength bits are visited).
 *
 *   K8 k_radix_hist     ONE read of the words: how many carry each value of every pass's 8- or 9-bit digit
 *      k_radix_bases    exclusive prefix over the digits = where each digit's words start, per pass
 *   K9 k_radix_scatter  one launch per pass, one 8192-word tile per workgroup (by ticket): stable ranks inside
 *                       a wavefront by one ballot per digit bit (lanes with the same digit find each other) on top of
 *                       per-wavefront digit counters in LDS; the tile sorted by digit in LDS; where it goes
 *                       behind the earlier tiles by a chained scan over per-(tile, digit) state words
 *                       (aggregate published early, look-back four tiles per round trip); the tile leaves
 *                       LDS in runs of one digit
 *   K10 k_fold_count    runs of equal words that start in each 8192-word tile (a word differs from its left
 *                       neighbour) -> runs before the tile, by a chained scan (one state word per tile)
 *       k_fold_records  record = (first word of the run, distance to the next run's start)
 *
 * LSD order (least significant digit first, every pass stable), ceil (2k / 9) passes of 8 or 9 bits,
 * two buffers.
 * An HBM-bound streaming sort: 16 bytes moved per word and pass, 8 more once for the histograms.
 * Measured (1e9 random 50-bit words, MI355X): 31.9 ms in six passes (36.1 in seven of 8 bits); the first version of this file (a histogram
 * kernel per pass, words scattered 8 bytes at a time) took 100.6 ms, rocprim::radix_sort_keys 40.5 ms
 * (profiles/round3/r3_sort_experiments.log).
 */
#define GT4_RESOLVE_LOOKBACK 0 /* (no chained scan of tile totals here) */
#include "gt4hip_device.h"
#include "gt4hip_host.h"

#include <stdio.h>
#include <string.h>

namespace gt4 {

namespace {

typedef u32 u32x2 __attribute__ ((ext_vector_type (2)));

__device__ __forceinline__ u64 readlane_u64 (u64 v, int l)
{
  return (u64) (u32) __builtin_amdgcn_readlane ((int) (u32) v, l) | ((u64) (u32) __builtin_amdgcn_readlane ((int) (u32) (v >> 32), l) << 32);
}

/* ---- radix sort: one histogram kernel for all passes, then one scatter kernel per pass */

#define GT4_RADIX_NT 512
#define GT4_RADIX_ITEMS 16
constexpr int RADIX_NT = GT4_RADIX_NT;        /* threads per tile */
constexpr int RADIX_ITEMS = GT4_RADIX_ITEMS;  /* words per thread */
constexpr int RADIX_TILE = RADIX_NT * RADIX_ITEMS;
constexpr int RADIX_NW = RADIX_NT / WAVE;
constexpr int RADIX_MAX_PASSES = 8;
#define GT4_RADIX_WAVES 4 /* wavefronts per SIMD the scatter kernel's registers must leave room for: two workgroups per CU */
#define GT4_RADIX_LOOK 4
constexpr int RADIX_LOOK = GT4_RADIX_LOOK; /* earlier tiles inspected per round trip of the look-back */
constexpr int HIST_NT = 256;
constexpr int HIST_ITEMS = 16;

/* Tile states of the chained scan, one 64-bit word per (tile, digit): what the tile holds of the digit
 * (AGG) or what all tiles up to and including it hold (PREFIX), tagged with the pass so that the words
 * need no clearing between passes.  A word is published and read whole: no fences. */
constexpr u64 RADIX_AGG = 1ull << 62, RADIX_PREFIX = 2ull << 62;
constexpr u64 RADIX_VALUE = (1ull << 56) - 1;
__device__ __forceinline__ u64 radix_tag (u32 pass) { return (u64) (pass + 1) << 56; }

constexpr int RADIX_MAX_DIGITS = 512; /* a pass takes 8 or 9 bits: ceil (bits / 9) passes, as many 9-bit ones as it takes */

struct RadixPlan {
  u32 passes;
  u32 shift[RADIX_MAX_PASSES];
  u32 bits[RADIX_MAX_PASSES];
};

/* all passes' digit counts in one read of the words: LDS counters per block, flushed by global atomics */
__global__ __launch_bounds__ (HIST_NT) void k_radix_hist (const u64 *__restrict__ in, u64 n, RadixPlan plan, u64 *__restrict__ ghist)
{
  __shared__ u32 h[RADIX_MAX_PASSES][RADIX_MAX_DIGITS];
  for (u32 p = 0; p < plan.passes; p++)
    for (u32 d = threadIdx.x; d < (u32) RADIX_MAX_DIGITS; d += HIST_NT) h[p][d] = 0;
  __syncthreads ();
  const u64 chunk = (u64) HIST_NT * HIST_ITEMS;
  for (u64 base = (u64) blockIdx.x * chunk; base < n; base += (u64) gridDim.x * chunk) {
    u64 w[HIST_ITEMS];
#pragma unroll
    for (int r = 0; r < HIST_ITEMS; r++) {
      const u64 i = base + (u64) r * HIST_NT + threadIdx.x;
      w[r] = i < n ? in[i] : 0;
    }
#pragma unroll
    for (int r = 0; r < HIST_ITEMS; r++) {
      if (base + (u64) r * HIST_NT + threadIdx.x < n)
        for (u32 p = 0; p < plan.passes; p++) atomicAdd (&h[p][(u32) (w[r] >> plan.shift[p]) & ((1u << plan.bits[p]) - 1u)], 1u);
    }
  }
  __syncthreads ();
  for (u32 p = 0; p < plan.passes; p++)
    for (u32 d = threadIdx.x; d < (1u << plan.bits[p]); d += HIST_NT)
      if (h[p][d]) atomicAdd ((unsigned long long *) &ghist[p * RADIX_MAX_DIGITS + d], (unsigned long long) h[p][d]);
}

/* per pass: where the words of every digit start (exclusive prefix over the digits); block = pass */
__global__ __launch_bounds__ (RADIX_MAX_DIGITS) void k_radix_bases (u64 *__restrict__ ghist)
{
  __shared__ u64 ws[RADIX_MAX_DIGITS / WAVE];
  u64 *row = ghist + (u64) blockIdx.x * RADIX_MAX_DIGITS;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const u64 v = row[threadIdx.x];
  const u64 incl = wave_inclusive_scan (v, lane);
  if (lane == 63) ws[wid] = incl;
  __syncthreads ();
  u64 before = 0;
  for (int w = 0; w < wid; w++) before += ws[w];
  row[threadIdx.x] = before + incl - v;
}

/* One pass over a digit of B = 8 or 9 bits: tile t's words go behind everything the earlier tiles hold of
 * the same digit, in the order they came (stable).  Tiles are taken by ticket, so a tile's predecessors have
 * all STARTED (a wait may only point at workgroups that run: the last workgroups of a launch are not
 * dispatched while earlier ones occupy the CUs they are bound for).
 *   1. a wavefront owns RADIX_ITEMS * 64 consecutive words, read 64 at a time; the tile's digit totals are
 *      counted at once (LDS atomics) and published (AGG);
 *   2. the lanes holding the same digit find each other with B ballots; the wavefront's private digit
 *      counters (LDS, 16 bits) give every word its rank among the wavefront's words of that digit;
 *   3. prefix over wavefronts and digits = every word's place in the tile sorted by digit; the words go
 *      there (LDS);
 *   4. one thread per digit looks back over the earlier tiles' states (RADIX_LOOK per round trip) until it
 *      meets a PREFIX; the tile's own PREFIX is published;
 *   5. the tile leaves LDS in digit order: runs of one digit go to consecutive addresses. */
/* (cache-policy switches of the words' streaming load and scattered store, as in gt4hip_device.h) */
template <int B>
__global__ __launch_bounds__ (RADIX_NT, GT4_RADIX_WAVES) void k_radix_scatter (const u64 *__restrict__ in, u64 *__restrict__ out, u64 n, u32 pass, u32 shift, const u64 *__restrict__ gbase,
                                                                              u64 *__restrict__ state, u32 *__restrict__ ticket, u32 *__restrict__ err, u32 spin_limit)
{
  constexpr int ND = 1 << B, NDW = ND / WAVE; /* digits; wavefronts that own one digit per lane */
  static_assert (ND <= RADIX_NT && ND <= RADIX_MAX_DIGITS && RADIX_TILE <= 65535, "a digit per thread; 16-bit places");
  __shared__ u64 keys[RADIX_TILE];
  __shared__ unsigned short wcnt[RADIX_NW][ND]; /* per wavefront: words of each digit so far; later: where the wavefront's words of the digit start in the sorted tile */
  __shared__ u32 hcnt[ND];          /* words of each digit in the tile */
  __shared__ u64 gofs[ND];          /* address of the tile's first word of each digit in `out`, minus its place in the sorted tile */
  __shared__ u32 wtot[NDW];
  __shared__ u32 tile_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) tile_s = atomicAdd (ticket, 1u);
  for (int i = tid; i < RADIX_NW * ND / 2; i += RADIX_NT) reinterpret_cast<u32 *> (&wcnt[0][0])[i] = 0;
  if (tid < ND) hcnt[tid] = 0;
  __syncthreads ();
  const u64 tile = tile_s;
  const u64 base = tile * RADIX_TILE;
  const u32 nv = n - base < (u64) RADIX_TILE ? (u32) (n - base) : (u32) RADIX_TILE;
  u64 key[RADIX_ITEMS];
  {
    /* a range-checked descriptor over the tile: no per-lane bounds, no addresses in registers */
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc ((void *) (in + base), 0, (int) (8 * nv), 0x00020000);
#pragma unroll
    for (int r = 0; r < RADIX_ITEMS; r++) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64 (rs, 8 * lane, 8 * WAVE * (wid * RADIX_ITEMS + r), 0);
      key[r] = (u64) v.x | ((u64) v.y << 32);
    }
  }
  /* what the tile holds of every digit, as early as it can be known (the stable ranks below take several
   * times as long): the later tiles look back for it */
#pragma unroll
  for (int r = 0; r < RADIX_ITEMS; r++) {
    const u32 q = (u32) (wid * RADIX_ITEMS + r) * WAVE + (u32) lane;
    if (q < nv) atomicAdd (&hcnt[(u32) (key[r] >> shift) & (u32) (ND - 1)], 1u);
  }
  __syncthreads ();
  u32 cnt_d = 0;
  if (tid < ND) {
    cnt_d = hcnt[tid];
    __hip_atomic_store (&state[tile * ND + tid], (tile == 0 ? RADIX_PREFIX : RADIX_AGG) | radix_tag (pass) | (u64) cnt_d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  u32 rk[RADIX_ITEMS / 2]; /* 16 bits each */
#pragma unroll
  for (int r = 0; r < RADIX_ITEMS; r++) {
    const u32 q = (u32) (wid * RADIX_ITEMS + r) * WAVE + (u32) lane;
    const bool valid = q < nv;
    const u32 d = (u32) (key[r] >> shift) & (u32) (ND - 1);
    u64 m = __builtin_amdgcn_ballot_w64 (valid);
#pragma unroll
    for (int b = 0; b < B; b++) {
      const u64 bal = __builtin_amdgcn_ballot_w64 ((d >> b) & 1u);
      m &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const u32 below = __builtin_amdgcn_mbcnt_hi ((u32) (m >> 32), __builtin_amdgcn_mbcnt_lo ((u32) m, 0u));
    /* every lane of the group reads the counter, then its first lane adds the group (LDS executes a
     * wavefront's accesses in order) */
    const u32 old = valid ? wcnt[wid][d] : 0u;
    if (valid && below == 0) wcnt[wid][d] = (unsigned short) (old + (u32) __popcll (m));
    rk[r / 2] = (r & 1) ? rk[r / 2] | ((old + below) << 16) : old + below;
  }
  /* digit d = tid: its first place in the sorted tile */
  u32 ls = 0;
  if (tid < ND) {
    const u32 incl = dpp_inclusive_scan_u32 (cnt_d);
    if (lane == 63) wtot[wid] = incl;
    ls = incl - cnt_d;
  }
  __syncthreads ();
  if (tid < ND) {
    for (int w = 0; w < wid; w++) ls += wtot[w];
    u32 run = ls;
#pragma unroll
    for (int w = 0; w < RADIX_NW; w++) { /* -> where wavefront w's words of the digit start */
      const u32 c = wcnt[w][tid];
      wcnt[w][tid] = (unsigned short) run;
      run += c;
    }
  }
  __syncthreads ();
  /* the tile sorted by digit, in LDS */
#pragma unroll
  for (int r = 0; r < RADIX_ITEMS; r++) {
    const u32 q = (u32) (wid * RADIX_ITEMS + r) * WAVE + (u32) lane;
    const u32 d = (u32) (key[r] >> shift) & (u32) (ND - 1);
    if (q < nv) keys[(u32) wcnt[wid][d] + ((rk[r / 2] >> (16 * (r & 1))) & 0xffffu)] = key[r];
  }
  /* look back: what the earlier tiles hold of digit tid */
  if (tid < ND) {
    u64 excl = 0;
    if (tile > 0) {
      /* RADIX_LOOK earlier tiles per round trip: the states are asked for together and summed in order
       * up to the first PREFIX (a serial walk meets ~20 AGG states per tile: 20 dependent round trips) */
      const u64 mine = radix_tag (pass) >> 56;
      auto ready = [&] (u64 v) { return (v >> 62) != 0 && ((v >> 56) & 63u) == mine; };
      bool done = false;
      for (u64 j = tile; !done; j -= RADIX_LOOK) {
        u64 v[RADIX_LOOK];
#pragma unroll
        for (int i = 0; i < RADIX_LOOK; i++)
          v[i] = j >= (u64) (1 + i) ? __hip_atomic_load (&state[(j - 1 - i) * ND + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (RADIX_PREFIX | radix_tag (pass));
#pragma unroll
        for (int i = 0; i < RADIX_LOOK; i++) {
          if (done) continue;
          /* bounded: a predecessor that never publishes (a fault, a device shared with a stuck process) must
           * not hang the sort.  The wait that gives up raises *err (the host returns GT4HIP_EHIP; the output
           * is garbage) and goes on as if it had met a PREFIX, so its own PREFIX lets the successors drain;
           * every other wait notices the flag at its next look. */
          for (u32 spins = 0; !ready (v[i]);) {
            if (++spins >= spin_limit || ((spins & 255u) == 0 && peek_u32 (err))) {
              atomicOr (err, 1u);
              v[i] = RADIX_PREFIX | radix_tag (pass);
              break;
            }
            v[i] = __hip_atomic_load (&state[(j - 1 - i) * ND + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          excl += v[i] & RADIX_VALUE;
          done = (v[i] & RADIX_PREFIX) != 0;
        }
      }
      __hip_atomic_store (&state[tile * ND + tid], RADIX_PREFIX | radix_tag (pass) | (excl + cnt_d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    gofs[tid] = gbase[pass * RADIX_MAX_DIGITS + tid] + excl - ls;
  }
  __syncthreads ();
#pragma unroll
  for (int r = 0; r < RADIX_ITEMS; r++) {
    const u32 q = (u32) r * RADIX_NT + (u32) tid;
    if (q < nv) {
      const u64 k = keys[q];
      out[gofs[(u32) (k >> shift) & (u32) (ND - 1)] + q] = k; /* (plain: non-temporal scattered stores measured slower, profiles/round5/r5_cache_policy.log) */
    }
  }
}

/* ---- folding equal words */

constexpr int FOLD_NT = 512;
constexpr int FOLD_ITEMS = 16;
constexpr int FOLD_TILE = FOLD_NT * FOLD_ITEMS;
constexpr int FOLD_NW = FOLD_NT / WAVE;
constexpr int FOLD_STRETCH = FOLD_ITEMS * WAVE; /* consecutive words of one wavefront */

/* A tile's words, 64 consecutive ones per wavefront and round (as the scatter kernel reads them), and
 * which of them start a run: bit r of the result = word r of this lane differs from the word before it
 * (the previous lane's, the previous round's last lane's, or -- first word of the wavefront -- the one
 * in memory before it; word 0 of all starts a run). */
__device__ __forceinline__ u32 fold_heads (const u64 *__restrict__ w, u64 n, u64 base, u32 nv, int lane, int wid, u64 (&word)[FOLD_ITEMS])
{
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc ((void *) (w + base), 0, (int) (8 * nv), 0x00020000);
#pragma unroll
  for (int r = 0; r < FOLD_ITEMS; r++) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64 (rs, 8 * lane, 8 * WAVE * (wid * FOLD_ITEMS + r), 0);
    word[r] = (u64) v.x | ((u64) v.y << 32);
  }
  const u64 first = base + (u64) wid * FOLD_STRETCH;
  u64 before = 0; /* (uniform) */
  if (first > 0 && first < n) before = w[first - 1];
  u32 flags = 0;
#pragma unroll
  for (int r = 0; r < FOLD_ITEMS; r++) {
    const u32 q = (u32) (wid * FOLD_ITEMS + r) * WAVE + (u32) lane;
    const u32 plo = (u32) __shfl_up ((u32) word[r], 1, WAVE), phi = (u32) __shfl_up ((u32) (word[r] >> 32), 1, WAVE);
    u64 prev = (u64) plo | ((u64) phi << 32);
    if (lane == 0) prev = r ? readlane_u64 (word[r ? r - 1 : 0], WAVE - 1) : before;
    const bool head = q < nv && (base + q == 0 || word[r] != prev);
    flags |= head ? 1u << r : 0u;
  }
  return flags;
}

/* Pass 1: runs that start in each tile -> the number of runs before the tile (chained scan over one
 * state word per tile: a whole wavefront looks back, 64 tiles per round trip); tiles by ticket. */
__global__ __launch_bounds__ (FOLD_NT) void k_fold_count (const u64 *__restrict__ w, u64 n, u64 *__restrict__ state, u64 *__restrict__ tile_excl, u64 tiles, u32 *__restrict__ ticket,
                                                        u64 *__restrict__ total, u32 *__restrict__ err, u32 spin_limit)
{
  __shared__ u32 ws[FOLD_NW];
  __shared__ u32 tile_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) tile_s = atomicAdd (ticket, 1u);
  __syncthreads ();
  const u64 tile = tile_s;
  const u64 base = tile * FOLD_TILE;
  const u32 nv = n - base < (u64) FOLD_TILE ? (u32) (n - base) : (u32) FOLD_TILE;
  u64 word[FOLD_ITEMS];
  const u32 flags = fold_heads (w, n, base, nv, lane, wid, word);
  const u32 c = dpp_wave_sum_u32 ((u32) __popc (flags));
  if (lane == 0) ws[wid] = c;
  __syncthreads ();
  if (wid != 0) return;
  u32 cnt = 0;
  for (int i = 0; i < FOLD_NW; i++) cnt += ws[i];
  if (lane == 0) __hip_atomic_store (&state[tile], (tile == 0 ? RADIX_PREFIX : RADIX_AGG) | (u64) cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u64 excl = 0;
  for (u64 j = tile; j > 0;) {
    const bool in = j >= (u64) (1 + lane);
    u64 v;
    u64 pm, need; /* lanes whose tile's state is a PREFIX; lanes up to the nearest of them */
    u32 spins = 0; /* bounded as the scatter kernel's waits are: *err, then on as if a PREFIX had been met */
    do {
      v = in ? __hip_atomic_load (&state[j - 1 - lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : RADIX_PREFIX;
      if (++spins >= spin_limit || ((spins & 255u) == 0 && peek_u32 (err))) { /* uniform */
        if (lane == 0) atomicOr (err, 1u);
        v = RADIX_PREFIX;
      }
      pm = __builtin_amdgcn_ballot_w64 ((v & RADIX_PREFIX) != 0);
      need = pm ? (2ull << __builtin_ctzll (pm)) - 1ull : ~0ull;
    } while (__builtin_amdgcn_ballot_w64 ((v >> 62) == 0) & need);
    excl += wave_sum (((need >> lane) & 1ull) ? (v & RADIX_VALUE) : 0ull);
    if (pm) break;
    j = j > (u64) WAVE ? j - WAVE : 0;
  }
  if (lane == 0) {
    if (tile) __hip_atomic_store (&state[tile], RADIX_PREFIX | (excl + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tile_excl[tile] = excl;
    if (tile + 1 == tiles) {
      tile_excl[tiles] = excl + cnt;
      *total = excl + cnt;
    }
  }
}

/* Pass 2: record j = (word at the start of run j, length of the run).  The length is the distance to
 * the next start: in the same 64 words, in the wavefront's later rounds, in the later wavefronts -- or
 * behind the tile, found by wavefront 0 (tiles without a start are skipped by their counts).  The count
 * is stored in 32 bits as the reference does (`freqs[wi] = count`, src/word-table.c:245-251). */
__global__ __launch_bounds__ (FOLD_NT) void k_fold_records (const u64 *__restrict__ w, u64 n, const u64 *__restrict__ tile_excl, u64 tiles, u32 *__restrict__ rec)
{
  __shared__ u32 wcount[FOLD_NW];
  __shared__ u32 wfirst[FOLD_NW]; /* first start in the wavefront's words (position in the tile), or FOLD_TILE */
  __shared__ u64 next_s;          /* first start behind the tile */
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const u64 tile = blockIdx.x;
  const u64 base = tile * FOLD_TILE;
  const u32 nv = n - base < (u64) FOLD_TILE ? (u32) (n - base) : (u32) FOLD_TILE;
  u64 word[FOLD_ITEMS];
  const u32 flags = fold_heads (w, n, base, nv, lane, wid, word);
  u32 mine = 0, first = (u32) FOLD_TILE;
#pragma unroll
  for (int r = FOLD_ITEMS - 1; r >= 0; r--) {
    const u64 m = __builtin_amdgcn_ballot_w64 ((flags >> r) & 1u);
    mine += (u32) __popcll (m);
    if (m) first = (u32) (wid * FOLD_ITEMS + r) * WAVE + (u32) __builtin_ctzll (m);
  }
  if (lane == 0) {
    wcount[wid] = mine;
    wfirst[wid] = first;
  }
  if (wid == 0) {
    /* the first start behind the tile: tiles without one are skipped, the first tile with one is
     * searched 64 words at a time */
    u64 u = tile + 1;
    while (u < tiles && tile_excl[u + 1] == tile_excl[u]) u++;
    u64 nx = n;
    if (u < tiles) {
      for (u64 i = u * FOLD_TILE;; i += WAVE) {
        const u64 at = i + lane;
        const bool head = at < n && w[at] != w[at - 1]; /* (at >= FOLD_TILE here) */
        const u64 m = __builtin_amdgcn_ballot_w64 (head);
        if (m) {
          nx = i + (u64) __builtin_ctzll (m);
          break;
        }
      }
    }
    if (lane == 0) next_s = nx;
  }
  __syncthreads ();
  u32 rank = 0; /* starts in the earlier wavefronts */
  for (int i = 0; i < wid; i++) rank += wcount[i];
  u64 nh = next_s; /* the first start behind this wavefront's words */
  for (int i = FOLD_NW - 1; i > wid; i--) nh = wfirst[i] < (u32) FOLD_TILE ? base + wfirst[i] : nh;
  const u64 out0 = tile_excl[tile] + rank;
  u32 running = mine;
#pragma unroll
  for (int r = FOLD_ITEMS - 1; r >= 0; r--) {
    const u64 m = __builtin_amdgcn_ballot_w64 ((flags >> r) & 1u);
    if (!m) continue; /* uniform */
    running -= (u32) __popcll (m);
    const u64 pos0 = base + (u64) (wid * FOLD_ITEMS + r) * WAVE;
    if ((flags >> r) & 1u) {
      const u64 above = lane == WAVE - 1 ? 0ull : m >> (lane + 1);
      const u64 next = above ? pos0 + (u64) lane + 1ull + (u64) __builtin_ctzll (above) : nh;
      const u64 j = out0 + running + __builtin_amdgcn_mbcnt_hi ((u32) (m >> 32), __builtin_amdgcn_mbcnt_lo ((u32) m, 0u));
      rec[3 * j] = (u32) word[r];
      rec[3 * j + 1] = (u32) (word[r] >> 32);
      rec[3 * j + 2] = (u32) (next - (pos0 + (u64) lane));
    }
    nh = pos0 + (u64) __builtin_ctzll (m);
  }
}

}  // namespace

}  // namespace gt4

using namespace gt4;

/* Sorts n 64-bit words in device memory ascending; `tmp` holds n more.  The sorted words end up in
 * `words` or in `tmp` (odd number of passes): *result says where. */
static int radix_sort_device (gt4hip_context *ctx, u64 *words, u64 *tmp, uint64_t n, uint32_t word_length, u64 **result)
{
  *result = words;
  if (n < 2) return GT4HIP_OK;
  if (n >= RADIX_VALUE) return gt4hip_fail (ctx, GT4HIP_EINVAL, "gt4hip_sort_words: %llu words", (unsigned long long) n);
  const uint32_t bits = word_length >= 32 ? 64 : 2 * word_length;
  /* ceil (bits / 9) passes; as many of them take 9 bits as it takes to cover the word, the others 8
   * (k = 25: 50 bits = 9 + 9 + 8 + 8 + 8 + 8, six passes instead of seven) */
  RadixPlan plan;
  memset (&plan, 0, sizeof plan);
  plan.passes = (bits + 8) / 9; /* digits of nine bits as far as they go, eight for the rest */
  {
    const uint32_t nine = bits > 8 * plan.passes ? bits - 8 * plan.passes : 0;
    uint32_t at = 0;
    for (uint32_t p = 0; p < plan.passes; p++) {
      plan.shift[p] = at;
      plan.bits[p] = p < nine ? 9 : 8;
      at += plan.bits[p];
    }
  }
  const uint32_t passes = plan.passes;
  const uint64_t tiles = (n + RADIX_TILE - 1) / RADIX_TILE;
  if (tiles >= (1ull << 32)) return gt4hip_fail (ctx, GT4HIP_EINVAL, "gt4hip_sort_words: %llu words", (unsigned long long) n);
  /* workspace: digit bases of every pass, a ticket per pass, the tile states */
  const size_t head = (size_t) RADIX_MAX_PASSES * RADIX_MAX_DIGITS * 8 + 64;
  const size_t state_bytes = (size_t) tiles * RADIX_MAX_DIGITS * 8;
  char *ws = NULL;
  void *ws_owner = NULL;
  if (gt4hip_block_alloc (ctx, head + state_bytes, (void **) &ws, &ws_owner))
    return gt4hip_fail (ctx, GT4HIP_ENOMEM, "gt4hip_sort_words: workspace of %llu bytes", (unsigned long long) (head + state_bytes));
  u64 *ghist = (u64 *) ws;
  u32 *tickets = (u32 *) (ws + (size_t) RADIX_MAX_PASSES * RADIX_MAX_DIGITS * 8);
  u32 *err = tickets + 15; /* (the 64 bytes behind the digit bases: a ticket per pass, then the error word) */
  const u32 spin_limit = ctx->spin_limit ? ctx->spin_limit : SPIN_LIMIT;
  u64 *state = (u64 *) (ws + head);
  hipStream_t st = ctx->stream;
  hipError_t e = hipMemsetAsync (ws, 0, head + state_bytes, st);
  u64 hb = (n + (u64) HIST_NT * HIST_ITEMS - 1) / ((u64) HIST_NT * HIST_ITEMS);
  if (hb > (u64) ctx->n_cus * 8) hb = (u64) ctx->n_cus * 8;
  hipLaunchKernelGGL (k_radix_hist, dim3 ((unsigned) hb), dim3 (HIST_NT), 0, st, words, n, plan, ghist);
  hipLaunchKernelGGL (k_radix_bases, dim3 (passes), dim3 (RADIX_MAX_DIGITS), 0, st, ghist);
  u64 *src = words, *dst = tmp;
  for (uint32_t p = 0; p < passes; p++) {
    if (plan.bits[p] == 9) hipLaunchKernelGGL (k_radix_scatter<9>, dim3 ((unsigned) tiles), dim3 (RADIX_NT), 0, st, src, dst, n, p, plan.shift[p], ghist, state, tickets + p, err, spin_limit);
    else hipLaunchKernelGGL (k_radix_scatter<8>, dim3 ((unsigned) tiles), dim3 (RADIX_NT), 0, st, src, dst, n, p, plan.shift[p], ghist, state, tickets + p, err, spin_limit);
    u64 *const t = src;
    src = dst;
    dst = t;
  }
  if (e == hipSuccess) e = hipGetLastError ();
  if (e == hipSuccess) e = hipMemcpyAsync (ctx->scratch_host, err, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize (st);
  gt4hip_block_free (ws_owner);
  if (e != hipSuccess) return gt4hip_fail (ctx, GT4HIP_EHIP, "gt4hip_sort_words: %s", hipGetErrorString (e));
  if ((u32) ctx->scratch_host[0]) return gt4hip_fail (ctx, GT4HIP_EHIP, "gt4hip_sort_words: a chained-scan wait gave up (device shared with a stuck workgroup?)");
  *result = src;
  return GT4HIP_OK;
}

extern "C" int gt4hip_sort_words (gt4hip_context *ctx, void *device_words, uint64_t n_words, uint32_t word_length)
{
  if (!ctx || (n_words && !device_words) || !word_length || word_length > 32) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  if (n_words < 2) return GT4HIP_OK;
  u64 *tmp = NULL;
  void *tmp_owner = NULL;
  if (gt4hip_block_alloc (ctx, (size_t) n_words * 8, (void **) &tmp, &tmp_owner)) return gt4hip_fail (ctx, GT4HIP_ENOMEM, "gt4hip_sort_words: %llu bytes of scratch", (unsigned long long) n_words * 8);
  u64 *res = NULL;
  int rc = radix_sort_device (ctx, (u64 *) device_words, tmp, n_words, word_length, &res);
  if (!rc && res != (u64 *) device_words) {
    hipError_t e = hipMemcpyAsync (device_words, res, (size_t) n_words * 8, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize (ctx->stream);
    if (e != hipSuccess) rc = gt4hip_fail (ctx, GT4HIP_EHIP, "gt4hip_sort_words: %s", hipGetErrorString (e));
  }
  gt4hip_block_free (tmp_owner);
  return rc;
}

/* sorted device words -> list of (word, occurrences) */
static int fold_sorted_words (gt4hip_context *ctx, const u64 *words, uint64_t n_words, uint32_t word_length, gt4hip_list **out)
{
  hipStream_t st = ctx->stream;
  int rc = GT4HIP_OK;
  gt4hip_list *l = NULL;
  hipError_t e;
  const uint64_t tiles = (n_words + FOLD_TILE - 1) / FOLD_TILE;
  if (tiles >= (1ull << 32)) return gt4hip_fail (ctx, GT4HIP_EINVAL, "gt4hip_words_to_list: %llu words", (unsigned long long) n_words);
  /* workspace: ticket, tile states, runs before every tile (+ the total) */
  char *ws = NULL;
  void *ws_owner = NULL;
  const size_t bytes = 64 + (size_t) tiles * 8 + (size_t) (tiles + 1) * 8;
  if (gt4hip_block_alloc (ctx, bytes, (void **) &ws, &ws_owner)) return gt4hip_fail (ctx, GT4HIP_ENOMEM, "gt4hip_words_to_list: workspace");
  u64 *state = (u64 *) (ws + 64), *tile_excl = state + tiles;
  e = hipMemsetAsync (ws, 0, 64 + (size_t) tiles * 8, st);
  hipLaunchKernelGGL (k_fold_count, dim3 ((unsigned) tiles), dim3 (FOLD_NT), 0, st, words, n_words, state, tile_excl, tiles, (u32 *) ws, ctx->scratch, (u32 *) ws + 15,
                      ctx->spin_limit ? ctx->spin_limit : SPIN_LIMIT);
  if (e == hipSuccess) e = hipMemcpyAsync (ctx->scratch_host, ctx->scratch, 8, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync (ctx->scratch_host + 1, (u32 *) ws + 15, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize (st);
  if (e != hipSuccess) rc = gt4hip_fail (ctx, GT4HIP_EHIP, "gt4hip_words_to_list: %s", hipGetErrorString (e));
  else if ((u32) ctx->scratch_host[1]) rc = gt4hip_fail (ctx, GT4HIP_EHIP, "gt4hip_words_to_list: a chained-scan wait gave up (device shared with a stuck workgroup?)");
  if (!rc) {
    const uint64_t n_heads = ctx->scratch_host[0];
    rc = gt4hip_list_new (ctx, n_heads, word_length, &l);
    if (!rc) {
      hipLaunchKernelGGL (k_fold_records, dim3 ((unsigned) tiles), dim3 (FOLD_NT), 0, st, words, n_words, tile_excl, tiles, (u32 *) l->dev);
      e = hipStreamSynchronize (st);
      if (e != hipSuccess) rc = gt4hip_fail (ctx, GT4HIP_EHIP, "gt4hip_words_to_list: %s", hipGetErrorString (e));
    }
  }
  gt4hip_block_free (ws_owner);
  if (rc) {
    if (l) gt4hip_list_free (l);
    return rc;
  }
  *out = l;
  return GT4HIP_OK;
}

/* The same for words that are in device memory already (sorted in place, then folded): what a k-mer
 * extraction kernel upstream would hand over; also what bench.py --workload sort times. */
extern "C" int gt4hip_device_words_to_list (gt4hip_context *ctx, void *device_words, uint64_t n_words, uint32_t word_length, gt4hip_list **out)
{
  if (!ctx || !out || (n_words && !device_words) || !word_length || word_length > 32) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  *out = NULL;
  if (!n_words) return gt4hip_list_new (ctx, 0, word_length, out);
  u64 *tmp = NULL;
  void *tmp_owner = NULL;
  if (gt4hip_block_alloc (ctx, (size_t) n_words * 8, (void **) &tmp, &tmp_owner))
    return gt4hip_fail (ctx, GT4HIP_ENOMEM, "gt4hip_device_words_to_list: %llu bytes of scratch", (unsigned long long) n_words * 8);
  hipEventRecord (ctx->ev[0], ctx->stream);
  u64 *res = NULL;
  int rc = radix_sort_device (ctx, (u64 *) device_words, tmp, n_words, word_length, &res);
  hipEventRecord (ctx->ev[1], ctx->stream);
  /* (the words are the caller's scratch from here on: sorted in place or not, they are folded from wherever the last pass left them) */
  if (!rc) rc = fold_sorted_words (ctx, res, n_words, word_length, out);
  hipEventRecord (ctx->ev[2], ctx->stream);
  hipStreamSynchronize (ctx->stream);
  float ms = 0;
  if (hipEventElapsedTime (&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess) ctx->sort_ms = ms;
  if (hipEventElapsedTime (&ms, ctx->ev[1], ctx->ev[2]) == hipSuccess) ctx->fold_ms = ms;
  gt4hip_block_free (tmp_owner);
  return rc;
}

extern "C" int gt4hip_words_to_list (gt4hip_context *ctx, const uint64_t *host_words, uint64_t n_words, uint32_t word_length, gt4hip_list **out)
{
  if (!ctx || !out || (n_words && !host_words) || !word_length || word_length > 32) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  *out = NULL;
  if (!n_words) return gt4hip_list_new (ctx, 0, word_length, out);
  u64 *words = NULL;
  if (gt4hip_dev_alloc (ctx, (void **) &words, (size_t) n_words * 8) != hipSuccess)
    return gt4hip_fail (ctx, GT4HIP_ENOMEM, "gt4hip_words_to_list: device buffers for %llu words", (unsigned long long) n_words);
  hipError_t e = hipMemcpyAsync (words, host_words, (size_t) n_words * 8, hipMemcpyHostToDevice, ctx->stream);
  int rc = e == hipSuccess ? GT4HIP_OK : gt4hip_fail (ctx, GT4HIP_EHIP, "gt4hip_words_to_list: upload failed: %s", hipGetErrorString (e));
  if (!rc) rc = gt4hip_device_words_to_list (ctx, words, n_words, word_length, out);
  hipFree (words);
  return rc;
}
