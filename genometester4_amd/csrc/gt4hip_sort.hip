/*
 * gt4hip_sort.hip -- glistmaker's table step on the device: packed k-mer words are
 * sorted and equal words folded into (word, number of occurrences) records, i.e. a sorted list as
 * every set operation of this library consumes and gt4_write_union collates.
 *
 * What it restates: wordtable_sort + wordtable_find_frequencies (reference src/word-table.c:217-260)
 * on top of hybridInPlaceRadixSort256 (src/utils.c:127-198: in-place MSD radix sort, 8-bit digits,
 * insertion sort below 32 words; only the digits below 2 * wordlength bits are visited).
 *
 *   K8 k_radix_hist     per block of 2048 words: how many words carry each value of the pass's 8-bit digit
 *      k_radix_scan_*   exclusive prefix of those counts in (digit, block) order = where every block's
 *                       words of every digit go
 *   K9 k_radix_scatter  stable placement: eight rounds of 256 words per block; inside a wavefront the
 *                       words of one digit find each other with eight ballots, wavefronts are ordered
 *                       through per-wavefront digit counts in LDS, rounds through running counts
 *   K10 k_fold_*        heads of the runs of equal words (word differs from its left neighbour) ->
 *                       their positions, compacted in order; count = distance to the next head
 *
 * LSD order (least significant digit first, every pass stable), ceil (2k / 8) passes, two buffers.
 * An HBM-bound streaming sort: 16 bytes moved per word and pass.
 */
#include "gt4hip_device.h"
#include "gt4hip_host.h"

#include <string.h>

namespace gt4 {

namespace {

constexpr int RADIX_NT = 256;
constexpr int RADIX_ITEMS = 8;
constexpr int RADIX_TILE = RADIX_NT * RADIX_ITEMS;

__global__ __launch_bounds__ (RADIX_NT) void k_radix_hist (const u64 *__restrict__ in, u64 n, u32 shift, u32 *__restrict__ hist, u32 n_blocks)
{
  __shared__ u32 h[256];
  h[threadIdx.x] = 0;
  __syncthreads ();
  const u64 base = (u64) blockIdx.x * RADIX_TILE;
#pragma unroll
  for (int r = 0; r < RADIX_ITEMS; r++) {
    const u64 i = base + (u64) r * RADIX_NT + threadIdx.x;
    if (i < n) atomicAdd (&h[(u32) (in[i] >> shift) & 255u], 1u);
  }
  __syncthreads ();
  hist[(u64) threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x]; /* digit-major: the scan walks one digit's blocks in a row */
}

/* one block per digit: exclusive prefix of its row in place, row total to totals[digit] */
__global__ __launch_bounds__ (1024) void k_radix_scan_rows (u32 *__restrict__ hist, u32 n_blocks, u64 *__restrict__ totals)
{
  __shared__ u32 wsum[16];
  __shared__ u64 carry_s;
  u32 *row = hist + (u64) blockIdx.x * n_blocks;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads ();
  for (u32 b0 = 0; b0 < n_blocks; b0 += 1024) {
    const u32 i = b0 + threadIdx.x;
    const u32 v = i < n_blocks ? row[i] : 0u;
    const u32 incl = dpp_inclusive_scan_u32 (v);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads ();
    u32 before = 0, all = 0;
    for (int w = 0; w < 16; w++) {
      const u32 s = wsum[w];
      before += w < wid ? s : 0u;
      all += s;
    }
    const u64 c = carry_s;
    /* offsets of one digit stay below 2^32 as long as the whole array does (checked by the host) */
    if (i < n_blocks) row[i] = (u32) (c + before + incl - v);
    __syncthreads ();
    if (threadIdx.x == 0) carry_s = c + all;
    __syncthreads ();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

__global__ void k_radix_scan_totals (u64 *totals)
{
  if (threadIdx.x) return;
  u64 run = 0;
  for (int d = 0; d < 256; d++) {
    const u64 v = totals[d];
    totals[d] = run;
    run += v;
  }
}

__global__ __launch_bounds__ (RADIX_NT) void k_radix_scatter (const u64 *__restrict__ in, u64 *__restrict__ out, u64 n, u32 shift, const u32 *__restrict__ hist,
                                                             const u64 *__restrict__ totals, u32 n_blocks)
{
  constexpr int NW = RADIX_NT / WAVE;
  __shared__ u32 running[256];  /* words of each digit placed by earlier rounds of this block */
  __shared__ u32 wcnt[NW][256]; /* words of each digit per wavefront, this round */
  __shared__ u64 gbase[256];    /* where this block's words of each digit start in `out` */
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  running[tid] = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) wcnt[w][tid] = 0;
  gbase[tid] = totals[tid] + hist[(u64) tid * n_blocks + blockIdx.x];
  __syncthreads ();
  const u64 base = (u64) blockIdx.x * RADIX_TILE;
  for (int r = 0; r < RADIX_ITEMS; r++) {
    const u64 i = base + (u64) r * RADIX_NT + tid;
    const bool in_range = i < n;
    const u64 key = in_range ? in[i] : 0;
    const u32 d = (u32) (key >> shift) & 255u;
    /* lanes of this wavefront with the same digit (out-of-range lanes match nobody) */
    u64 m = __builtin_amdgcn_ballot_w64 (in_range);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const u64 bal = __builtin_amdgcn_ballot_w64 ((d >> b) & 1u);
      m &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const u32 below = (u32) __popcll (m & ((1ull << lane) - 1ull));
    if (in_range && below == 0) wcnt[wid][d] = (u32) __popcll (m);
    __syncthreads ();
    u32 pos = running[d] + below;
    for (int w = 0; w < NW; w++) pos += w < wid ? wcnt[w][d] : 0u;
    const u64 dst = gbase[d] + pos;
    __syncthreads ();
    {
      u32 t = 0;
#pragma unroll
      for (int w = 0; w < NW; w++) {
        t += wcnt[w][tid];
        wcnt[w][tid] = 0;
      }
      running[tid] += t;
    }
    if (in_range) out[dst] = key;
    __syncthreads ();
  }
}

/* ---- folding equal words */

constexpr int FOLD_NT = 256;
constexpr int FOLD_ITEMS = 8;
constexpr int FOLD_TILE = FOLD_NT * FOLD_ITEMS;

__device__ __forceinline__ bool is_head (const u64 *__restrict__ w, u64 i, u64 n) { return i < n && (i == 0 || w[i] != w[i - 1]); }

__global__ __launch_bounds__ (FOLD_NT) void k_fold_count (const u64 *__restrict__ w, u64 n, u64 *__restrict__ block_heads)
{
  __shared__ u32 ws[FOLD_NT / WAVE];
  u32 c = 0;
  const u64 base = (u64) blockIdx.x * FOLD_TILE;
#pragma unroll
  for (int r = 0; r < FOLD_ITEMS; r++) c += is_head (w, base + (u64) r * FOLD_NT + threadIdx.x, n) ? 1u : 0u;
  const u32 s = dpp_wave_sum_u32 (c);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads ();
  if (threadIdx.x == 0) {
    u32 t = 0;
    for (int i = 0; i < FOLD_NT / WAVE; i++) t += ws[i];
    block_heads[blockIdx.x] = t;
  }
}

/* exclusive prefix of block_heads in place (one block walks the array), total to *total */
__global__ __launch_bounds__ (1024) void k_fold_scan (u64 *__restrict__ block_heads, u64 n_blocks, u64 *total)
{
  __shared__ u64 wsum[16];
  __shared__ u64 carry_s;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads ();
  for (u64 b0 = 0; b0 < n_blocks; b0 += 1024) {
    const u64 i = b0 + threadIdx.x;
    const u64 v = i < n_blocks ? block_heads[i] : 0;
    const u64 incl = wave_inclusive_scan (v, lane);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads ();
    u64 before = 0, all = 0;
    for (int w = 0; w < 16; w++) {
      const u64 s = wsum[w];
      before += w < wid ? s : 0;
      all += s;
    }
    const u64 c = carry_s;
    if (i < n_blocks) block_heads[i] = c + before + incl - v;
    __syncthreads ();
    if (threadIdx.x == 0) carry_s = c + all;
    __syncthreads ();
  }
  if (threadIdx.x == 0) *total = carry_s;
}

/* position of every head, compacted in order (row-major over the block's rounds: ascending index) */
__global__ __launch_bounds__ (FOLD_NT) void k_fold_positions (const u64 *__restrict__ w, u64 n, const u64 *__restrict__ block_heads, u64 *__restrict__ head_pos)
{
  constexpr int NW = FOLD_NT / WAVE;
  __shared__ u32 cnt[FOLD_ITEMS * NW];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const u64 base = (u64) blockIdx.x * FOLD_TILE;
  u64 masks[FOLD_ITEMS];
#pragma unroll
  for (int r = 0; r < FOLD_ITEMS; r++) {
    masks[r] = __builtin_amdgcn_ballot_w64 (is_head (w, base + (u64) r * FOLD_NT + tid, n));
    if (lane == 0) cnt[r * NW + wid] = (u32) __popcll (masks[r]);
  }
  __syncthreads ();
  const u64 out0 = block_heads[blockIdx.x];
#pragma unroll
  for (int r = 0; r < FOLD_ITEMS; r++) {
    u32 before = 0;
    for (int q = 0; q < r * NW + wid; q++) before += cnt[q];
    if ((masks[r] >> lane) & 1ull) head_pos[out0 + before + (u32) __popcll (masks[r] & ((1ull << lane) - 1ull))] = base + (u64) r * FOLD_NT + tid;
  }
}

/* record j = (word at head j, distance to head j + 1); the count is stored in 32 bits as the
 * reference does (`freqs[wi] = count`, src/word-table.c:245-251) */
__global__ void k_fold_records (const u64 *__restrict__ w, u64 n, const u64 *__restrict__ head_pos, u64 n_heads, u32 *__restrict__ rec)
{
  const u64 step = (u64) gridDim.x * blockDim.x;
  for (u64 j = (u64) blockIdx.x * blockDim.x + threadIdx.x; j < n_heads; j += step) {
    const u64 p = head_pos[j], q = j + 1 < n_heads ? head_pos[j + 1] : n;
    const u64 key = w[p];
    rec[3 * j] = (u32) key;
    rec[3 * j + 1] = (u32) (key >> 32);
    rec[3 * j + 2] = (u32) (q - p);
  }
}

}  // namespace

}  // namespace gt4

using namespace gt4;

/* Sorts n 64-bit words in device memory ascending; `tmp` holds n more.  The result is in `words`. */
static int radix_sort_device (gt4hip_context *ctx, u64 *words, u64 *tmp, uint64_t n, uint32_t word_length)
{
  if (n < 2) return GT4HIP_OK;
  if (n >= (1ull << 32)) return gt4hip_fail (ctx, GT4HIP_EINVAL, "gt4hip_sort_words: at most 2^32 - 1 words per call (%llu given)", (unsigned long long) n);
  const uint32_t bits = word_length >= 32 ? 64 : 2 * word_length;
  const uint32_t passes = (bits + 7) / 8;
  const uint32_t n_blocks = (uint32_t) ((n + RADIX_TILE - 1) / RADIX_TILE);
  u32 *hist = NULL;
  u64 *totals = NULL;
  if (gt4hip_dev_alloc (ctx, (void **) &hist, (size_t) 256 * n_blocks * 4) != hipSuccess ||
      gt4hip_dev_alloc (ctx, (void **) &totals, 256 * 8) != hipSuccess) {
    if (hist) hipFree (hist);
    return gt4hip_fail (ctx, GT4HIP_ENOMEM, "gt4hip_sort_words: workspace allocation failed");
  }
  u64 *src = words, *dst = tmp;
  hipStream_t st = ctx->stream;
  for (uint32_t p = 0; p < passes; p++) {
    const uint32_t shift = 8 * p;
    hipLaunchKernelGGL (k_radix_hist, dim3 (n_blocks), dim3 (RADIX_NT), 0, st, src, n, shift, hist, n_blocks);
    hipLaunchKernelGGL (k_radix_scan_rows, dim3 (256), dim3 (1024), 0, st, hist, n_blocks, totals);
    hipLaunchKernelGGL (k_radix_scan_totals, dim3 (1), dim3 (64), 0, st, totals);
    hipLaunchKernelGGL (k_radix_scatter, dim3 (n_blocks), dim3 (RADIX_NT), 0, st, src, dst, n, shift, hist, totals, n_blocks);
    u64 *const t = src;
    src = dst;
    dst = t;
  }
  hipError_t e = hipGetLastError ();
  if (e == hipSuccess && src != words) e = hipMemcpyAsync (words, src, (size_t) n * 8, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize (st);
  hipFree (hist);
  hipFree (totals);
  if (e != hipSuccess) return gt4hip_fail (ctx, GT4HIP_EHIP, "gt4hip_sort_words: %s", hipGetErrorString (e));
  return GT4HIP_OK;
}

extern "C" int gt4hip_sort_words (gt4hip_context *ctx, void *device_words, uint64_t n_words, uint32_t word_length)
{
  if (!ctx || (n_words && !device_words) || !word_length || word_length > 32) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  if (n_words < 2) return GT4HIP_OK;
  u64 *tmp = NULL;
  if (gt4hip_dev_alloc (ctx, (void **) &tmp, (size_t) n_words * 8) != hipSuccess) return gt4hip_fail (ctx, GT4HIP_ENOMEM, "gt4hip_sort_words: %llu bytes of scratch", (unsigned long long) n_words * 8);
  const int rc = radix_sort_device (ctx, (u64 *) device_words, tmp, n_words, word_length);
  hipFree (tmp);
  return rc;
}

/* sorted device words -> list of (word, occurrences); `tmp` holds n_words u64 of scratch */
static int fold_sorted_words (gt4hip_context *ctx, const u64 *words, u64 *tmp, uint64_t n_words, uint32_t word_length, gt4hip_list **out)
{
  hipStream_t st = ctx->stream;
  int rc = GT4HIP_OK;
  gt4hip_list *l = NULL;
  hipError_t e;
  /* block head counts, then head positions */
  const uint64_t n_blocks = (n_words + FOLD_TILE - 1) / FOLD_TILE;
  u64 *block_heads = NULL;
  if (gt4hip_dev_alloc (ctx, (void **) &block_heads, (size_t) (n_blocks + 1) * 8) != hipSuccess) rc = gt4hip_fail (ctx, GT4HIP_ENOMEM, "gt4hip_words_to_list: workspace");
  if (!rc) {
    hipLaunchKernelGGL (k_fold_count, dim3 ((unsigned) n_blocks), dim3 (FOLD_NT), 0, st, words, n_words, block_heads);
    hipLaunchKernelGGL (k_fold_scan, dim3 (1), dim3 (1024), 0, st, block_heads, n_blocks, ctx->scratch);
    hipLaunchKernelGGL (k_fold_positions, dim3 ((unsigned) n_blocks), dim3 (FOLD_NT), 0, st, words, n_words, block_heads, tmp);
    e = hipMemcpyAsync (ctx->scratch_host, ctx->scratch, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize (st);
    if (e != hipSuccess) rc = gt4hip_fail (ctx, GT4HIP_EHIP, "gt4hip_words_to_list: %s", hipGetErrorString (e));
  }
  if (!rc) {
    const uint64_t n_heads = ctx->scratch_host[0];
    rc = gt4hip_list_new (ctx, n_heads, word_length, &l);
    if (!rc) {
      u64 g = (n_heads + 255) / 256;
      if (g > 8192) g = 8192;
      if (g < 1) g = 1;
      hipLaunchKernelGGL (k_fold_records, dim3 ((unsigned) g), dim3 (256), 0, st, words, n_words, tmp, n_heads, (u32 *) l->dev);
      e = hipStreamSynchronize (st);
      if (e != hipSuccess) rc = gt4hip_fail (ctx, GT4HIP_EHIP, "gt4hip_words_to_list: %s", hipGetErrorString (e));
    }
  }
  if (block_heads) hipFree (block_heads);
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
  if (gt4hip_dev_alloc (ctx, (void **) &tmp, (size_t) n_words * 8) != hipSuccess)
    return gt4hip_fail (ctx, GT4HIP_ENOMEM, "gt4hip_device_words_to_list: %llu bytes of scratch", (unsigned long long) n_words * 8);
  hipEventRecord (ctx->ev[0], ctx->stream);
  int rc = radix_sort_device (ctx, (u64 *) device_words, tmp, n_words, word_length);
  hipEventRecord (ctx->ev[1], ctx->stream);
  if (!rc) rc = fold_sorted_words (ctx, (const u64 *) device_words, tmp, n_words, word_length, out);
  hipEventRecord (ctx->ev[2], ctx->stream);
  hipStreamSynchronize (ctx->stream);
  float ms = 0;
  if (hipEventElapsedTime (&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess) ctx->sort_ms = ms;
  if (hipEventElapsedTime (&ms, ctx->ev[1], ctx->ev[2]) == hipSuccess) ctx->fold_ms = ms;
  hipFree (tmp);
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
