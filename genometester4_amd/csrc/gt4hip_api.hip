/*
 * gt4hip_api.hip -- implementation of the C ABI declared in include/gt4hip.h.
 *
 * Host-side orchestration only: contexts, HBM-resident lists, workspace, kernel sequencing and
 * timing.  All per-record work happens in gt4hip_kernels.hip.  Nothing here falls back to the CPU.
 */
#include "gt4hip_host.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#include <chrono>
#include <new>
#include <utility>
#include <vector>

using namespace gt4;

static thread_local char g_create_err[512] = ""; /* per thread: a worker's loader, merger and writer threads each create a context (found by ThreadSanitizer on the CPU harness) */

static void pool_flush (gt4hip_context *ctx);

int gt4hip_fail (gt4hip_context *ctx, int code, const char *fmt, ...)
{
  va_list ap;
  va_start (ap, fmt);
  vsnprintf (ctx ? ctx->err : g_create_err, 512, fmt, ap);
  va_end (ap);
  return code;
}

extern "C" const char *gt4hip_strerror (int code)
{
  switch (code) {
    case GT4HIP_OK: return "ok";
    case GT4HIP_EINVAL: return "invalid argument";
    case GT4HIP_ENODEVICE: return "no usable HIP device";
    case GT4HIP_ENOMEM: return "out of memory";
    case GT4HIP_ERULE: return "rule not allowed for this operation";
    case GT4HIP_EHIP: return "HIP runtime error";
    case GT4HIP_EWORDLEN: return "lists have different word lengths";
    case GT4HIP_EINTERNAL: return "internal consistency check failed";
    case GT4HIP_ECALLBACK: return "stopped by callback";
    case GT4HIP_EIO: return "file I/O failed";
    case GT4HIP_ECOMM: return "RCCL communication failed";
    default: return "unknown error";
  }
}

extern "C" const char *gt4hip_last_error (const gt4hip_context *ctx)
{
  return ctx ? ctx->err : g_create_err;
}

extern "C" int gt4hip_device_count (void)
{
  int n = 0;
  if (hipGetDeviceCount (&n) != hipSuccess) return 0;
  return n;
}

extern "C" int gt4hip_create (int device, gt4hip_context **out)
{
  if (!out || device < 0) return gt4hip_fail (NULL, GT4HIP_EINVAL, "gt4hip_create: bad arguments");
  *out = NULL;
  int n = 0;
  hipError_t e = hipGetDeviceCount (&n);
  if (e != hipSuccess || n <= 0)
    return gt4hip_fail (NULL, GT4HIP_ENODEVICE, "gt4hip_create: no HIP device (%s)", e != hipSuccess ? hipGetErrorString (e) : "count 0");
  if (device >= n) return gt4hip_fail (NULL, GT4HIP_ENODEVICE, "gt4hip_create: device %d not present (%d visible)", device, n);
  if ((e = hipSetDevice (device)) != hipSuccess)
    return gt4hip_fail (NULL, GT4HIP_ENODEVICE, "gt4hip_create: hipSetDevice(%d): %s", device, hipGetErrorString (e));
  gt4hip_context *ctx = new (std::nothrow) gt4hip_context ();
  if (!ctx) return gt4hip_fail (NULL, GT4HIP_ENOMEM, "gt4hip_create: host allocation failed");
  memset (ctx, 0, sizeof *ctx);
  ctx->device = device;
  ctx->pool = new (std::nothrow) std::vector<std::pair<void *, size_t>> ();
  ctx->pool_enabled = ctx->pool != NULL;
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties (&prop, device)) != hipSuccess) {
    delete ctx->pool;
    delete ctx;
    return gt4hip_fail (NULL, GT4HIP_ENODEVICE, "gt4hip_create: hipGetDeviceProperties: %s", hipGetErrorString (e));
  }
  ctx->n_cus = prop.multiProcessorCount;
  ctx->pool_cap = prop.totalGlobalMem / 2;
  {
    const char *e = getenv ("GT4HIP_DYNAMIC"); /* diagnostic: the whole test-suite through the other dealing */
    ctx->dynamic = e ? atoi (e) : 0; /* 0: automatic */
  }
  ctx->kway_max = 32;
  ctx->kway_enabled = 1; /* N-way unions of three lists or more take the one-pass tile kernel (gt4hip_nway.hip); option "kway": 0 the pairwise tree, 2 also two lists */
  snprintf (ctx->info, sizeof ctx->info, "%s|%s|%d|%zu", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.totalGlobalMem);
  if ((e = hipStreamCreateWithFlags (&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
    delete ctx->pool;
    delete ctx;
    return gt4hip_fail (NULL, GT4HIP_ENODEVICE, "gt4hip_create: hipStreamCreate: %s", hipGetErrorString (e));
  }
  for (int i = 0; i < 4; i++) hipEventCreate (&ctx->ev[i]);
  if (hipMalloc ((void **) &ctx->ctl, sizeof (PairControl)) != hipSuccess ||
      hipHostMalloc ((void **) &ctx->ctl_host, sizeof (PairControl), hipHostMallocDefault) != hipSuccess ||
      hipMalloc ((void **) &ctx->scratch, 64) != hipSuccess ||
      hipHostMalloc ((void **) &ctx->scratch_host, 64, hipHostMallocDefault) != hipSuccess) {
    gt4hip_destroy (ctx);
    return gt4hip_fail (NULL, GT4HIP_ENOMEM, "gt4hip_create: control block allocation failed");
  }
  *out = ctx;
  return GT4HIP_OK;
}

extern "C" void gt4hip_destroy (gt4hip_context *ctx)
{
  if (!ctx) return;
  hipSetDevice (ctx->device);
  if (ctx->stream) hipStreamSynchronize (ctx->stream);
  gt4hip_io_destroy (ctx);
  if (ctx->pool) {
    pool_flush (ctx);
    delete ctx->pool;
  }
  if (ctx->part) hipFree (ctx->part);
  if (ctx->kway_part) hipFree (ctx->kway_part);
  if (ctx->kway_cnt) hipFree (ctx->kway_cnt);
  if (ctx->kway_part2) hipFree (ctx->kway_part2);
  if (ctx->kway_need) hipFree (ctx->kway_need);
  if (ctx->desc) hipFree (ctx->desc);
  if (ctx->block_sums) hipFree (ctx->block_sums);
  if (ctx->ctl) hipFree (ctx->ctl);
  if (ctx->ctl_host) hipHostFree (ctx->ctl_host);
  if (ctx->scratch) hipFree (ctx->scratch);
  if (ctx->scratch_host) hipHostFree (ctx->scratch_host);
  for (int i = 0; i < 4; i++) if (ctx->ev[i]) hipEventDestroy (ctx->ev[i]);
  if (ctx->stream) hipStreamDestroy (ctx->stream);
  delete ctx;
}

extern "C" const char *gt4hip_device_info (const gt4hip_context *ctx)
{
  return ctx ? ctx->info : "";
}

extern "C" int gt4hip_context_device (const gt4hip_context *ctx) { return ctx ? ctx->device : -1; }

extern "C" int gt4hip_trim (gt4hip_context *ctx)
{
  if (!ctx) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
  pool_flush (ctx);
  return GT4HIP_OK;
}

extern "C" int gt4hip_device_memory (gt4hip_context *ctx, uint64_t *free_bytes, uint64_t *total_bytes)
{
  if (!ctx) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  size_t f = 0, t = 0;
  HIPCHK (ctx, hipMemGetInfo (&f, &t));
  /* what the context's pool holds can be had back at once */
  if (free_bytes) *free_bytes = (uint64_t) f + ctx->pool_bytes;
  if (total_bytes) *total_bytes = (uint64_t) t;
  return GT4HIP_OK;
}

extern "C" int gt4hip_set_option (gt4hip_context *ctx, const char *name, int64_t value)
{
  if (!ctx || !name) return GT4HIP_EINVAL;
  if (!strcmp (name, "two_pass")) ctx->two_pass = value != 0;
  else if (!strcmp (name, "pool")) {
    ctx->pool_enabled = value != 0 && ctx->pool;
    if (!ctx->pool_enabled) pool_flush (ctx);
  }
  else if (!strcmp (name, "pool_cap_mb")) {
    ctx->pool_cap = value > 0 ? (size_t) value << 20 : 0;
    if (ctx->pool_bytes > ctx->pool_cap) pool_flush (ctx);
  }
  else if (!strcmp (name, "grid")) ctx->grid_override = value;
  else if (!strcmp (name, "scan_group")) ctx->scan_group = (int) value;
  else if (!strcmp (name, "dynamic")) ctx->dynamic = (int) value;
  else if (!strcmp (name, "kway")) ctx->kway_enabled = (int) value;
  else if (!strcmp (name, "kway_max")) ctx->kway_max = value == 8 ? 8 : (value == 33 ? 33 : 32);
  else if (!strcmp (name, "kway_g")) ctx->kway_g = value;
  else if (!strcmp (name, "kway_vt")) ctx->kway_vt = value;
  else if (!strcmp (name, "spin_limit")) ctx->spin_limit = value > 0 ? (uint32_t) value : 0u;
  else if (!strcmp (name, "geom1")) ctx->force_geom = value != 0 ? 1 : 0;
  else if (!strcmp (name, "geom0")) ctx->force_geom = value != 0 ? -1 : 0;
  else return gt4hip_fail (ctx, GT4HIP_EINVAL, "unknown option %s", name);
  return GT4HIP_OK;
}

extern "C" int gt4hip_get_counter (gt4hip_context *ctx, const char *name, uint64_t *value)
{
  if (!ctx || !name || !value) return GT4HIP_EINVAL;
  if (!strcmp (name, "single_pass_fallbacks")) *value = ctx->single_pass_fallbacks;
  else if (!strcmp (name, "kway_calls")) *value = ctx->kway_calls;
  else if (!strcmp (name, "kway_overflows")) *value = ctx->kway_overflows;
  else if (!strcmp (name, "kway_declined")) *value = ctx->kway_declined;
  else if (!strcmp (name, "kway_splits")) *value = ctx->kway_splits;
  else if (!strcmp (name, "kway_shared_x100")) *value = ctx->kway_shared_x100;
  else if (!strcmp (name, "kway_width")) *value = ctx->kway_width;
  else if (!strcmp (name, "nway_kernel_us")) *value = (uint64_t) (ctx->nway_kernel_ms * 1000.0);
  else if (!strcmp (name, "nway_tiles")) *value = ctx->nway_tiles;
  else if (!strcmp (name, "nway_one_pass")) *value = (uint64_t) ctx->last_multi_one_pass;
  else if (!strcmp (name, "sort_us")) *value = (uint64_t) (ctx->sort_ms * 1000.0);
  else if (!strcmp (name, "fold_us")) *value = (uint64_t) (ctx->fold_ms * 1000.0);
  else if (!strcmp (name, "table_us")) *value = (uint64_t) (ctx->table_ms * 1000.0);
  else return gt4hip_fail (ctx, GT4HIP_EINVAL, "unknown counter %s", name);
  return GT4HIP_OK;
}

extern "C" int gt4hip_synchronize (gt4hip_context *ctx)
{
  if (!ctx) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
  return GT4HIP_OK;
}

/* ------------------------------------------------------------------ device memory */

static void pool_flush (gt4hip_context *ctx)
{
  if (!ctx->pool) return;
  for (auto &b : *ctx->pool) hipFree (b.first);
  ctx->pool->clear ();
  ctx->pool_bytes = 0;
}

/* Every device allocation of the library goes through here: when the driver is out of memory the
 * pooled blocks (freed list storage kept for reuse) are given back and the allocation is retried. */
hipError_t gt4hip_dev_alloc (gt4hip_context *ctx, void **p, size_t bytes)
{
  hipError_t e = hipMalloc (p, bytes);
  if (e != hipSuccess && ctx->pool && !ctx->pool->empty ()) {
    (void) hipGetLastError ();
    pool_flush (ctx);
    e = hipMalloc (p, bytes);
  }
  if (e != hipSuccess) (void) hipGetLastError ();
  return e;
}

/* ------------------------------------------------------------------ lists */

int gt4hip_list_new (gt4hip_context *ctx, uint64_t capacity, uint32_t word_length, gt4hip_list **out)
{
  gt4hip_list *l = new (std::nothrow) gt4hip_list ();
  if (!l) return gt4hip_fail (ctx, GT4HIP_ENOMEM, "host allocation failed");
  l->ctx = ctx;
  l->dev = NULL;
  l->n_words = capacity;
  l->capacity = capacity;
  l->word_length = word_length;
  l->owns = 1;
  /* 16 bytes of slack so that 16-byte vector loads that straddle the end stay inside the allocation */
  const size_t bytes = (((size_t) capacity * GT4HIP_RECORD_BYTES + 16) + 255) & ~(size_t) 255;
  /* reuse a pooled block that fits without wasting more than half of it */
  if (ctx->pool_enabled) {
    size_t best = (size_t) -1, best_i = 0;
    for (size_t i = 0; i < ctx->pool->size (); i++) {
      const size_t b = (*ctx->pool)[i].second;
      if (b >= bytes && b / 2 <= bytes && b < best) {
        best = b;
        best_i = i;
      }
    }
    if (best != (size_t) -1) {
      l->dev = (*ctx->pool)[best_i].first;
      l->bytes = best;
      ctx->pool_bytes -= best;
      ctx->pool->erase (ctx->pool->begin () + (long) best_i);
    }
  }
  if (!l->dev) {
    const hipError_t e = gt4hip_dev_alloc (ctx, &l->dev, bytes);
    if (e != hipSuccess) {
      delete l;
      return gt4hip_fail (ctx, GT4HIP_ENOMEM, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString (e));
    }
    l->bytes = bytes;
  }
  *out = l;
  return GT4HIP_OK;
}

extern "C" int gt4hip_list_alloc (gt4hip_context *ctx, uint64_t capacity, uint32_t word_length, gt4hip_list **out)
{
  if (!ctx || !out) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  return gt4hip_list_new (ctx, capacity, word_length, out);
}

extern "C" int gt4hip_list_upload (gt4hip_context *ctx, const void *host_records, uint64_t n_words, uint32_t word_length,
                                    gt4hip_list **out)
{
  if (!ctx || !out || (n_words && !host_records)) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  gt4hip_list *l = NULL;
  int rc = gt4hip_list_new (ctx, n_words, word_length, &l);
  if (rc) return rc;
  if (n_words) {
    rc = gt4hip_list_load (ctx, l, host_records, n_words); /* large buffers: pinned staging on the copy threads */
    if (rc) {
      gt4hip_list_free (l);
      return rc;
    }
  }
  *out = l;
  return GT4HIP_OK;
}

extern "C" int gt4hip_list_upload_index (gt4hip_context *ctx, const void *host_kmers, uint64_t n_words, uint64_t num_locations,
                                          uint32_t word_length, gt4hip_list **out)
{
  if (!ctx || !out || (n_words && !host_kmers)) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  gt4hip_list *l = NULL;
  int rc = gt4hip_list_new (ctx, n_words, word_length, &l);
  if (rc) return rc;
  if (n_words) {
    void *tmp = NULL;
    hipError_t e = gt4hip_dev_alloc (ctx, &tmp, (size_t) n_words * 16);
    if (e != hipSuccess) {
      gt4hip_list_free (l);
      return gt4hip_fail (ctx, GT4HIP_ENOMEM, "hipMalloc of %llu bytes for the index table failed", (unsigned long long) n_words * 16);
    }
    e = hipMemcpyAsync (tmp, host_kmers, (size_t) n_words * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = launch_decode_index (ctx->stream, (const unsigned long long *) tmp, n_words, num_locations, (uint32_t *) l->dev);
    if (e == hipSuccess) e = hipStreamSynchronize (ctx->stream);
    hipFree (tmp);
    if (e != hipSuccess) {
      gt4hip_list_free (l);
      return gt4hip_fail (ctx, GT4HIP_EHIP, "index upload failed: %s", hipGetErrorString (e));
    }
  }
  *out = l;
  return GT4HIP_OK;
}

extern "C" int gt4hip_list_wrap (gt4hip_context *ctx, void *device_records, uint64_t n_words, uint32_t word_length, gt4hip_list **out)
{
  if (!ctx || !out || (n_words && !device_records)) return GT4HIP_EINVAL;
  if (((uintptr_t) device_records) & 3) return gt4hip_fail (ctx, GT4HIP_EINVAL, "device records must be 4-byte aligned");
  gt4hip_list *l = new (std::nothrow) gt4hip_list ();
  if (!l) return gt4hip_fail (ctx, GT4HIP_ENOMEM, "host allocation failed");
  l->ctx = ctx;
  l->dev = device_records;
  l->n_words = n_words;
  l->capacity = n_words;
  l->word_length = word_length;
  l->owns = 0;
  *out = l;
  return GT4HIP_OK;
}

extern "C" int gt4hip_list_slice (gt4hip_context *ctx, const gt4hip_list *list, uint64_t first, uint64_t count, gt4hip_list **out)
{
  if (!ctx || !list || !out || first > list->n_words || count > list->n_words - first) return GT4HIP_EINVAL;
  return gt4hip_list_wrap (ctx, (char *) list->dev + first * GT4HIP_RECORD_BYTES, count, list->word_length, out);
}

extern "C" int gt4hip_list_download_range (gt4hip_context *ctx, const gt4hip_list *list, uint64_t first, uint64_t count, void *host)
{
  if (!ctx || !list || first > list->n_words || count > list->n_words - first || (count && !host)) return GT4HIP_EINVAL;
  if (!count) return GT4HIP_OK;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  if ((size_t) count * GT4HIP_RECORD_BYTES >= ((size_t) 32 << 20))
    return gt4hip_io_download (ctx, (const char *) list->dev + first * GT4HIP_RECORD_BYTES, host, (size_t) count * GT4HIP_RECORD_BYTES);
  HIPCHK (ctx, hipMemcpyAsync (host, (const char *) list->dev + first * GT4HIP_RECORD_BYTES, (size_t) count * GT4HIP_RECORD_BYTES,
                               hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
  return GT4HIP_OK;
}

extern "C" int gt4hip_list_download (gt4hip_context *ctx, const gt4hip_list *list, void *host)
{
  if (!list) return GT4HIP_EINVAL;
  return gt4hip_list_download_range (ctx, list, 0, list->n_words, host);
}

extern "C" void gt4hip_list_free (gt4hip_list *l)
{
  if (!l) return;
  if (l->owns && l->dev) {
    gt4hip_context *const ctx = l->ctx;
    if (ctx->pool_enabled && l->bytes <= ctx->pool_cap) {
      /* the pool never holds more than its cap: the oldest blocks go back to the driver first */
      while (ctx->pool_bytes + l->bytes > ctx->pool_cap && !ctx->pool->empty ()) {
        hipSetDevice (ctx->device);
        hipFree (ctx->pool->front ().first);
        ctx->pool_bytes -= ctx->pool->front ().second;
        ctx->pool->erase (ctx->pool->begin ());
      }
      ctx->pool->push_back (std::make_pair (l->dev, l->bytes));
      ctx->pool_bytes += l->bytes;
    } else {
      hipSetDevice (l->ctx->device);
      hipFree (l->dev);
    }
  }
  delete l;
}

extern "C" uint64_t gt4hip_list_n_words (const gt4hip_list *l) { return l ? l->n_words : 0; }
extern "C" uint32_t gt4hip_list_word_length (const gt4hip_list *l) { return l ? l->word_length : 0; }
extern "C" void *gt4hip_list_device_ptr (const gt4hip_list *l) { return l ? l->dev : NULL; }

extern "C" int gt4hip_list_set_n_words (gt4hip_list *l, uint64_t n)
{
  if (!l || n > l->capacity) return GT4HIP_EINVAL;
  l->n_words = n;
  return GT4HIP_OK;
}

static int read_scratch (gt4hip_context *ctx, unsigned n)
{
  HIPCHK (ctx, hipMemcpyAsync (ctx->scratch_host, ctx->scratch, n * sizeof (unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
  return GT4HIP_OK;
}

extern "C" int gt4hip_list_sum_counts (gt4hip_context *ctx, const gt4hip_list *l, uint64_t *sum)
{
  if (!ctx || !l || !sum) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  HIPCHK (ctx, hipMemsetAsync (ctx->scratch, 0, 64, ctx->stream));
  if (l->n_words) HIPCHK (ctx, launch_sum_counts (ctx->stream, (const uint32_t *) l->dev, l->n_words, ctx->scratch));
  int rc = read_scratch (ctx, 1);
  if (rc) return rc;
  *sum = ctx->scratch_host[0];
  return GT4HIP_OK;
}

extern "C" int gt4hip_list_is_sorted (gt4hip_context *ctx, const gt4hip_list *l, int *sorted)
{
  if (!ctx || !l || !sorted) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  HIPCHK (ctx, hipMemsetAsync (ctx->scratch, 0, 64, ctx->stream));
  if (l->n_words > 1) HIPCHK (ctx, launch_check_sorted (ctx->stream, (const uint32_t *) l->dev, l->n_words, (unsigned int *) ctx->scratch));
  int rc = read_scratch (ctx, 1);
  if (rc) return rc;
  *sorted = (ctx->scratch_host[0] & 0xffffffffu) == 0;
  return GT4HIP_OK;
}

extern "C" int gt4hip_list_lower_bound (gt4hip_context *ctx, const gt4hip_list *l, uint64_t key, uint64_t *index)
{
  if (!ctx || !l || !index) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  if (!l->n_words) {
    *index = 0;
    return GT4HIP_OK;
  }
  HIPCHK (ctx, launch_lower_bound (ctx->stream, (const uint32_t *) l->dev, l->n_words, key, ctx->scratch));
  int rc = read_scratch (ctx, 1);
  if (rc) return rc;
  *index = ctx->scratch_host[0];
  return GT4HIP_OK;
}

extern "C" int gt4hip_list_get_word (gt4hip_context *ctx, const gt4hip_list *l, uint64_t idx, uint64_t *word, uint32_t *count)
{
  if (!ctx || !l || idx >= l->n_words) return GT4HIP_EINVAL;
  unsigned char rec[12];
  int rc = gt4hip_list_download_range (ctx, l, idx, 1, rec);
  if (rc) return rc;
  if (word) memcpy (word, rec, 8);
  if (count) memcpy (count, rec + 8, 4);
  return GT4HIP_OK;
}

extern "C" int gt4hip_generate_ex (gt4hip_context *ctx, gt4hip_list *l, uint64_t n, uint64_t key_seed, uint64_t count_seed,
                                   uint32_t max_count, uint64_t mult, uint64_t add)
{
  if (!ctx || !l || n > l->capacity || !max_count || !l->word_length || l->word_length > 32 || !mult || add >= mult) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  l->n_words = n;
  if (!n) return GT4HIP_OK;
  /* keyspace 4^k (2^64 for k = 32), thinned by `mult`; stride = floor(keyspace / mult / n) */
  unsigned __int128 space = l->word_length == 32 ? ((unsigned __int128) 1 << 64) : ((unsigned __int128) 1 << (2 * l->word_length));
  space /= mult;
  unsigned __int128 st = space / n;
  if (st > 0xffffffffffffffffull) st = 0xffffffffffffffffull;
  const uint64_t stride = (uint64_t) st;
  if (!stride) return gt4hip_fail (ctx, GT4HIP_EINVAL, "gt4hip_generate: %llu keys do not fit k=%u", (unsigned long long) n, l->word_length);
  HIPCHK (ctx, launch_generate (ctx->stream, (uint32_t *) l->dev, n, stride, key_seed, count_seed, max_count, mult, add));
  HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
  return GT4HIP_OK;
}

extern "C" int gt4hip_generate (gt4hip_context *ctx, gt4hip_list *l, uint64_t n, uint64_t seed, uint32_t max_count)
{
  return gt4hip_generate_ex (ctx, l, n, seed, seed + 1, max_count, 1, 0);
}

/* every stride-th key of a list (the last key of every full block of `stride` records) -> out[0 .. n / stride) */
__global__ void k_sample_stride (const uint32_t *__restrict__ rec, uint64_t n, uint64_t stride, unsigned long long *__restrict__ out)
{
  const uint64_t m = n / stride;
  for (uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t) gridDim.x * blockDim.x) {
    const uint32_t *q = rec + 3 * ((j + 1) * stride - 1);
    out[j] = (unsigned long long) q[0] | ((unsigned long long) q[1] << 32);
  }
}

/* SAMPLED splitters: equal-width key ranges balance the shards only for uniformly spread keys;
 * 2-bit packed k-mers of a real genome are not (src/sequence.c:116-130: the word IS the sequence, low-complexity and
 * GC-poor prefixes are crowded).  Every S-th key of every list (S = all records / 65536), merged on the host; the
 * first key of shard g is the merged sample at g / n_shards: the shards' INPUT records then differ by at most
 * n_lists * S.  Every rank that holds the same lists computes the same cuts.  first_keys[0] = 0. */
extern "C" int gt4hip_shard_cuts (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n, uint32_t n_shards, uint64_t *first_keys)
{
  if (!ctx || !lists || !n || !n_shards || !first_keys) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  uint64_t total = 0;
  uint32_t wl = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (!lists[i]) return GT4HIP_EINVAL;
    total += lists[i]->n_words;
    wl = lists[i]->word_length > wl ? lists[i]->word_length : wl;
  }
  const uint64_t target = 65536;
  const uint64_t stride = total / target ? total / target : 1;
  uint64_t m_total = 0;
  for (uint32_t i = 0; i < n; i++) m_total += lists[i]->n_words / stride;
  first_keys[0] = 0;
  if (m_total < (uint64_t) n_shards) { /* (hardly any records: equal-width ranges) */
    for (uint32_t g = 1; g < n_shards; g++) first_keys[g] = gt4hip_shard_first_key (wl, n_shards, g);
    return GT4HIP_OK;
  }
  void *dev = NULL, *owner = NULL;
  int rc = gt4hip_block_alloc (ctx, (size_t) m_total * 8, &dev, &owner);
  if (rc) return rc;
  std::vector<unsigned long long> host ((size_t) m_total);
  uint64_t at = 0;
  for (uint32_t i = 0; i < n; i++) {
    const uint64_t m = lists[i]->n_words / stride;
    if (!m) continue;
    const unsigned grid = (unsigned) ((m + 255) / 256 < 4096 ? (m + 255) / 256 : 4096);
    hipLaunchKernelGGL (k_sample_stride, dim3 (grid), dim3 (256), 0, ctx->stream, (const uint32_t *) lists[i]->dev, lists[i]->n_words, stride, (unsigned long long *) dev + at);
    at += m;
  }
  hipError_t e = hipMemcpyAsync (host.data (), dev, (size_t) m_total * 8, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize (ctx->stream);
  gt4hip_block_free (owner);
  if (e != hipSuccess) return gt4hip_fail (ctx, GT4HIP_EHIP, "gt4hip_shard_cuts: %s", hipGetErrorString (e));
  std::sort (host.begin (), host.end ());
  for (uint32_t g = 1; g < n_shards; g++) {
    /* (the sample itself goes to the shard on its left: the cut is one above it) */
    const unsigned long long sk = host[(size_t) (((unsigned __int128) m_total * g) / n_shards) - 1];
    const uint64_t cut = sk == ~0ull ? sk : sk + 1;
    first_keys[g] = cut > first_keys[g - 1] ? cut : first_keys[g - 1];
  }
  return GT4HIP_OK;
}

extern "C" uint64_t gt4hip_shard_first_key (uint32_t word_length, uint32_t n_shards, uint32_t g)
{
  if (!n_shards || g >= n_shards) return 0;
  const unsigned __int128 space = word_length >= 32 ? ((unsigned __int128) 1 << 64) : ((unsigned __int128) 1 << (2 * word_length));
  return (uint64_t) (space * g / n_shards);
}

/* ------------------------------------------------------------------ workspace */

static int grow (gt4hip_context *ctx, void **p, size_t *have, size_t need)
{
  if (*have >= need) return GT4HIP_OK;
  if (*p) {
    HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
    HIPCHK (ctx, hipFree (*p));
    *p = NULL;
    *have = 0;
  }
  need += need / 8; /* slack so that slightly larger follow-up calls do not reallocate */
  hipError_t e = gt4hip_dev_alloc (ctx, p, need);
  if (e != hipSuccess) return gt4hip_fail (ctx, GT4HIP_ENOMEM, "workspace hipMalloc of %zu bytes failed: %s", need, hipGetErrorString (e));
  *have = need;
  return GT4HIP_OK;
}

/* ------------------------------------------------------------------ pair operation core */

/* descriptor workspace: single pass agg u32[4][rows*64] + carry u64[4][rows+1] + rowsum u64[4][rows];
 * the two-pass path keeps u64[4] per tile in the same buffer */
static size_t desc_bytes_for (uint64_t tiles)
{
  const uint64_t rows = (tiles + 63) / 64;
  const size_t single = (size_t) rows * 64 * 16 + (size_t) (rows + 1) * 32 + (size_t) rows * 32;
  const size_t two = (size_t) tiles * 32;
  return ((single > two ? single : two) + 255) & ~(size_t) 255;
}


struct PairRun {
  uint64_t n_words[4];
  uint64_t total_count[4];
  double merge_ms, device_ms;
  uint64_t tiles;
};

/* Runs the merge of (a, b) with fully resolved kernel parameters.  dst[s] (device record buffers)
 * must be non-null for every requested stream unless count_only. */
static int run_pair (gt4hip_context *ctx, const uint32_t *A, uint64_t nA, const uint32_t *B, uint64_t nB,
                     const PairParams &p_in, bool count_only, uint32_t *const dst[4], PairRun *run, bool force_two_pass = false)
{
  if (p_in.ops == 8u) {
    /* the second complement alone is the first complement of the swapped pair
     * (include_in_complement (f2, f1, 0), glistcompare.c:862, :896): same specialised kernel */
    PairParams q = p_in;
    q.ops = 4u;
    q.rule[2] = p_in.rule[3];
    q.subtract = 0;
    uint32_t *const d2[4] = { NULL, NULL, dst ? dst[3] : NULL, NULL };
    const int rc = run_pair (ctx, B, nB, A, nA, q, count_only, d2, run, force_two_pass);
    run->n_words[3] = run->n_words[2];
    run->total_count[3] = run->total_count[2];
    run->n_words[2] = run->total_count[2] = 0;
    return rc;
  }
  if (p_in.ops == 2u && nA > nB && p_in.rule[1] != 2u && p_in.rule[1] != RULE_MINZ) {
    /* an intersection searches with the records of its first list: let that be the shorter one.
     * Keep test and every rule but SUBTRACT / the N-way running MIN are symmetric in (f1, f2);
     * FIRST and SECOND trade places. */
    PairParams q = p_in;
    if (q.rule[1] == 5u) q.rule[1] = 6u;
    else if (q.rule[1] == 6u) q.rule[1] = 5u;
    return run_pair (ctx, B, nB, A, nA, q, count_only, dst, run, force_two_pass);
  }
  PairParams p = p_in;
  p.spin_limit = ctx->spin_limit;
  memset (run, 0, sizeof *run);
  const uint64_t total = nA + nB;
  if (!total || !p.ops) return GT4HIP_OK;
  /* count-only calls: 512-thread workgroups; everything that materialises records: 1024 */
  const int geom = ctx->force_geom ? (ctx->force_geom > 0 ? 1 : 0) : (count_only ? 0 : 1);
  const uint64_t tile_records = merge_tile_records (geom, p.ops);
  const uint64_t tiles = (total + tile_records - 1) / tile_records;
  if (tiles >= 0xffffffffull) return gt4hip_fail (ctx, GT4HIP_EINVAL, "lists too long: %llu merge tiles", (unsigned long long) tiles);
  run->tiles = tiles;
  /* the scanner as a group of wavefronts pays off where one wavefront cannot keep up (more than ~2e4
   * rows of 64 tiles per launch: the small geometry on billions of records); below that the single
   * wavefront's shorter path to the carry is worth more (option "scan_group": -1 never, 1 always) */
  /* tiles by ticket (dynamic dealing) for the record-writing single-pass kernels of the large
   * geometry: their ~40 tiles per microsecond are well below the ~88 returning atomics per microsecond
   * one counter sustains (the count-only geometry's 200+ are not: 8.7 -> 23.8 ms), and arrival order
   * spares the fast workers the wait for the slow ones in the chained scan (measured at 2 x 2e9:
   * intersection 12.75 -> 12.55 ms, union 21.7 -> 21.45, union + intersection 27.05 -> 26.25; the first
   * complement alone is 2 % SLOWER and stays round-robin).  Option "dynamic": 1 always, -1 never. */
  p.dynamic = ctx->dynamic > 0 ? 1u : (ctx->dynamic < 0 ? 0u : ((geom == 1 && !count_only && p.ops != 4u && p.ops != 8u) ? 1u : 0u));
  p.scan_group = ctx->scan_group > 0 ? 1u : (ctx->scan_group < 0 ? 0u : (tiles > (20000ull << 6) ? 1u : 0u));
  int rc;
  if ((rc = grow (ctx, (void **) &ctx->part, &ctx->part_bytes, (size_t) (tiles + 1) * 16 + (size_t) (tiles / 64 + 3) * 8))) return rc; /* tile ranges + coarse co-ranks */
  const bool two_pass = (ctx->two_pass || force_two_pass) && !count_only;
  const bool need_desc = !count_only;
  if (need_desc && (rc = grow (ctx, (void **) &ctx->desc, &ctx->desc_bytes, desc_bytes_for (tiles)))) return rc;
  if (two_pass) {
    const size_t nb = (size_t) ((tiles + 2047) / 2048) * 32;
    if ((rc = grow (ctx, (void **) &ctx->block_sums, &ctx->block_sums_bytes, nb))) return rc;
  }
  const int first_mode = count_only ? MODE_COUNT : (two_pass ? MODE_COUNT : MODE_LOOKBACK);
  int grid = ctx->n_cus * merge_blocks_per_cu (geom, first_mode, p.ops, &p);
  if (ctx->grid_override > 0) grid = (int) ctx->grid_override;
  if ((uint64_t) grid > tiles + 1) grid = (int) tiles + 1; /* workers + the scanner workgroup */
  int grid2 = ctx->n_cus * merge_blocks_per_cu (geom, MODE_OFFSETS, p.ops, &p);
  if ((uint64_t) grid2 > tiles) grid2 = (int) tiles;

  PairOutputs outs;
  for (int s = 0; s < 4; s++) outs.rec[s] = (count_only || !dst) ? NULL : dst[s];

  hipStream_t st = ctx->stream;
  HIPCHK (ctx, hipEventRecord (ctx->ev[0], st));
  HIPCHK (ctx, hipMemsetAsync (ctx->ctl, 0, sizeof (PairControl), st));
  if (need_desc && !two_pass) HIPCHK (ctx, hipMemsetAsync (ctx->desc, 0, desc_bytes_for (tiles), st));
  HIPCHK (ctx, launch_partition (st, A, nA, B, nB, tiles, tile_records, ctx->part));
  HIPCHK (ctx, hipEventRecord (ctx->ev[1], st));
  if (count_only) {
    HIPCHK (ctx, launch_pair_merge (st, geom, MODE_COUNT, grid, A, nA, B, nB, ctx->part, tiles, p, outs, NULL, ctx->ctl));
  } else if (two_pass) {
    HIPCHK (ctx, launch_pair_merge (st, geom, MODE_COUNT, grid, A, nA, B, nB, ctx->part, tiles, p, outs, ctx->desc, ctx->ctl));
    HIPCHK (ctx, launch_scan_tiles (st, ctx->desc, tiles, ctx->block_sums));
    HIPCHK (ctx, hipMemsetAsync (ctx->ctl, 0, sizeof (PairControl), st));
    HIPCHK (ctx, launch_pair_merge (st, geom, MODE_OFFSETS, grid2, A, nA, B, nB, ctx->part, tiles, p, outs, ctx->desc, ctx->ctl));
  } else {
    HIPCHK (ctx, launch_pair_merge (st, geom, MODE_LOOKBACK, grid, A, nA, B, nB, ctx->part, tiles, p, outs, ctx->desc, ctx->ctl));
  }
  HIPCHK (ctx, hipEventRecord (ctx->ev[2], st));
  HIPCHK (ctx, hipMemcpyAsync (ctx->ctl_host, ctx->ctl, sizeof (PairControl), hipMemcpyDeviceToHost, st));
  HIPCHK (ctx, hipEventRecord (ctx->ev[3], st));
  HIPCHK (ctx, hipStreamSynchronize (st));
  float ms = 0;
  if (hipEventElapsedTime (&ms, ctx->ev[1], ctx->ev[2]) == hipSuccess) run->merge_ms = ms;
  if (hipEventElapsedTime (&ms, ctx->ev[0], ctx->ev[3]) == hipSuccess) run->device_ms = ms;
PROF (
  {
    static const char *names[8] = { "p0 wait+lds", "B0", "ring+fetch issue", "p1 rank", "B1", "p2 scan/publish", "B2+out+B3+scatter", "housekeeping" };
    unsigned long long tot = 0;
    for (int i = 0; i < 8; i++) tot += ctx->ctl_host->phase_cycles[i];
    fprintf (stderr, "[phases] tiles %llu merge %.3f ms:", (unsigned long long) tiles, run->merge_ms);
    for (int i = 0; i < 8; i++) fprintf (stderr, " %s %.1f%%", names[i], tot ? 100.0 * ctx->ctl_host->phase_cycles[i] / tot : 0.0);
    fprintf (stderr, " | avg cycles/tile %.0f\n", tiles ? (double) tot / tiles : 0.0);
    const unsigned long long *rs = ctx->ctl_host->resolve_stats;
    if (rs[0]) fprintf (stderr, "[resolve] sampled %llu avg spins %.2f first-look agg-not-ready %.1f%% carry-not-ready %.1f%% | [scanner] rows %llu polling rounds %llu rows complete at batch load %llu\n", rs[0], (double) rs[1] / rs[0], 100.0 * rs[3] / rs[0], 100.0 * rs[4] / rs[0], rs[7], rs[5], rs[6]);
  }
)
  if (ctx->ctl_host->error) {
    const unsigned flags = ctx->ctl_host->error;
    if (!two_pass && !count_only && !(flags & 2u)) {
      /* a bounded wait of the single-pass path gave up (a worker was not resident, or the device
       * is shared): the count + scan + write path has no inter-workgroup dependency -- rerun there */
      ctx->single_pass_fallbacks++;
      return run_pair (ctx, A, nA, B, nB, p, count_only, dst, run, true);
    }
    return gt4hip_fail (ctx, GT4HIP_EINTERNAL, "merge kernel reported error flags 0x%x", flags);
  }
  for (int s = 0; s < 4; s++) {
    run->n_words[s] = ctx->ctl_host->n_words[s];
    run->total_count[s] = ctx->ctl_host->total_count[s];
  }
  return GT4HIP_OK;
}

static uint64_t worst_case (int s, uint64_t nA, uint64_t nB)
{
  switch (s) {
    case 0: return nA + nB;
    case 1: return nA < nB ? nA : nB;
    case 2: return nA;
    default: return nB;
  }
}

/* Allocates missing outputs, runs, trims.  `given[s]` optional caller lists. */
static int pair_with_outputs (gt4hip_context *ctx, const gt4hip_list *a, const gt4hip_list *b, const PairParams &p,
                              bool count_only, gt4hip_list *out[4], PairRun *run)
{
  gt4hip_list *made[4] = { NULL, NULL, NULL, NULL };
  uint32_t *dst[4] = { NULL, NULL, NULL, NULL };
  int rc = GT4HIP_OK;
  if (!count_only) {
    for (int s = 0; s < 4 && !rc; s++) {
      if (!((p.ops >> s) & 1u)) continue;
      const uint64_t need = worst_case (s, a->n_words, b->n_words);
      if (out[s]) {
        if (out[s]->capacity < need) rc = gt4hip_fail (ctx, GT4HIP_EINVAL, "output %d: capacity %llu < worst case %llu", s,
                                                 (unsigned long long) out[s]->capacity, (unsigned long long) need);
      } else {
        rc = gt4hip_list_new (ctx, need, a->word_length, &made[s]);
        if (!rc) out[s] = made[s];
      }
      if (!rc) dst[s] = (uint32_t *) out[s]->dev;
    }
  }
  if (!rc) rc = run_pair (ctx, (const uint32_t *) a->dev, a->n_words, (const uint32_t *) b->dev, b->n_words, p, count_only, dst, run);
  if (rc) {
    for (int s = 0; s < 4; s++)
      if (made[s]) {
        gt4hip_list_free (made[s]);
        out[s] = NULL;
      }
    return rc;
  }
  if (!count_only)
    for (int s = 0; s < 4; s++)
      if ((p.ops >> s) & 1u) {
        out[s]->n_words = run->n_words[s];
        out[s]->word_length = a->word_length;
      }
  return GT4HIP_OK;
}

extern "C" int gt4hip_compare (gt4hip_context *ctx, const gt4hip_list *a, const gt4hip_list *b,
                                const gt4hip_compare_params *prm, gt4hip_compare_result *res)
{
  if (!ctx || !a || !b || !prm || !res) return GT4HIP_EINVAL;
  if (prm->ops & ~15u) return gt4hip_fail (ctx, GT4HIP_EINVAL, "gt4hip_compare: unknown op bits 0x%x", prm->ops);
  if (prm->rule < 0 || prm->rule > 7) return gt4hip_fail (ctx, GT4HIP_EINVAL, "gt4hip_compare: unknown rule %d", prm->rule);
  if (a->word_length != b->word_length) return gt4hip_fail (ctx, GT4HIP_EWORDLEN, "word lengths differ (%u != %u)", b->word_length, a->word_length);
  HIPCHK (ctx, hipSetDevice (ctx->device));
  PairParams p;
  memset (&p, 0, sizeof p);
  p.ops = prm->ops;
  /* DEFAULT resolves per output: ADD for union (:463), MIN for intersection (:472), SUBTRACT for
   * both complements (:486) */
  const uint32_t r = (uint32_t) prm->rule;
  p.rule[0] = r ? r : GT4HIP_RULE_ADD;
  p.rule[1] = r ? r : GT4HIP_RULE_MIN;
  p.rule[2] = r ? r : GT4HIP_RULE_SUBTRACT;
  p.rule[3] = r ? r : GT4HIP_RULE_SUBTRACT;
  p.cutoff = prm->cutoff;
  p.subtract = prm->subtract ? 1u : 0u;
  p.count_override = prm->count_override;
  p.filter = FILTER_REFERENCE;
  PairRun run;
  gt4hip_list *out[4];
  for (int s = 0; s < 4; s++) out[s] = ((prm->ops >> s) & 1u) && !prm->count_only ? res->out[s] : NULL;
  int rc = pair_with_outputs (ctx, a, b, p, prm->count_only != 0, out, &run);
  if (rc) return rc;
  for (int s = 0; s < 4; s++) {
    res->n_words[s] = run.n_words[s];
    res->total_count[s] = run.total_count[s];
    res->out[s] = out[s];
  }
  res->merge_kernel_ms = run.merge_ms;
  res->device_ms = run.device_ms;
  res->merge_tiles = run.tiles;
  return GT4HIP_OK;
}

/* ------------------------------------------------------------------ N-way operations */

static PairParams nway_params (uint32_t op_bit, uint32_t rule, uint32_t cutoff, uint32_t ovr, uint32_t filter)
{
  PairParams p;
  memset (&p, 0, sizeof p);
  p.ops = op_bit;
  for (int s = 0; s < 4; s++) p.rule[s] = rule;
  p.cutoff = cutoff;
  p.count_override = ovr;
  p.filter = filter;
  return p;
}

/* Final step shared by both N-way ops: merge (a, b) into the caller-visible result. */
static int nway_final (gt4hip_context *ctx, const gt4hip_list *a, const gt4hip_list *b, const PairParams &p, int stream_idx,
                       bool count_only, gt4hip_multi_result *res)
{
  gt4hip_list *out[4] = { NULL, NULL, NULL, NULL };
  out[stream_idx] = count_only ? NULL : res->out;
  PairRun run;
  int rc = pair_with_outputs (ctx, a, b, p, count_only, out, &run);
  if (rc) return rc;
  res->n_words = run.n_words[stream_idx];
  res->total_count = run.total_count[stream_idx];
  res->out = count_only ? NULL : out[stream_idx];
  res->device_ms += run.device_ms;
  res->records_read += a->n_words + b->n_words;
  if (!count_only) res->records_written += run.n_words[stream_idx];
  return GT4HIP_OK;
}

static int empty_result (gt4hip_context *ctx, uint32_t word_length, bool count_only, gt4hip_multi_result *res)
{
  res->n_words = 0;
  res->total_count = 0;
  if (count_only) {
    res->out = NULL;
    return GT4HIP_OK;
  }
  if (res->out) {
    res->out->n_words = 0;
    return GT4HIP_OK;
  }
  return gt4hip_list_new (ctx, 0, word_length, &res->out);
}

/* N-way union by the one-pass tile kernel (gt4hip_nway.hip): groups of up to eight lists per launch;
 * more than eight lists take levels of eight-way merges that keep every key (ADD / MAX are
 * associative, NUMBER ignores the counts), the cutoff is applied once, at the last level (:574).
 * *done = 0: nothing was produced, the caller takes the pairwise tree. */
/* In how many of the lists does a key of the lists lie?  256 keys of each of four probe lists, looked up in every list
 * (binary searches): matches[0] += lists holding the key.  One pass over 9 .. 32 lists ranks a key among everything in its
 * bucket, and a key that sixteen lists share puts sixteen records there: measured (32 x 1.25e8 records, profiles/round5):
 * one pass 47.9 ms against 67.7 for levels of eight-way merges where few keys are shared, 64.6 against 54.1 where sixteen
 * of the lists are the same. */
struct ShareProbe {
  const uint32_t *list[32];
  uint64_t n[32];
  uint32_t k;
  uint32_t probe[4];
};

__global__ __launch_bounds__ (256) void k_share_probe (ShareProbe sp, unsigned long long *matches)
{
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t b = id % sp.k, j = (id / sp.k) % 256u, a = sp.probe[(id / sp.k) / 256u];
  if (id >= 4u * 256u * sp.k || !sp.n[a] || !sp.n[b]) return;
  const uint32_t *pa = sp.list[a] + 3 * (uint64_t) (((unsigned __int128) sp.n[a] * j) / 256u);
  const uint64_t key = (uint64_t) pa[0] | ((uint64_t) pa[1] << 32);
  uint64_t lo = 0, hi = sp.n[b];
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    const uint32_t *q = sp.list[b] + 3 * mid;
    if (((uint64_t) q[0] | ((uint64_t) q[1] << 32)) < key) lo = mid + 1;
    else hi = mid;
  }
  bool hit = false;
  if (lo < sp.n[b]) {
    const uint32_t *q = sp.list[b] + 3 * lo;
    hit = ((uint64_t) q[0] | ((uint64_t) q[1] << 32)) == key;
  }
  const unsigned long long m = __builtin_amdgcn_ballot_w64 (hit);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd (matches, (unsigned long long) __popcll (m));
}

/* the mean number of lists a probed key lies in (1: the lists share nothing) */
static int shared_key_multiplicity (gt4hip_context *ctx, const std::vector<const gt4hip_list *> &lists, double *mean)
{
  ShareProbe sp;
  memset (&sp, 0, sizeof sp);
  sp.k = (uint32_t) (lists.size () < 32 ? lists.size () : 32);
  for (uint32_t i = 0; i < sp.k; i++) {
    sp.list[i] = (const uint32_t *) lists[i]->dev;
    sp.n[i] = lists[i]->n_words;
  }
  sp.probe[0] = 0;
  sp.probe[1] = (sp.k / 4 + 1) % sp.k;
  sp.probe[2] = sp.k / 2;
  sp.probe[3] = (3 * sp.k / 4 + 1) % sp.k;
  HIPCHK (ctx, hipMemsetAsync (ctx->scratch, 0, 8, ctx->stream));
  hipLaunchKernelGGL (k_share_probe, dim3 ((4 * 256 * sp.k + 255) / 256), dim3 (256), 0, ctx->stream, sp, ctx->scratch);
  const int rc = read_scratch (ctx, 1);
  if (rc) return rc;
  uint32_t probed = 0;
  for (int q = 0; q < 4; q++) probed += sp.n[sp.probe[q]] ? 256u : 0u;
  *mean = probed ? (double) ctx->scratch_host[0] / probed : 1.0;
  return GT4HIP_OK;
}

static int union_multi_kway (gt4hip_context *ctx, const std::vector<const gt4hip_list *> &work, uint32_t rule, uint32_t cutoff, uint32_t ovr,
                             bool count_only, gt4hip_multi_result *res, int *done)
{
  *done = 0;
  std::vector<const gt4hip_list *> cur = work;
  std::vector<gt4hip_list *> owned;
  const uint32_t wl = work[0]->word_length;
  int rc = GT4HIP_OK;
  uint64_t rd = 0, wr = 0;
  double ms_total = 0;
  auto drop = [&] () {
    for (gt4hip_list *l : owned) gt4hip_list_free (l);
    owned.clear ();
  };
  /* lists per launch of the tile kernel: up to 32 in ONE pass (round 5; glistmaker's collation width, reference
   * src/glistmaker.c:787-835), option "kway_max" = 8 restores the levels of eight-way merges */
  size_t W = ctx->kway_max == 8 ? 8 : 32;
  if (W == 32 && cur.size () > 8 && ctx->kway_max != 33) { /* ("kway_max" = 33: one pass whatever the keys; tests) */
    double m = 1.0;
    if ((rc = shared_key_multiplicity (ctx, cur, &m))) return rc;
    ctx->kway_shared_x100 = (uint64_t) (100.0 * m);
    if (m > 5.0) W = 8; /* keys that many lists share: levels of eight-way merges fold them step by step */
  }
  ctx->kway_width = cur.size () <= 8 ? 8u : (uint64_t) W; /* (up to eight lists take the eight-list instance of the kernel) */
  while (cur.size () > W && !rc) {
    std::vector<const gt4hip_list *> next;
    std::vector<gt4hip_list *> next_owned;
    for (size_t i = 0; i < cur.size () && !rc; i += W) {
      const size_t g = cur.size () - i < W ? cur.size () - i : W;
      if (g < 3) { /* one or two left over: carried to the next level as they are */
        for (size_t j = 0; j < g; j++) next.push_back (cur[i + j]);
        continue;
      }
      uint64_t cap = 0;
      for (size_t j = 0; j < g; j++) cap += cur[i + j]->n_words;
      gt4hip_list *o = NULL;
      rc = gt4hip_list_new (ctx, cap, wl, &o);
      if (rc) break;
      uint64_t n = 0, t = 0;
      double ms = 0;
      int used = 0;
      rc = gt4hip_nway_union (ctx, &cur[i], (uint32_t) g, rule, cutoff, ovr, FILTER_RAW, false, o, &n, &t, &ms, &used);
      if (rc || !used) {
        gt4hip_list_free (o);
        for (gt4hip_list *l : next_owned) gt4hip_list_free (l);
        drop ();
        return rc;
      }
      o->n_words = n;
      rd += cap;
      wr += n;
      ms_total += ms;
      next.push_back (o);
      next_owned.push_back (o);
    }
    /* the previous level's temporaries are consumed, except those carried over */
    for (gt4hip_list *l : owned) {
      bool carried = false;
      for (const gt4hip_list *n : next) carried |= (n == l);
      if (carried) next_owned.push_back (l);
      else gt4hip_list_free (l);
    }
    owned.swap (next_owned);
    cur.swap (next);
  }
  if (rc) {
    drop ();
    return rc;
  }
  if (cur.size () < (ctx->kway_enabled == 2 ? 2u : 3u)) {
    /* (possible only behind a level of eight-way merges) the last one or two go through the pair kernel */
    gt4hip_list empty_b;
    memset (&empty_b, 0, sizeof empty_b);
    empty_b.ctx = ctx;
    empty_b.word_length = wl;
    const PairParams fin = nway_params (GT4HIP_OP_UNION, rule, cutoff, ovr, FILTER_RESULT);
    res->device_ms += ms_total;
    res->records_read += rd;
    res->records_written += wr;
    rc = nway_final (ctx, cur[0], cur.size () > 1 ? cur[1] : &empty_b, fin, 0, count_only, res);
    drop ();
    *done = rc == GT4HIP_OK;
    if (*done) ctx->kway_calls++;
    return rc;
  }
  uint64_t cap = 0;
  for (const gt4hip_list *l : cur) cap += l->n_words;
  gt4hip_list *o = NULL, *made = NULL;
  if (!count_only) {
    if (res->out) {
      if (res->out->capacity < cap) {
        drop ();
        return gt4hip_fail (ctx, GT4HIP_EINVAL, "output: capacity %llu < worst case %llu", (unsigned long long) res->out->capacity, (unsigned long long) cap);
      }
      o = res->out;
    } else {
      rc = gt4hip_list_new (ctx, cap, wl, &made);
      if (rc) {
        drop ();
        return rc;
      }
      o = made;
    }
  }
  uint64_t n = 0, t = 0;
  double ms = 0;
  int used = 0;
  rc = gt4hip_nway_union (ctx, cur.data (), (uint32_t) cur.size (), rule, cutoff, ovr, FILTER_RESULT, count_only, o, &n, &t, &ms, &used);
  drop ();
  if (rc || !used) {
    if (made) gt4hip_list_free (made);
    return rc;
  }
  if (o) {
    o->n_words = n;
    o->word_length = wl;
  }
  res->n_words = n;
  res->total_count = t;
  res->out = count_only ? NULL : o;
  res->device_ms += ms_total + ms;
  res->records_read += rd + cap;
  res->records_written += wr + (count_only ? 0 : n);
  ctx->kway_calls++;
  *done = 1;
  return GT4HIP_OK;
}

extern "C" int gt4hip_union_multi (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n_lists, uint32_t cutoff,
                                    int32_t rule, uint32_t ovr, int32_t count_only, gt4hip_multi_result *res)
{
  if (!ctx || !lists || !n_lists || !res) return GT4HIP_EINVAL;
  /* src/glistcompare.c:518-523 */
  if (rule == GT4HIP_RULE_DEFAULT) rule = GT4HIP_RULE_ADD;
  else if (rule != GT4HIP_RULE_ADD && rule != GT4HIP_RULE_MAX && rule != GT4HIP_RULE_NUMBER)
    return gt4hip_fail (ctx, GT4HIP_ERULE, "union_multi: Invalid rule %u (only ADD, MAX and NUMBER allowed)", (unsigned) rule);
  for (uint32_t j = 0; j < n_lists; j++) {
    if (!lists[j]) return GT4HIP_EINVAL;
    if (lists[j]->word_length != lists[0]->word_length) return gt4hip_fail (ctx, GT4HIP_EWORDLEN, "word lengths differ");
  }
  HIPCHK (ctx, hipSetDevice (ctx->device));
  res->device_ms = 0;
  res->records_read = res->records_written = 0;
  std::vector<const gt4hip_list *> work;
  std::vector<gt4hip_list *> owned; /* intermediate levels, freed as soon as consumed */
  for (uint32_t j = 0; j < n_lists; j++)
    if (lists[j]->n_words) work.push_back (lists[j]); /* :525-532 empty lists are dropped */
  const uint32_t wl = lists[0]->word_length;
  ctx->last_multi_one_pass = 0;
  if (work.empty ()) return empty_result (ctx, wl, count_only != 0, res);
  if (ctx->kway_enabled && work.size () >= (ctx->kway_enabled == 2 ? 2u : 3u)) {
    int done = 0;
    const int krc = union_multi_kway (ctx, work, (uint32_t) rule, cutoff, ovr, count_only != 0, res, &done);
    if (!krc && done) ctx->last_multi_one_pass = 1;
    if (krc || done) return krc;
    res->device_ms = 0;
    res->records_read = res->records_written = 0;
  }
  gt4hip_list empty_b;
  memset (&empty_b, 0, sizeof empty_b);
  empty_b.ctx = ctx;
  empty_b.word_length = wl;
  int rc = GT4HIP_OK;
  /* pairwise tree in HBM: intermediate levels keep every key (count rules ADD/MAX are associative
   * and commutative), the cutoff is applied once, on the final count (:574) */
  const PairParams raw = nway_params (GT4HIP_OP_UNION, (uint32_t) rule, cutoff, ovr, FILTER_RAW);
  while (work.size () > 2 && !rc) {
    std::vector<const gt4hip_list *> next;
    std::vector<gt4hip_list *> next_owned;
    for (size_t i = 0; i + 1 < work.size () && !rc; i += 2) {
      gt4hip_list *out[4] = { NULL, NULL, NULL, NULL };
      PairRun run;
      rc = pair_with_outputs (ctx, work[i], work[i + 1], raw, false, out, &run);
      if (!rc) {
        res->device_ms += run.device_ms;
        res->records_read += work[i]->n_words + work[i + 1]->n_words;
        res->records_written += run.n_words[0];
        next.push_back (out[0]);
        next_owned.push_back (out[0]);
      }
    }
    if (work.size () & 1) next.push_back (work.back ());
    /* the previous level's temporaries are consumed, except an odd one carried over */
    for (gt4hip_list *l : owned) {
      bool carried = false;
      for (const gt4hip_list *n : next) carried |= (n == l);
      if (carried) next_owned.push_back (l);
      else gt4hip_list_free (l);
    }
    owned.swap (next_owned);
    work.swap (next);
  }
  if (!rc) {
    const PairParams fin = nway_params (GT4HIP_OP_UNION, (uint32_t) rule, cutoff, ovr, FILTER_RESULT);
    rc = nway_final (ctx, work[0], work.size () > 1 ? work[1] : &empty_b, fin, 0, count_only != 0, res);
  }
  for (gt4hip_list *l : owned) gt4hip_list_free (l);
  return rc;
}

extern "C" int gt4hip_intersect_multi (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n_lists, uint32_t cutoff,
                                        int32_t rule, uint32_t ovr, int32_t count_only, gt4hip_multi_result *res)
{
  if (!ctx || !lists || !n_lists || !res) return GT4HIP_EINVAL;
  /* src/glistcompare.c:622-627 */
  if (rule == GT4HIP_RULE_DEFAULT) rule = GT4HIP_RULE_MIN;
  else if (rule != GT4HIP_RULE_ADD && rule != GT4HIP_RULE_MIN && rule != GT4HIP_RULE_MAX && rule != GT4HIP_RULE_NUMBER)
    return gt4hip_fail (ctx, GT4HIP_ERULE, "intersect_multi: Invalid rule %u (only ADD, MIN, MAX and NUMBER allowed)", (unsigned) rule);
  bool any_empty = false;
  for (uint32_t j = 0; j < n_lists; j++) {
    if (!lists[j]) return GT4HIP_EINVAL;
    if (lists[j]->word_length != lists[0]->word_length) return gt4hip_fail (ctx, GT4HIP_EWORDLEN, "word lengths differ");
    any_empty |= lists[j]->n_words == 0;
  }
  HIPCHK (ctx, hipSetDevice (ctx->device));
  res->device_ms = 0;
  res->records_read = res->records_written = 0;
  const uint32_t wl = lists[0]->word_length;
  if (any_empty) return empty_result (ctx, wl, count_only != 0, res); /* :633-636 */
  /* Left-to-right chain R_k = R_{k-1} n L_k, exactly the reference's fold order over the lists
   * (:655-678): the running MIN restarts at 0 (RULE_MINZ), which is not associative, so no tree. */
  const uint32_t krule = rule == GT4HIP_RULE_MIN ? RULE_MINZ : (uint32_t) rule;
  if (n_lists == 1) {
    gt4hip_list empty_b;
    memset (&empty_b, 0, sizeof empty_b);
    empty_b.ctx = ctx;
    empty_b.word_length = wl;
    /* fold(0, c) of one list: c for MIN/MAX/ADD, the override for NUMBER */
    const PairParams fin = nway_params (GT4HIP_OP_UNION, rule == GT4HIP_RULE_NUMBER ? GT4HIP_RULE_NUMBER : GT4HIP_RULE_FIRST, cutoff, ovr, FILTER_RESULT);
    return nway_final (ctx, lists[0], &empty_b, fin, 0, count_only != 0, res);
  }
  const gt4hip_list *acc = lists[0];
  gt4hip_list *acc_owned = NULL;
  int rc = GT4HIP_OK;
  for (uint32_t k = 1; k + 1 < n_lists && !rc; k++) {
    gt4hip_list *out[4] = { NULL, NULL, NULL, NULL };
    PairRun run;
    rc = pair_with_outputs (ctx, acc, lists[k], nway_params (GT4HIP_OP_INTRSEC, krule, cutoff, ovr, FILTER_RAW), false, out, &run);
    if (acc_owned) gt4hip_list_free (acc_owned);
    acc_owned = NULL;
    if (!rc) {
      res->device_ms += run.device_ms;
      res->records_read += acc->n_words + lists[k]->n_words;
      res->records_written += run.n_words[1];
      acc = acc_owned = out[1];
    }
  }
  if (!rc) rc = nway_final (ctx, acc, lists[n_lists - 1], nway_params (GT4HIP_OP_INTRSEC, krule, cutoff, ovr, FILTER_RESULT), 1, count_only != 0, res);
  if (acc_owned) gt4hip_list_free (acc_owned);
  return rc;
}

/* ------------------------------------------------------------------ per-key count table (gt4_union) */

/* A column of the table is itself a MERGE: with K the table's key list (every key of list j is in
 * K), union (K, L_j) under rule SECOND, every key kept, is K's keys with L_j's count or 0 -- a list
 * aligned with K, whose count column is copied into the table.  One streaming pass over K and L_j
 * per list instead of a binary search over L_j per key. */
static int table_column_by_union (gt4hip_context *ctx, const gt4hip_list *keys, const gt4hip_list *lj, gt4hip_list *tmp, gt4hip_count_table *t,
                                  uint32_t column)
{
  PairParams p = nway_params (GT4HIP_OP_UNION, GT4HIP_RULE_SECOND, 0, 0, FILTER_RAW);
  uint32_t *dst[4] = { (uint32_t *) tmp->dev, NULL, NULL, NULL };
  PairRun run;
  int rc = run_pair (ctx, (const uint32_t *) keys->dev, keys->n_words, (const uint32_t *) lj->dev, lj->n_words, p, false, dst, &run);
  if (rc) return rc;
  if (run.n_words[0] != keys->n_words) return gt4hip_fail (ctx, GT4HIP_EINTERNAL, "count table: a list holds keys outside the key list");
  HIPCHK (ctx, launch_extract_column (ctx->stream, (const uint32_t *) tmp->dev, keys->n_words, (uint32_t *) t->device_counts, t->n_lists, column));
  return GT4HIP_OK;
}

/* Device memory that comes from the context's block pool and goes back to it (big hipMalloc / hipFree
 * pairs per call stall for a second every few calls on this driver): a list object owns the block. */
int gt4hip_block_alloc (gt4hip_context *ctx, size_t bytes, void **dev, void **owner)
{
  gt4hip_list *l = NULL;
  const int rc = gt4hip_list_new (ctx, bytes / GT4HIP_RECORD_BYTES + 1, 1, &l);
  if (rc) return rc;
  *dev = l->dev;
  *owner = l;
  return GT4HIP_OK;
}

void gt4hip_block_free (void *owner)
{
  if (owner) gt4hip_list_free ((gt4hip_list *) owner);
}

int gt4hip_table_alloc (gt4hip_context *ctx, gt4hip_count_table *table, uint64_t n, uint32_t n_lists)
{
  int rc = gt4hip_block_alloc (ctx, (size_t) n * 8, &table->device_keys, &table->owner[0]);
  if (!rc) rc = gt4hip_block_alloc (ctx, (size_t) n * n_lists * 4, &table->device_counts, &table->owner[1]);
  if (rc) {
    gt4hip_table_free (table);
    return gt4hip_fail (ctx, GT4HIP_ENOMEM, "count table allocation failed (%llu keys x %u lists)", (unsigned long long) n, n_lists);
  }
  return GT4HIP_OK;
}

static int table_alloc (gt4hip_context *ctx, gt4hip_count_table *table, uint64_t n, uint32_t n_lists) { return gt4hip_table_alloc (ctx, table, n, n_lists); }

/* ---- ragged tables (see gt4hip_count_table in include/gt4hip.h) */
namespace {
struct TableRagged {
  uint64_t tiles;
  unsigned long long *compact, *padded; /* device, tiles + 1 entries each, one block */
  void *owner;
};

/* rows [first, first + count) of a ragged table, gathered: one thread per row finds its tile (the last one whose
 * compact base is not beyond the row) and copies the row from where the tile's rows lie */
__global__ void k_table_gather (const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ counts, uint32_t n_lists,
                                const unsigned long long *__restrict__ compact, const unsigned long long *__restrict__ padded, uint64_t tiles,
                                uint64_t first, uint64_t count, unsigned long long *__restrict__ out_keys, uint32_t *__restrict__ out_counts)
{
  const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint64_t r = first + i;
  uint64_t lo = 0, hi = tiles; /* compact[lo] <= r < compact[hi] (compact[tiles] = n_keys) */
  while (hi - lo > 1) {
    const uint64_t mid = (lo + hi) >> 1;
    if (compact[mid] <= r) lo = mid;
    else hi = mid;
  }
  const uint64_t src = padded[lo] + (r - compact[lo]);
  if (out_keys) out_keys[i] = keys[src];
  if (out_counts)
    for (uint32_t j = 0; j < n_lists; j++) out_counts[i * n_lists + j] = counts[src * n_lists + j];
}
}  // namespace

int gt4hip_table_set_ragged (gt4hip_context *ctx, gt4hip_count_table *table, uint64_t tiles)
{
  TableRagged *r = new (std::nothrow) TableRagged ();
  if (!r) return gt4hip_fail (ctx, GT4HIP_ENOMEM, "host allocation failed");
  void *dev = NULL;
  if (gt4hip_block_alloc (ctx, (size_t) (tiles + 1) * 16, &dev, &r->owner)) {
    delete r;
    return gt4hip_fail (ctx, GT4HIP_ENOMEM, "count table index of %llu tiles", (unsigned long long) tiles);
  }
  r->tiles = tiles;
  r->compact = (unsigned long long *) dev;
  r->padded = r->compact + tiles + 1;
  table->ragged = r;
  return GT4HIP_OK;
}

void *gt4hip_table_compact_bases (gt4hip_count_table *table) { return table && table->ragged ? ((TableRagged *) table->ragged)->compact : NULL; }
void *gt4hip_table_padded_bases (gt4hip_count_table *table) { return table && table->ragged ? ((TableRagged *) table->ragged)->padded : NULL; }

static hipError_t table_gather (gt4hip_context *ctx, const gt4hip_count_table *t, uint64_t first, uint64_t count, void *out_keys, void *out_counts)
{
  const TableRagged *r = (const TableRagged *) t->ragged;
  hipLaunchKernelGGL (k_table_gather, dim3 ((unsigned) ((count + 255) / 256)), dim3 (256), 0, ctx->stream, (const unsigned long long *) t->device_keys,
                      (const uint32_t *) t->device_counts, t->n_lists, r->compact, r->padded, r->tiles, first, count, (unsigned long long *) out_keys, (uint32_t *) out_counts);
  return hipGetLastError ();
}

extern "C" int gt4hip_table_compact (gt4hip_context *ctx, gt4hip_count_table *t)
{
  if (!ctx || !t) return GT4HIP_EINVAL;
  if (!t->ragged) return GT4HIP_OK;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  gt4hip_count_table c;
  memset (&c, 0, sizeof c);
  c.n_keys = t->n_keys;
  c.n_lists = t->n_lists;
  if (t->n_keys) {
    const int rc = gt4hip_table_alloc (ctx, &c, t->n_keys, t->n_lists);
    if (rc) return rc;
    /* (at most 2^31 blocks of 256 rows per launch) */
    for (uint64_t first = 0; first < t->n_keys; first += 1ull << 32) {
      const uint64_t cnt = t->n_keys - first < (1ull << 32) ? t->n_keys - first : (1ull << 32);
      const hipError_t e = table_gather (ctx, t, first, cnt, (char *) c.device_keys + first * 8, (char *) c.device_counts + first * t->n_lists * 4);
      if (e != hipSuccess) {
        gt4hip_table_free (&c);
        return gt4hip_fail (ctx, GT4HIP_EHIP, "gt4hip_table_compact: %s", hipGetErrorString (e));
      }
    }
    HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
  }
  gt4hip_table_free (t);
  *t = c;
  return GT4HIP_OK;
}

extern "C" int gt4hip_union_table (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n_lists, gt4hip_count_table *table)
{
  if (!ctx || !lists || !n_lists || !table) return GT4HIP_EINVAL;
  const auto t_begin = std::chrono::steady_clock::now ();
  struct Stamp { /* wall time of the whole call (several launches and read-backs), for bench.py --workload table */
    gt4hip_context *c;
    std::chrono::steady_clock::time_point t0;
    ~Stamp () { c->table_ms = std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - t0).count (); }
  } stamp = { ctx, t_begin };
  memset (table, 0, sizeof *table);
  table->n_lists = n_lists;
  /* up to 32 non-empty lists (eight with option "kway_max" = 8): ONE launch of the N-way tile kernel writes keys and counts
   * directly, every tile's rows where its records start (a ragged table: one row slot per input RECORD, up to n_lists
   * times the distinct keys) */
  const uint32_t table_width = ctx->kway_max == 8 ? 8u : 32u;
  if (ctx->kway_enabled) {
    const gt4hip_list *work[32];
    uint32_t cols[32], k = 0;
    bool fits = true;
    for (uint32_t j = 0; j < n_lists && fits; j++) {
      if (!lists[j]) return GT4HIP_EINVAL;
      if (!lists[j]->n_words) continue;
      if (k == table_width || lists[j]->word_length != lists[0]->word_length) fits = false;
      else {
        work[k] = lists[j];
        cols[k++] = j;
      }
    }
    if (fits && k >= 2) {
      HIPCHK (ctx, hipSetDevice (ctx->device));
      int used = 0;
      const int rc = gt4hip_nway_table (ctx, work, k, cols, table, 0, 0, &used);
      /* the ragged table did not fit: the path below needs one row per DISTINCT key only */
      if (rc && rc != GT4HIP_ENOMEM) return rc;
      if (rc == GT4HIP_ENOMEM) {
        used = 0;
        ctx->err[0] = 0;
      }
      if (used) {
        ctx->kway_calls++;
        return GT4HIP_OK;
      }
      memset (table, 0, sizeof *table);
      table->n_lists = n_lists;
    }
  }
  /* otherwise: all distinct keys ascending = N-way union with nothing filtered out (count >= 0), then a merge per column */
  gt4hip_multi_result u;
  memset (&u, 0, sizeof u);
  int rc = gt4hip_union_multi (ctx, lists, n_lists, 0, GT4HIP_RULE_MAX, 0, 0, &u);
  if (rc) return rc;
  const uint64_t n = u.n_words;
  table->n_keys = n;
  if (!n) {
    gt4hip_list_free (u.out);
    return GT4HIP_OK;
  }
  uint64_t longest = 0;
  for (uint32_t j = 0; j < n_lists; j++)
    if (lists[j]->n_words > longest) longest = lists[j]->n_words;
  gt4hip_list *tmp = NULL;
  rc = table_alloc (ctx, table, n, n_lists);
  if (!rc) rc = gt4hip_list_new (ctx, n + longest, lists[0]->word_length, &tmp);
  if (!rc) {
    hipError_t e = launch_extract_keys (ctx->stream, (const uint32_t *) u.out->dev, n, (unsigned long long *) table->device_keys);
    if (e != hipSuccess) rc = gt4hip_fail (ctx, GT4HIP_EHIP, "count table kernels failed: %s", hipGetErrorString (e));
  }
  for (uint32_t j = 0; j < n_lists && !rc; j++) rc = table_column_by_union (ctx, u.out, lists[j], tmp, table, j);
  if (!rc && hipStreamSynchronize (ctx->stream) != hipSuccess) rc = gt4hip_fail (ctx, GT4HIP_EHIP, "count table kernels failed");
  if (tmp) gt4hip_list_free (tmp);
  gt4hip_list_free (u.out);
  if (rc) gt4hip_table_free (table);
  return rc;
}

/* keys = the keys of lists[0]; column j = the count of each of them in list j (0 when absent), or,
 * with `presence`, 1 when list j holds the key and 0 when it does not (a count may be 0 itself).
 * Two merges per list: lists[0] n L_j keeping L_j's count, then aligned with lists[0] as above. */
extern "C" int gt4hip_probe_table_ex (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n_lists, int presence, gt4hip_count_table *table)
{
  if (!ctx || !lists || !n_lists || !table || !lists[0]) return GT4HIP_EINVAL;
  for (uint32_t j = 0; j < n_lists; j++)
    if (!lists[j]) return GT4HIP_EINVAL;
  memset (table, 0, sizeof *table);
  table->n_lists = n_lists;
  const gt4hip_list *base = lists[0];
  const uint64_t n = base->n_words;
  table->n_keys = n;
  if (!n) return GT4HIP_OK;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  /* up to 32 non-empty lists (the base first; eight with option "kway_max" = 8): one launch of the N-way tile kernel */
  if (ctx->kway_enabled) {
    const uint32_t table_width = ctx->kway_max == 8 ? 8u : 32u;
    const gt4hip_list *work[32];
    uint32_t cols[32], k = 0;
    bool fits = true;
    for (uint32_t j = 0; j < n_lists && fits; j++) {
      if (j && !lists[j]->n_words) continue;
      if (k == table_width || lists[j]->word_length != base->word_length) fits = false;
      else {
        work[k] = lists[j];
        cols[k++] = j;
      }
    }
    if (fits && k >= 2) {
      int used = 0;
      const int rc = gt4hip_nway_table (ctx, work, k, cols, table, 1, presence, &used);
      if (rc) return rc;
      if (used) {
        ctx->kway_calls++;
        return GT4HIP_OK;
      }
      memset (table, 0, sizeof *table);
      table->n_lists = n_lists;
      table->n_keys = n;
    }
  }
  gt4hip_list *tmp = NULL, *inter = NULL;
  int rc = table_alloc (ctx, table, n, n_lists);
  if (!rc) rc = gt4hip_list_new (ctx, 2 * n, base->word_length, &tmp);
  if (!rc) rc = gt4hip_list_new (ctx, n, base->word_length, &inter);
  if (!rc) {
    hipError_t e = launch_extract_keys (ctx->stream, (const uint32_t *) base->dev, n, (unsigned long long *) table->device_keys);
    if (e != hipSuccess) rc = gt4hip_fail (ctx, GT4HIP_EHIP, "count table kernels failed: %s", hipGetErrorString (e));
  }
  for (uint32_t j = 0; j < n_lists && !rc; j++) {
    PairParams p = nway_params (GT4HIP_OP_INTRSEC, presence ? GT4HIP_RULE_NUMBER : GT4HIP_RULE_SECOND, 0, 1, FILTER_RAW);
    uint32_t *dst[4] = { NULL, (uint32_t *) inter->dev, NULL, NULL };
    PairRun run;
    rc = run_pair (ctx, (const uint32_t *) base->dev, n, (const uint32_t *) lists[j]->dev, lists[j]->n_words, p, false, dst, &run);
    if (rc) break;
    inter->n_words = run.n_words[1];
    rc = table_column_by_union (ctx, base, inter, tmp, table, j);
  }
  if (!rc && hipStreamSynchronize (ctx->stream) != hipSuccess) rc = gt4hip_fail (ctx, GT4HIP_EHIP, "count table kernels failed");
  if (tmp) gt4hip_list_free (tmp);
  if (inter) gt4hip_list_free (inter);
  if (rc) gt4hip_table_free (table);
  return rc;
}

extern "C" int gt4hip_probe_table (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n_lists, gt4hip_count_table *table)
{
  return gt4hip_probe_table_ex (ctx, lists, n_lists, 0, table);
}

extern "C" int gt4hip_table_download (gt4hip_context *ctx, const gt4hip_count_table *t, uint64_t first, uint64_t count,
                                       uint64_t *host_keys, uint32_t *host_counts)
{
  if (!ctx || !t || first > t->n_keys || count > t->n_keys - first) return GT4HIP_EINVAL;
  if (!count) return GT4HIP_OK;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  if (t->ragged) {
    /* gathered into a staging block on the device, then copied */
    char *tmp = NULL;
    void *owner = NULL;
    const size_t kb = (size_t) count * 8, cb = (size_t) count * t->n_lists * 4;
    if (gt4hip_block_alloc (ctx, kb + cb, (void **) &tmp, &owner)) return gt4hip_fail (ctx, GT4HIP_ENOMEM, "gt4hip_table_download: %zu bytes of staging", kb + cb);
    hipError_t e = table_gather (ctx, t, first, count, host_keys ? tmp : NULL, host_counts ? tmp + kb : NULL);
    if (e == hipSuccess && host_keys) e = hipMemcpyAsync (host_keys, tmp, kb, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && host_counts) e = hipMemcpyAsync (host_counts, tmp + kb, cb, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize (ctx->stream);
    gt4hip_block_free (owner);
    if (e != hipSuccess) return gt4hip_fail (ctx, GT4HIP_EHIP, "gt4hip_table_download: %s", hipGetErrorString (e));
    return GT4HIP_OK;
  }
  if (host_keys) HIPCHK (ctx, hipMemcpyAsync (host_keys, (const char *) t->device_keys + first * 8, (size_t) count * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (host_counts)
    HIPCHK (ctx, hipMemcpyAsync (host_counts, (const char *) t->device_counts + first * t->n_lists * 4, (size_t) count * t->n_lists * 4,
                                 hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
  return GT4HIP_OK;
}

extern "C" void gt4hip_table_free (gt4hip_count_table *t)
{
  if (!t) return;
  if (t->owner[0]) gt4hip_block_free (t->owner[0]);
  else if (t->device_keys) hipFree (t->device_keys);
  if (t->owner[1]) gt4hip_block_free (t->owner[1]);
  else if (t->device_counts) hipFree (t->device_counts);
  if (t->ragged) {
    TableRagged *r = (TableRagged *) t->ragged;
    if (r->owner) gt4hip_block_free (r->owner);
    delete r;
  }
  t->ragged = NULL;
  t->device_keys = t->device_counts = NULL;
  t->owner[0] = t->owner[1] = NULL;
  t->n_keys = 0;
}
