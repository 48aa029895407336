/*
 * gt4hip_stub.c -- TEST INFRASTRUCTURE ONLY: a CPU stand-in for the part of include/gt4hip.h that the
 * C host (csrc/gt4_glistcompare_cli.c, gt4_shard.c, gt4_listfile.c) calls, so that the host's
 * argv handling, key-range planning, fork / barrier / semaphore pipeline, shared totals, pwrite
 * extents, header back-patching and failure paths run WITHOUT a GPU -- under AddressSanitizer,
 * UndefinedBehaviorSanitizer and ThreadSanitizer, with two or three worker processes
 * (tests/test_host_sanitizers.py; VERDICT round 2, "Next" 8; sanitizers never run on the GPU box).
 *
 * "Device lists" are host arrays; the set operations are the CPU oracle's (oracle/gt4_oracle.c: test
 * infrastructure calling test infrastructure).  The gather "collective" goes through files under
 * /dev/shm named by the communicator id.  Nothing under genometester4_amd/ links or loads this file.
 *
 * Failure injection: GT4HIP_STUB_FAIL=<what>:<rank> with what = create | alloc_gather | gatherv | merge
 * makes that call fail in worker <rank> (recognised by the device number it asks for: gt4_shard.c
 * gives worker r device r mod count, and the stub reports GT4HIP_STUB_DEVICES devices).
 */
#define _GNU_SOURCE
#include "../../include/gt4hip.h"
#include "../../oracle/gt4_oracle.h"

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

struct gt4hip_context {
  int device;
  char err[512];
  char info[64];
};

struct gt4hip_list {
  gt4hip_context *ctx;
  uint8_t *rec;
  uint64_t n_words, capacity;
  uint32_t word_length;
};

static __thread char g_err[512]; /* per thread, as the library's: several threads may fail to create a context at once */

static int fail (gt4hip_context *ctx, int code, const char *msg)
{
  snprintf (ctx ? ctx->err : g_err, 512, "%s", msg);
  return code;
}

static int inject (const char *what, int rank)
{
  const char *e = getenv ("GT4HIP_STUB_FAIL");
  if (!e) return 0;
  const size_t n = strlen (what);
  return !strncmp (e, what, n) && e[n] == ':' && atoi (e + n + 1) == rank;
}

int gt4hip_device_count (void)
{
  const char *e = getenv ("GT4HIP_STUB_DEVICES");
  return e ? atoi (e) : 1;
}

int gt4hip_create (int device, gt4hip_context **out)
{
  *out = NULL;
  if (device < 0 || device >= gt4hip_device_count ()) return fail (NULL, GT4HIP_ENODEVICE, "gt4hip_create: device not present (stub)");
  if (inject ("create", device)) return fail (NULL, GT4HIP_ENODEVICE, "gt4hip_create: injected failure (stub)");
  gt4hip_context *c = (gt4hip_context *) calloc (1, sizeof *c);
  if (!c) return fail (NULL, GT4HIP_ENOMEM, "host allocation failed");
  c->device = device;
  snprintf (c->info, sizeof c->info, "CPU stub|none|0|0");
  *out = c;
  return GT4HIP_OK;
}

void gt4hip_destroy (gt4hip_context *ctx) { free (ctx); }
const char *gt4hip_last_error (const gt4hip_context *ctx) { return ctx ? ctx->err : g_err; }
const char *gt4hip_strerror (int code) { return code ? "error (stub)" : "ok"; }
const char *gt4hip_device_info (const gt4hip_context *ctx) { return ctx ? ctx->info : ""; }

int gt4hip_device_memory (gt4hip_context *ctx, uint64_t *free_bytes, uint64_t *total_bytes)
{
  (void) ctx;
  const char *e = getenv ("GT4HIP_STUB_FREE");
  const uint64_t f = e ? strtoull (e, NULL, 10) : (1ull << 32);
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = f;
  return GT4HIP_OK;
}

int gt4hip_set_option (gt4hip_context *ctx, const char *name, int64_t value)
{
  (void) ctx;
  (void) name;
  (void) value;
  return GT4HIP_OK;
}

static int list_new (gt4hip_context *ctx, uint64_t capacity, uint32_t wl, gt4hip_list **out)
{
  gt4hip_list *l = (gt4hip_list *) calloc (1, sizeof *l);
  if (!l) return fail (ctx, GT4HIP_ENOMEM, "host allocation failed");
  l->rec = (uint8_t *) malloc (capacity ? capacity * 12 : 1);
  if (!l->rec) {
    free (l);
    return fail (ctx, GT4HIP_ENOMEM, "host allocation failed");
  }
  l->ctx = ctx;
  l->n_words = l->capacity = capacity;
  l->word_length = wl;
  *out = l;
  return GT4HIP_OK;
}

int gt4hip_list_alloc (gt4hip_context *ctx, uint64_t capacity, uint32_t wl, gt4hip_list **out)
{
  if (inject ("alloc_gather", ctx->device)) return fail (ctx, GT4HIP_ENOMEM, "hipMalloc failed: injected (stub)");
  return list_new (ctx, capacity, wl, out);
}

int gt4hip_list_upload (gt4hip_context *ctx, const void *host, uint64_t n, uint32_t wl, gt4hip_list **out)
{
  int rc = list_new (ctx, n, wl, out);
  if (!rc && n) memcpy ((*out)->rec, host, n * 12);
  return rc;
}

int gt4hip_list_upload_fd (gt4hip_context *ctx, int fd, uint64_t off, uint64_t n, uint32_t wl, gt4hip_list **out)
{
  int rc = list_new (ctx, n, wl, out);
  if (rc) return rc;
  uint64_t done = 0;
  while (done < n * 12) {
    const ssize_t r = pread (fd, (*out)->rec + done, n * 12 - done, (off_t) (off + done));
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) {
      gt4hip_list_free (*out);
      *out = NULL;
      return fail (ctx, GT4HIP_EIO, r ? "upload: file I/O failed (stub)" : "upload: file I/O failed: the file is shorter than its header promises (stub)");
    }
    done += (uint64_t) r;
  }
  return GT4HIP_OK;
}

int gt4hip_list_upload_index (gt4hip_context *ctx, const void *kmers, uint64_t n, uint64_t num_locations, uint32_t wl, gt4hip_list **out)
{
  int rc = list_new (ctx, n, wl, out);
  if (rc) return rc;
  const uint8_t *k = (const uint8_t *) kmers;
  for (uint64_t i = 0; i < n; i++) {
    uint64_t word, loc, next = num_locations;
    memcpy (&word, k + 16 * i, 8);
    memcpy (&loc, k + 16 * i + 8, 8);
    if (i + 1 < n) memcpy (&next, k + 16 * (i + 1) + 8, 8);
    const uint32_t c = (uint32_t) (next - loc);
    memcpy ((*out)->rec + 12 * i, &word, 8);
    memcpy ((*out)->rec + 12 * i + 8, &c, 4);
  }
  return GT4HIP_OK;
}

void gt4hip_list_free (gt4hip_list *l)
{
  if (!l) return;
  free (l->rec);
  free (l);
}

int gt4hip_list_is_sorted (gt4hip_context *ctx, const gt4hip_list *l, int *sorted)
{
  (void) ctx;
  *sorted = 1;
  for (uint64_t i = 0; i + 1 < l->n_words; i++) {
    uint64_t a, b;
    memcpy (&a, l->rec + 12 * i, 8);
    memcpy (&b, l->rec + 12 * (i + 1), 8);
    if (a >= b) *sorted = 0;
  }
  return GT4HIP_OK;
}

int gt4hip_list_download_range (gt4hip_context *ctx, const gt4hip_list *l, uint64_t first, uint64_t count, void *host)
{
  (void) ctx;
  if (first > l->n_words || count > l->n_words - first) return GT4HIP_EINVAL;
  memcpy (host, l->rec + 12 * first, count * 12);
  return GT4HIP_OK;
}

int gt4hip_list_write_fd (gt4hip_context *ctx, const gt4hip_list *l, uint64_t first, uint64_t count, int fd, uint64_t off)
{
  uint64_t done = 0;
  while (done < count * 12) {
    const ssize_t r = pwrite (fd, l->rec + 12 * first + done, count * 12 - done, (off_t) (off + done));
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) return fail (ctx, GT4HIP_EIO, "write: file I/O failed (stub)");
    done += (uint64_t) r;
  }
  return GT4HIP_OK;
}

int gt4hip_lists_write_fd (gt4hip_context *ctx, uint32_t n, const gt4hip_list *const lists[], const uint64_t first[], const uint64_t count[],
                           const int fds[], const uint64_t offs[])
{
  for (uint32_t i = 0; i < n; i++) {
    const int rc = gt4hip_list_write_fd (ctx, lists[i], first[i], count[i], fds[i], offs[i]);
    if (rc) return rc;
  }
  return GT4HIP_OK;
}

int gt4hip_compare (gt4hip_context *ctx, const gt4hip_list *a, const gt4hip_list *b, const gt4hip_compare_params *p, gt4hip_compare_result *res)
{
  if (inject ("merge", ctx->device)) return fail (ctx, GT4HIP_EINTERNAL, "merge: injected failure (stub)");
  uint8_t *out[4] = { NULL, NULL, NULL, NULL };
  gt4hip_list *made[4] = { NULL, NULL, NULL, NULL };
  for (int s = 0; s < 4; s++) {
    if (!((p->ops >> s) & 1u) || p->count_only) continue;
    if (list_new (ctx, a->n_words + b->n_words, a->word_length, &made[s])) return GT4HIP_ENOMEM;
    out[s] = made[s]->rec;
  }
  gt4o_stat st[4];
  gt4o_compare (a->rec, a->n_words, b->rec, b->n_words, p->ops, p->rule, p->cutoff, p->subtract, p->count_override, out, st);
  for (int s = 0; s < 4; s++) {
    res->n_words[s] = st[s].n_words;
    res->total_count[s] = st[s].total_count;
    res->out[s] = made[s];
    if (made[s]) made[s]->n_words = st[s].n_words;
  }
  return GT4HIP_OK;
}

static int multi (gt4hip_context *ctx, int is_union, const gt4hip_list *const lists[], uint32_t n, uint32_t cutoff, int32_t rule, uint32_t ovr,
                  int32_t count_only, gt4hip_multi_result *res)
{
  if (inject ("merge", ctx->device)) return fail (ctx, GT4HIP_EINTERNAL, "merge: injected failure (stub)");
  const uint8_t **recs = (const uint8_t **) malloc (n * sizeof *recs);
  uint64_t *ns = (uint64_t *) malloc (n * sizeof *ns), total = 0;
  if (!recs || !ns) {
    free (recs);
    free (ns);
    return fail (ctx, GT4HIP_ENOMEM, "host allocation failed");
  }
  for (uint32_t i = 0; i < n; i++) {
    recs[i] = lists[i]->rec;
    ns[i] = lists[i]->n_words;
    total += ns[i];
  }
  gt4hip_list *o = NULL;
  if (!count_only && list_new (ctx, total, lists[0]->word_length, &o)) {
    free (recs);
    free (ns);
    return GT4HIP_ENOMEM;
  }
  gt4o_stat st = { 0, 0 };
  const int rc = is_union ? gt4o_union_multi (recs, ns, n, cutoff, rule, ovr, o ? o->rec : NULL, &st)
                          : gt4o_intersect_multi (recs, ns, n, cutoff, rule, ovr, o ? o->rec : NULL, &st);
  free (recs);
  free (ns);
  if (rc) {
    gt4hip_list_free (o);
    snprintf (ctx->err, sizeof ctx->err, is_union ? "union_multi: Invalid rule %u (only ADD, MAX and NUMBER allowed)"
                                                  : "intersect_multi: Invalid rule %u (only ADD, MIN, MAX and NUMBER allowed)", (unsigned) rule);
    return GT4HIP_ERULE;
  }
  if (o) o->n_words = st.n_words;
  res->n_words = st.n_words;
  res->total_count = st.total_count;
  res->out = o;
  return GT4HIP_OK;
}

int gt4hip_union_multi (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n, uint32_t cutoff, int32_t rule, uint32_t ovr, int32_t count_only,
                        gt4hip_multi_result *res)
{
  return multi (ctx, 1, lists, n, cutoff, rule, ovr, count_only, res);
}

int gt4hip_intersect_multi (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n, uint32_t cutoff, int32_t rule, uint32_t ovr, int32_t count_only,
                            gt4hip_multi_result *res)
{
  return multi (ctx, 0, lists, n, cutoff, rule, ovr, count_only, res);
}

/* ------------------------------------------------------------------ the "collective": files under /dev/shm */

struct gt4hip_comm {
  gt4hip_context *ctx;
  char id[40];
  int n_ranks, rank;
  unsigned long seq;
};

const char *gt4hip_comm_last_error (void) { return g_err; }

int gt4hip_comm_unique_id (void *id_out)
{
  /* unique across runs: a failed run leaves its exchange files behind, and process numbers come round again
   * (a later run that drew the same name would read the stale file before its peer has renamed the new one) */
  struct timespec ts;
  clock_gettime (CLOCK_REALTIME, &ts);
  memset (id_out, 0, GT4HIP_COMM_ID_BYTES);
  snprintf ((char *) id_out, 40, "gt4stub_%ld_%llx", (long) getpid (), (unsigned long long) ts.tv_sec * 1000000000ull + (unsigned long long) ts.tv_nsec);
  return GT4HIP_OK;
}

int gt4hip_comm_create (gt4hip_context *ctx, const void *id, int n_ranks, int rank, gt4hip_comm **comm)
{
  gt4hip_comm *c = (gt4hip_comm *) calloc (1, sizeof *c);
  if (!c) return fail (ctx, GT4HIP_ENOMEM, "host allocation failed");
  c->ctx = ctx;
  memcpy (c->id, id, sizeof c->id - 1);
  c->n_ranks = n_ranks;
  c->rank = rank;
  *comm = c;
  return GT4HIP_OK;
}

void gt4hip_comm_destroy (gt4hip_comm *c) { free (c); }

int gt4hip_comm_gatherv (gt4hip_comm *c, const gt4hip_list *local, const uint64_t counts[], int root, gt4hip_list *gathered)
{
  if (inject ("gatherv", c->rank)) return fail (c->ctx, GT4HIP_ECOMM, "gatherv: injected failure (stub)");
  const unsigned long seq = c->seq++;
  char name[128], tmp[140];
  if (c->rank != root) {
    snprintf (name, sizeof name, "/dev/shm/%s_%lu_%d", c->id, seq, c->rank);
    snprintf (tmp, sizeof tmp, "%s.part", name);
    const int fd = open (tmp, O_WRONLY | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) return fail (c->ctx, GT4HIP_ECOMM, "gatherv: cannot create the exchange file (stub)");
    uint64_t done = 0;
    const uint64_t bytes = counts[c->rank] * 12;
    while (done < bytes) {
      const ssize_t r = write (fd, local->rec + done, bytes - done);
      if (r <= 0) {
        close (fd);
        return fail (c->ctx, GT4HIP_ECOMM, "gatherv: write failed (stub)");
      }
      done += (uint64_t) r;
    }
    close (fd);
    if (rename (tmp, name)) return fail (c->ctx, GT4HIP_ECOMM, "gatherv: rename failed (stub)");
    return GT4HIP_OK;
  }
  uint64_t at = 0;
  for (int r = 0; r < c->n_ranks; r++) {
    const uint64_t bytes = counts[r] * 12;
    if (r == root) {
      if (bytes) memcpy (gathered->rec + at, local->rec, bytes);
    } else {
      snprintf (name, sizeof name, "/dev/shm/%s_%lu_%d", c->id, seq, r);
      int fd = -1;
      for (int tries = 0; tries < 3000 && fd < 0; tries++) { /* the peer publishes by rename: wait for it (30 s at most) */
        fd = open (name, O_RDONLY);
        if (fd < 0) usleep (10000);
      }
      if (fd < 0) return fail (c->ctx, GT4HIP_ECOMM, "gatherv: a peer never sent its records (stub)");
      uint64_t done = 0;
      while (done < bytes) {
        const ssize_t rd = read (fd, gathered->rec + at + done, bytes - done);
        if (rd <= 0) {
          close (fd);
          unlink (name);
          return fail (c->ctx, GT4HIP_ECOMM, "gatherv: read failed (stub)");
        }
        done += (uint64_t) rd;
      }
      close (fd);
      unlink (name);
    }
    at += bytes;
  }
  gathered->n_words = at / 12;
  return GT4HIP_OK;
}
