/*
 * gt4_shard.c -- key-range sharded execution for the C host (see gt4_shard.h).
 *
 * Every set operation of glistcompare is key-local, so the merged key sequence can be cut at
 * arbitrary keys: chunk c holds, of every input, the records with keys in [cut[c], cut[c+1]), and
 * the chunks' outputs concatenated in chunk order are the complete sorted outputs.  Chunk c is
 * owned by worker c mod G (one worker process per GPU, forked before any HIP call).  A worker runs
 * three threads over its chunks, each with its own library context on the worker's GPU:
 *
 *   loader   file -> HBM           gt4hip_list_upload_fd (pinned staging, several copy threads)
 *   merger   the operation itself  gt4hip_compare / gt4hip_union_multi / gt4hip_intersect_multi
 *   writer   HBM -> file           gt4hip_list_write_fd at byte 48 + 12 * (records of all earlier chunks)
 *
 * Two chunks are in flight per worker, so loading chunk i+1 and writing chunk i-1 overlap the merge of
 * chunk i.  The only exchange between workers is the chunks' (n_words, total_count) per output,
 * through a shared memory block (the all-gather of the header totals); by default
 * every worker then writes its own extents of the output files (pwrite), with GT4HIP_GATHER=rccl the
 * chunks of a round are gathered on worker 0 over RCCL (gt4hip_comm_gatherv) and written there.
 */
#define _GNU_SOURCE
#include "gt4_shard.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <semaphore.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#define MAX_RANKS 64
#define SLOTS 2

/* ------------------------------------------------------------------ host-side views of the inputs */

uint64_t gt4_listfile_key_at (const GT4ListFile *lf, uint64_t idx)
{
  uint64_t k;
  if (lf->index_kmers) memcpy (&k, lf->index_kmers + 16 * idx, 8);
  else memcpy (&k, lf->records + 12 * idx, 8);
  return k;
}

uint64_t gt4_listfile_lower_bound (const GT4ListFile *lf, uint64_t key)
{
  uint64_t lo = 0, hi = lf->header.n_words;
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if (gt4_listfile_key_at (lf, mid) < key) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

/* ------------------------------------------------------------------ shared between the workers */

/* cross-process flags: release stores behind the data they guard, acquire loads in front of it */
typedef struct {
  uint32_t done;
  uint64_t n[4], t[4];
} ChunkTotals;

static int chunk_done (const ChunkTotals *c) { return __atomic_load_n (&c->done, __ATOMIC_ACQUIRE) != 0; }
static void chunk_set_done (ChunkTotals *c) { __atomic_store_n (&c->done, 1u, __ATOMIC_RELEASE); }

typedef struct {
  int failed;                   /* any worker: stop */
  int rule_rejected;
  char message[512];
  uint64_t limits[MAX_RANKS];   /* every worker's memory budget (bytes); the plan is made from the smallest */
  uint64_t hbm_limit;           /* the budget the plan was made from (worker 0, for the record) */
  unsigned char comm_id[GT4HIP_COMM_ID_BYTES];
  pthread_barrierattr_t bar_attr;
  pthread_barrier_t bar;
  unsigned int n_chunks;
  ChunkTotals chunk[];
} Shared;

typedef struct {
  unsigned int n_chunks;
  uint64_t *cut; /* [n_files][n_chunks + 1] record indices */
} Plan;

static uint64_t plan_cut (const Plan *p, unsigned int f, unsigned int c) { return p->cut[(size_t) f * (p->n_chunks + 1) + c]; }

static unsigned int n_streams (const GT4ShardJob *job)
{
  if (job->mode != GT4_SHARD_PAIR) return 1;
  unsigned int n = 0;
  for (int s = 0; s < 4; s++) n += (job->prm.ops >> s) & 1u;
  return n ? n : 1;
}

/* Cuts at keys of the longest input, at equal strides; the number of chunks is the smallest
 * multiple of the worker count for which every chunk fits the budget (doubling until it does). */
static int make_plan (const GT4ShardJob *job, uint64_t hbm_limit, Plan *plan)
{
  unsigned int longest = 0;
  uint64_t total = 0;
  for (unsigned int f = 0; f < job->n_files; f++) {
    total += job->files[f].header.n_words;
    if (job->files[f].header.n_words > job->files[longest].header.n_words) longest = f;
  }
  /* device bytes per input record of a chunk in flight: two input slots, two output sets (each
   * output at most the chunk's inputs), the N-way tree's intermediates, the gather buffer */
  uint64_t per_record = 12ull * (SLOTS + SLOTS * n_streams (job) + (job->mode == GT4_SHARD_PAIR ? 0 : 2) +
                                 (job->gather_rccl ? (uint64_t) n_streams (job) * job->n_ranks : 0));
  uint64_t budget = hbm_limit / per_record;
  if (budget < 1) budget = 1;
  const unsigned int G = (unsigned int) job->n_ranks;
  uint64_t want = (total + budget - 1) / budget;
  if (want < G) want = G;
  want = (want + G - 1) / G * G;
  const uint64_t n_long = job->files[longest].header.n_words;
  const uint64_t cap = (uint64_t) (1u << 22) / G * G; /* the shared block holds 2^22 chunk records; every worker owns equally many chunks */
  for (;;) {
    if (want > cap) want = cap;
    const unsigned int C = (unsigned int) want;
    if ((uint64_t) job->n_files * ((uint64_t) C + 1) > (1ull << 28)) return 2; /* a cut table beyond 2 GiB: the budget is far too small for these inputs */
    uint64_t *cut = (uint64_t *) malloc ((size_t) job->n_files * (C + 1) * sizeof (uint64_t));
    if (!cut) return 1;
    for (unsigned int c = 0; c <= C; c++) {
      const uint64_t idx = c == C ? n_long : (uint64_t) (((unsigned __int128) n_long * c) / C);
      for (unsigned int f = 0; f < job->n_files; f++) {
        uint64_t v;
        if (c == 0) v = 0;
        else if (c == C || idx >= n_long) v = job->files[f].header.n_words;
        else if (f == longest) v = idx;
        else v = gt4_listfile_lower_bound (&job->files[f], gt4_listfile_key_at (&job->files[longest], idx));
        cut[(size_t) f * (C + 1) + c] = v;
      }
    }
    uint64_t worst = 0;
    for (unsigned int c = 0; c < C; c++) {
      uint64_t sum = 0;
      for (unsigned int f = 0; f < job->n_files; f++) sum += cut[(size_t) f * (C + 1) + c + 1] - cut[(size_t) f * (C + 1) + c];
      if (sum > worst) worst = sum;
    }
    if (worst <= budget) {
      plan->n_chunks = C;
      plan->cut = cut;
      return 0;
    }
    free (cut);
    if ((uint64_t) C >= cap || (uint64_t) C >= 2 * n_long + G) return 2; /* finer cuts do not exist or would not help: the budget cannot be met */
    want *= 2;
  }
}

/* ------------------------------------------------------------------ one worker */

typedef struct {
  const GT4ShardJob *job;
  Shared *sh;
  const Plan *plan;
  int rank;
  int device;
  unsigned int n_mine;          /* chunks this worker owns: rank, rank + G, ... */
  /* per slot hand-over */
  sem_t slot_free[SLOTS];       /* writer -> loader: inputs and outputs of the slot are released */
  sem_t slot_loaded[SLOTS];     /* loader -> merger */
  sem_t slot_merged[SLOTS];     /* merger -> writer */
  sem_t outs_released[SLOTS];   /* writer -> merger: the slot's output lists may be freed */
  gt4hip_list *in[SLOTS][1024];
  gt4hip_list *out[SLOTS][4];
  uint64_t out_n[SLOTS][4];
  int out_fd[4];
  gt4hip_comm *comm;
  int local_failed;
  int check_sorted;            /* GT4HIP_CHECK_SORTED */
  uint64_t budget_bytes;       /* device bytes the plan was made for */
  double t_load, t_merge, t_write;
} Worker;

static double now_s (void)
{
  struct timespec ts;
  clock_gettime (CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static int shared_failed (const Shared *sh) { return __atomic_load_n (&sh->failed, __ATOMIC_ACQUIRE); }

static void worker_fail (Worker *w, const char *fmt, const char *detail)
{
  int expected = 0;
  /* the first failure writes the message; the flag is published behind it */
  static pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER; /* (this worker's threads) */
  pthread_mutex_lock (&lock);
  if (!shared_failed (w->sh)) {
    snprintf (w->sh->message, sizeof w->sh->message, fmt, detail);
    __atomic_compare_exchange_n (&w->sh->failed, &expected, 1, 0, __ATOMIC_RELEASE, __ATOMIC_RELAXED);
  }
  pthread_mutex_unlock (&lock);
  __atomic_store_n (&w->local_failed, 1, __ATOMIC_RELEASE);
}

static int stopped (const Worker *w) { return shared_failed (w->sh) || __atomic_load_n (&w->local_failed, __ATOMIC_ACQUIRE); }

static void *loader_main (void *arg)
{
  Worker *w = (Worker *) arg;
  const GT4ShardJob *job = w->job;
  gt4hip_context *ctx = NULL;
  int fds[1024];
  for (unsigned int f = 0; f < job->n_files; f++) fds[f] = -1;
  if (gt4hip_create (w->device, &ctx)) worker_fail (w, "Error: %s", gt4hip_last_error (NULL));
  /* the worker's contexts share one device: each may keep at most a third of the budget pooled */
  if (ctx) gt4hip_set_option (ctx, "pool_cap_mb", (int64_t) (w->budget_bytes / 3 >> 20));
  for (unsigned int f = 0; f < job->n_files && !stopped (w); f++) {
    if (job->files[f].index_kmers) continue;
    fds[f] = open (job->files[f].filename, O_RDONLY);
    if (fds[f] < 0) worker_fail (w, "Error: Cannot open %s", job->files[f].filename);
  }
  for (unsigned int i = 0; i < w->n_mine; i++) {
    const int slot = (int) (i % SLOTS);
    const unsigned int c = (unsigned int) w->rank + i * (unsigned int) job->n_ranks;
    sem_wait (&w->slot_free[slot]);
    const double t0 = now_s ();
    for (unsigned int f = 0; f < job->n_files; f++) {
      if (w->in[slot][f]) {
        gt4hip_list_free (w->in[slot][f]);
        w->in[slot][f] = NULL;
      }
      if (stopped (w)) continue;
      const GT4ListFile *lf = &job->files[f];
      const uint64_t first = plan_cut (w->plan, f, c), last = plan_cut (w->plan, f, c + 1);
      int rc;
      if (lf->index_kmers) {
        /* the count of the slice's last entry reaches to the next entry's first location */
        uint64_t end_loc = lf->index_locations;
        if (last < lf->header.n_words) memcpy (&end_loc, lf->index_kmers + 16 * last + 8, 8);
        rc = gt4hip_list_upload_index (ctx, lf->index_kmers + 16 * first, last - first, end_loc, job->word_length, &w->in[slot][f]);
      } else {
        rc = gt4hip_list_upload_fd (ctx, fds[f], lf->header.list_start + 12 * first, last - first, job->word_length, &w->in[slot][f]);
      }
      if (rc) worker_fail (w, "Error: uploading to the GPU failed: %s", gt4hip_last_error (ctx));
      else if (w->check_sorted && last > first) {
        /* GT4HIP_CHECK_SORTED=1: the chunk strictly ascending, and above the record in front of it (the
         * cuts were made by binary searches that trusted the order) */
        int sorted = 0;
        if (gt4hip_list_is_sorted (ctx, w->in[slot][f], &sorted) || !sorted ||
            (first > 0 && gt4_listfile_key_at (lf, first - 1) >= gt4_listfile_key_at (lf, first)))
          worker_fail (w, "Error: File %s is not sorted by k-mer (strictly ascending, unique)", lf->filename);
      }
    }
    w->t_load += now_s () - t0;
    sem_post (&w->slot_loaded[slot]);
  }
  /* the lists belong to this thread's context: release them once the merger is done with them */
  for (int slot = 0; slot < SLOTS; slot++) {
    if (w->n_mine > (unsigned int) slot) sem_wait (&w->slot_free[slot]);
    for (unsigned int f = 0; f < job->n_files; f++)
      if (w->in[slot][f]) gt4hip_list_free (w->in[slot][f]);
  }
  for (unsigned int f = 0; f < job->n_files; f++)
    if (fds[f] >= 0) close (fds[f]);
  if (ctx) gt4hip_destroy (ctx);
  return NULL;
}

static void *writer_main (void *arg)
{
  Worker *w = (Worker *) arg;
  const GT4ShardJob *job = w->job;
  gt4hip_context *ctx = NULL;
  const int writes = !job->prm.count_only;
  if (writes && gt4hip_create (w->device, &ctx)) worker_fail (w, "Error: %s", gt4hip_last_error (NULL));
  uint64_t start[4] = { 0, 0, 0, 0 }; /* records of chunks 0 .. summed - 1 per output: a running prefix */
  unsigned int summed = 0;
  for (unsigned int i = 0; i < w->n_mine; i++) {
    const int slot = (int) (i % SLOTS);
    const unsigned int c = (unsigned int) w->rank + i * (unsigned int) job->n_ranks;
    sem_wait (&w->slot_merged[slot]);
    const double t0 = now_s ();
    if (writes && !stopped (w)) {
      /* in gather mode worker 0 holds the whole round (chunks c .. c + G - 1) in its output lists */
      const unsigned int first_chunk = c;
      if (!job->gather_rccl || w->rank == 0) {
        /* records of all earlier chunks = where this one starts in each output file */
        for (; summed < first_chunk && !stopped (w); summed++) {
          while (!chunk_done (&w->sh->chunk[summed]) && !stopped (w)) usleep (50);
          if (stopped (w)) break;
          for (int s = 0; s < 4; s++) start[s] += w->sh->chunk[summed].n[s];
        }
        /* every output stream of the chunk in one call: the copy threads are dealt to the files */
        const gt4hip_list *wl[4];
        uint64_t wfirst[4], wcount[4], woff[4];
        int wfd[4];
        uint32_t nw = 0;
        for (int s = 0; s < 4; s++) {
          if (!w->out[slot][s] || !w->out_n[slot][s]) continue;
          wl[nw] = w->out[slot][s];
          wfirst[nw] = 0;
          wcount[nw] = w->out_n[slot][s];
          wfd[nw] = w->out_fd[s];
          woff[nw] = (job->out_name[s] ? 48 : job->out_base[s]) + 12 * start[s];
          nw++;
        }
        if (nw && !stopped (w) && gt4hip_lists_write_fd (ctx, nw, wl, wfirst, wcount, wfd, woff))
          worker_fail (w, "Error: writing results failed: %s", gt4hip_last_error (ctx));
      }
    }
    w->t_write += now_s () - t0;
    sem_post (&w->outs_released[slot]);
    sem_post (&w->slot_free[slot]);
  }
  if (ctx) gt4hip_destroy (ctx);
  return NULL;
}

/* the merge of one chunk; fills w->out[slot] / out_n[slot] and the chunk's totals */
static void merge_chunk (Worker *w, gt4hip_context *ctx, int slot, unsigned int c)
{
  const GT4ShardJob *job = w->job;
  ChunkTotals *ct = &w->sh->chunk[c];
  for (int s = 0; s < 4; s++) {
    ct->n[s] = ct->t[s] = 0;
    w->out[slot][s] = NULL;
    w->out_n[slot][s] = 0;
  }
  if (job->mode == GT4_SHARD_PAIR) {
    gt4hip_compare_result res;
    memset (&res, 0, sizeof res);
    if (job->prm.ops) {
      if (gt4hip_compare (ctx, w->in[slot][0], w->in[slot][1], &job->prm, &res)) {
        worker_fail (w, "Error: %s", gt4hip_last_error (ctx));
        return;
      }
    }
    for (int s = 0; s < 4; s++) {
      ct->n[s] = res.n_words[s];
      ct->t[s] = res.total_count[s];
      w->out[slot][s] = res.out[s];
      w->out_n[slot][s] = res.n_words[s];
    }
  } else {
    gt4hip_multi_result res;
    memset (&res, 0, sizeof res);
    const int rc = job->mode == GT4_SHARD_UNION_MULTI
                     ? gt4hip_union_multi (ctx, (const gt4hip_list *const *) w->in[slot], job->n_files, job->prm.cutoff, job->prm.rule,
                                           job->prm.count_override, job->prm.count_only, &res)
                     : gt4hip_intersect_multi (ctx, (const gt4hip_list *const *) w->in[slot], job->n_files, job->prm.cutoff, job->prm.rule,
                                               job->prm.count_override, job->prm.count_only, &res);
    if (rc == GT4HIP_ERULE) {
      __atomic_store_n (&w->sh->rule_rejected, 1, __ATOMIC_RELEASE);
      worker_fail (w, "%s", gt4hip_last_error (ctx));
      return;
    }
    if (rc) {
      worker_fail (w, "Error: %s", gt4hip_last_error (ctx));
      return;
    }
    ct->n[0] = res.n_words;
    ct->t[0] = res.total_count;
    w->out[slot][0] = res.out;
    w->out_n[slot][0] = res.n_words;
  }
}

/* GT4HIP_GATHER=rccl: the chunks of one round (c0 .. c0 + G - 1, one per worker) are gathered on
 * worker 0, which then holds the round's records of every output stream in rank = key order */
static void gather_round (Worker *w, gt4hip_context *ctx, int slot, unsigned int c0)
{
  const GT4ShardJob *job = w->job;
  const int G = job->n_ranks;
  if (c0 + (unsigned int) G > w->plan->n_chunks) {
    worker_fail (w, "Error: %s", "internal: a gather round reaches beyond the chunk plan");
    return;
  }
  for (int q = 0; q < G; q++)
    while (!chunk_done (&w->sh->chunk[c0 + q]) && !stopped (w)) usleep (20);
  if (stopped (w)) return;
  for (int s = 0; s < 4; s++) {
    if (!job->out_name[s]) continue;
    uint64_t counts[MAX_RANKS], total = 0;
    for (int q = 0; q < G; q++) total += counts[q] = w->sh->chunk[c0 + q].n[s];
    gt4hip_list *gathered = NULL;
    if (w->rank == 0 && gt4hip_list_alloc (ctx, total ? total : 1, job->word_length, &gathered)) {
      worker_fail (w, "Error: %s", gt4hip_last_error (ctx));
      return;
    }
    if (gt4hip_comm_gatherv (w->comm, w->out[slot][s], counts, 0, gathered)) {
      worker_fail (w, "Error: %s", gt4hip_last_error (ctx));
      return;
    }
    if (w->out[slot][s]) gt4hip_list_free (w->out[slot][s]);
    w->out[slot][s] = gathered; /* NULL on the other workers: nothing left for their writers */
    w->out_n[slot][s] = w->rank == 0 ? total : 0;
  }
}

static int worker_main (const GT4ShardJob *job, Shared *sh, int rank)
{
  Worker *w = (Worker *) calloc (1, sizeof (Worker));
  if (!w) return 1;
  FILE *errf = stderr;
  const double t_begin = now_s ();
  w->job = job;
  w->sh = sh;
  w->rank = rank;
  w->check_sorted = getenv ("GT4HIP_CHECK_SORTED") && atoi (getenv ("GT4HIP_CHECK_SORTED"));
  const int G = job->n_ranks;
  for (int s = 0; s < 4; s++) w->out_fd[s] = -1;

  /* ---- the device: worker r takes GPU r mod (visible devices) */
  gt4hip_context *ctx = NULL;
  {
    const int n_dev = gt4hip_device_count ();
    const char *dev = getenv ("GT4HIP_DEVICE");
    w->device = n_dev > 0 ? ((dev ? atoi (dev) : 0) + rank) % n_dev : 0;
    if (G == 1 && job->device_plus_1 > 0 && job->device_plus_1 <= n_dev) w->device = job->device_plus_1 - 1; /* (the device the caller measured its budget on) */
    if (gt4hip_create (w->device, &ctx)) worker_fail (w, "Error: %s", gt4hip_last_error (NULL));
    if (ctx && job->debug) fprintf (errf, "Worker %d of %d: device %d: %s\n", rank, G, w->device, gt4hip_device_info (ctx));
  }
  /* ---- the memory budget: the caller's, or 70 % of what THIS worker's device has free right now
   * (divided among the workers that share the device); the plan is made from the smallest budget of
   * all workers, so it is the same everywhere */
  {
    uint64_t limit = job->hbm_limit;
    if (!limit && ctx) {
      uint64_t free_b = 0, total_b = 0;
      gt4hip_device_memory (ctx, &free_b, &total_b);
      const int n_dev = gt4hip_device_count ();
      const int sharing = n_dev > 0 ? (G + n_dev - 1) / n_dev : G;
      limit = free_b / 10 * 7 / (uint64_t) (sharing > 0 ? sharing : 1);
      if (job->auto_budget) {
        /* inputs that fit the device, cut into chunks only to overlap file reads, merges and file writes:
         * about an eighth of (inputs + worst-case outputs) in flight -- counting runs get many small chunks
         * (the reads pace them), record-writing runs few large ones (measured on 2 x 2e9 records in tmpfs,
         * profiles/round3/r3_cli_e2e_2x2e9_with_reference.log and round4) -- never more than the 70 % above */
        uint64_t in_records = 0;
        for (unsigned int f = 0; f < job->n_files; f++) in_records += job->files[f].header.n_words;
        const uint64_t need = 12 * in_records * (job->mode == GT4_SHARD_PAIR ? 1 + (uint64_t) n_streams (job) : 4);
        uint64_t want = need / 8;
        if (want < (1ull << 30)) want = 1ull << 30;
        if (want < limit) limit = want;
        if (job->debug) fprintf (errf, "Inputs of %llu bytes: streaming in key-range chunks of about %llu device bytes\n", 12ull * (unsigned long long) in_records, (unsigned long long) limit);
      }
    }
    sh->limits[rank] = limit ? limit : (1ull << 30);
  }
  if (rank == 0 && job->gather_rccl && !stopped (w) && gt4hip_comm_unique_id (sh->comm_id)) worker_fail (w, "Error: %s", gt4hip_comm_last_error ());
  if (G > 1) pthread_barrier_wait (&sh->bar);
  uint64_t budget_bytes = sh->limits[0];
  for (int r = 1; r < G; r++)
    if (sh->limits[r] < budget_bytes) budget_bytes = sh->limits[r];
  Plan plan = { 0, NULL };
  {
    const int prc = make_plan (job, budget_bytes, &plan);
    if (prc == 1) worker_fail (w, "Error: %s", "out of memory while planning the chunks");
    else if (prc) worker_fail (w, "Error: %s", "the inputs cannot be cut into key-range chunks that fit the device memory budget (GT4HIP_HBM_LIMIT)");
  }
  if (rank == 0) {
    sh->hbm_limit = budget_bytes;
    sh->n_chunks = plan.n_chunks;
  }
  if (ctx) gt4hip_set_option (ctx, "pool_cap_mb", (int64_t) (budget_bytes / 3 >> 20));
  w->plan = &plan;
  w->budget_bytes = budget_bytes;
  w->n_mine = stopped (w) ? 0 : (plan.n_chunks - (unsigned int) rank + (unsigned int) G - 1) / (unsigned int) G;
  if (job->gather_rccl && G >= 1 && !stopped (w)) {
    if (gt4hip_comm_create (ctx, sh->comm_id, G, rank, &w->comm)) worker_fail (w, "Error: %s", gt4hip_last_error (ctx));
  }
  /* ---- output files: worker 0 creates "<name>.tmp" with the placeholder header, the others open them */
  const int writes = !job->prm.count_only;
  if (writes && rank == 0 && !stopped (w)) {
    for (int s = 0; s < 4; s++) {
      if (!job->out_name[s]) continue;
      char tmp[2048];
      snprintf (tmp, sizeof tmp, "%s.tmp", job->out_name[s]);
      GT4ListWriter lw;
      if (gt4_listwriter_begin (&lw, tmp, job->word_length, job->out_mode)) {
        worker_fail (w, "Error: Cannot create output file %.400s", tmp);
        break;
      }
      w->out_fd[s] = lw.fd;
    }
    /* (a descriptor of the caller's: single worker, nothing to create) */
    for (int s = 0; s < 4 && G == 1; s++)
      if (!job->out_name[s] && job->out_fd[s] > 0) w->out_fd[s] = job->out_fd[s];
  }
  if (G > 1) pthread_barrier_wait (&sh->bar);
  if (!stopped (w) && plan.n_chunks != sh->n_chunks) worker_fail (w, "Error: %s", "internal: chunk plans differ");
  if (writes && rank != 0 && !job->gather_rccl && !stopped (w)) {
    for (int s = 0; s < 4; s++) {
      if (!job->out_name[s]) continue;
      char tmp[2048];
      snprintf (tmp, sizeof tmp, "%s.tmp", job->out_name[s]);
      w->out_fd[s] = open (tmp, O_WRONLY);
      if (w->out_fd[s] < 0) worker_fail (w, "Error: Cannot open output file %.400s", tmp);
    }
  }

  /* ---- the pipeline */
  const double t_setup = now_s ();
  for (int slot = 0; slot < SLOTS; slot++) {
    sem_init (&w->slot_free[slot], 0, 1);
    sem_init (&w->slot_loaded[slot], 0, 0);
    sem_init (&w->slot_merged[slot], 0, 0);
    sem_init (&w->outs_released[slot], 0, 0);
  }
  pthread_t loader, writer;
  pthread_create (&loader, NULL, loader_main, w);
  pthread_create (&writer, NULL, writer_main, w);
  int outs_pending[SLOTS] = { 0, 0 };
  for (unsigned int i = 0; i < w->n_mine; i++) {
    const int slot = (int) (i % SLOTS);
    const unsigned int c = (unsigned int) rank + i * (unsigned int) G;
    sem_wait (&w->slot_loaded[slot]);
    if (outs_pending[slot]) {
      /* the outputs this slot produced two chunks ago have been written: give them back */
      sem_wait (&w->outs_released[slot]);
      for (int s = 0; s < 4; s++)
        if (w->out[slot][s]) gt4hip_list_free (w->out[slot][s]);
      outs_pending[slot] = 0;
    }
    const double t0 = now_s ();
    if (!stopped (w)) merge_chunk (w, ctx, slot, c);
    chunk_set_done (&sh->chunk[c]);
    if (job->gather_rccl && writes && !stopped (w)) gather_round (w, ctx, slot, c - (unsigned int) rank);
    w->t_merge += now_s () - t0;
    outs_pending[slot] = 1;
    sem_post (&w->slot_merged[slot]);
  }
  pthread_join (writer, NULL);
  pthread_join (loader, NULL);
  for (int slot = 0; slot < SLOTS; slot++)
    if (outs_pending[slot])
      for (int s = 0; s < 4; s++)
        if (w->out[slot][s]) gt4hip_list_free (w->out[slot][s]);
  const double t_pipe = now_s ();
  if (job->debug)
    fprintf (errf, "Worker %d: %u of %u chunks, load %.3f s, merge %.3f s, write %.3f s (threads overlap); set-up %.3f s, pipeline %.3f s\n", rank,
             w->n_mine, plan.n_chunks, w->t_load, w->t_merge, w->t_write, t_setup - t_begin, t_pipe - t_setup);
  if (w->comm) gt4hip_comm_destroy (w->comm);
  /* a worker that stopped early still marks its chunks so that nobody waits for them */
  if (stopped (w))
    for (unsigned int i = 0; i < w->n_mine; i++) chunk_set_done (&sh->chunk[(unsigned int) rank + i * (unsigned int) G]);
  for (int s = 0; s < 4; s++)
    if (w->out_fd[s] >= 0 && rank != 0) close (w->out_fd[s]);
  if (G > 1 && job->gather_rccl && stopped (w)) {
    /* the peers may be inside a collective that this worker will never join: leave at once (this is a
     * forked worker); the parent sees the exit code, ends the others and removes the temporaries */
    fflush (stdout);
    fflush (stderr);
    _exit (1);
  }
  if (G > 1) pthread_barrier_wait (&sh->bar); /* every extent is written */

  /* ---- worker 0: back-patch the headers (reference :907-915, :592-595) and rename into place */
  int rc = stopped (w) ? 1 : 0;
  if (rank == 0 && writes) {
    for (int s = 0; s < 4; s++) {
      if (!job->out_name[s] || w->out_fd[s] < 0) continue;
      char tmp[2048];
      snprintf (tmp, sizeof tmp, "%s.tmp", job->out_name[s]);
      if (rc) {
        close (w->out_fd[s]);
        unlink (tmp);
        continue;
      }
      uint64_t n = 0, t = 0;
      for (unsigned int c = 0; c < plan.n_chunks; c++) {
        n += sh->chunk[c].n[s];
        t += sh->chunk[c].t[s];
      }
      GT4ListWriter lw;
      gt4_list_header_init (&lw.header, job->word_length);
      lw.fd = w->out_fd[s];
      if (gt4_listwriter_finish (&lw, n, t)) {
        fprintf (errf, "Error: writing %s failed: %s\n", tmp, strerror (errno));
        unlink (tmp);
        rc = 1;
      } else if (rename (tmp, job->out_name[s])) {
        fprintf (errf, "Error: Cannot rename %s to %s\n", tmp, job->out_name[s]);
        rc = 1;
      }
    }
  }
  const double t_fin = now_s ();
  if (ctx) gt4hip_destroy (ctx);
  if (job->debug) fprintf (errf, "Worker %d: headers and renames %.3f s, context teardown %.3f s\n", rank, t_fin - t_pipe, now_s () - t_fin);
  free (plan.cut);
  free (w);
  if (errf != stderr) fflush (errf);
  return rc;
}

/* ------------------------------------------------------------------ entry */

int gt4_shard_run (const GT4ShardJob *job, GT4ShardResult *res)
{
  memset (res, 0, sizeof *res);
  if (job->n_ranks < 1 || job->n_ranks > MAX_RANKS || job->n_files > 1024 || job->n_files < 1) {
    fprintf (stderr, "Error: between 1 and %d GPUs can be used\n", MAX_RANKS);
    return 1;
  }
  /* the shared block: one record of totals per chunk, reserved for the largest plan the budget rule
   * can produce (address space only: untouched pages cost nothing) */
  const size_t max_chunks = (size_t) 1 << 22;
  const size_t bytes = sizeof (Shared) + max_chunks * sizeof (ChunkTotals);
  Shared *sh = (Shared *) mmap (NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (sh == MAP_FAILED) {
    fprintf (stderr, "Error: shared memory for the workers could not be mapped\n");
    return 1;
  }
  const int G = job->n_ranks;
  if (G > 1) {
    pthread_barrierattr_init (&sh->bar_attr);
    pthread_barrierattr_setpshared (&sh->bar_attr, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init (&sh->bar, &sh->bar_attr, (unsigned int) G);
  }
  int rc = 0;
  if (G == 1) {
    /* one worker, in this process (also with the RCCL gather of one rank: no fork, so it does not
     * matter whether the caller has touched HIP already) */
    rc = worker_main (job, sh, 0);
  } else {
    /* several GPUs: one worker process each, forked BEFORE anything in this process touches HIP; the
     * workers find their budgets and make the plan themselves */
    const GT4ShardJob jn = *job;
    fflush (stdout);
    fflush (stderr);
    pid_t pids[MAX_RANKS];
    for (int r = 0; r < G; r++) {
      pids[r] = fork ();
      if (pids[r] == 0) {
        const int wrc = worker_main (&jn, sh, r);
        fflush (stdout);
        fflush (stderr);
        _exit (wrc);
      }
      if (pids[r] < 0) {
        fprintf (stderr, "Error: fork failed: %s\n", strerror (errno));
        for (int q = 0; q < r; q++) kill (pids[q], SIGKILL);
        for (int q = 0; q < r; q++) waitpid (pids[q], NULL, 0);
        munmap (sh, bytes);
        return 1;
      }
    }
    /* a worker that dies (not merely fails) would leave the others at a barrier: end them */
    int left = G, killed = 0; /* killed: the remaining workers were ended by this process */
    while (left > 0) {
      int status = 0;
      const pid_t pid = waitpid (-1, &status, 0);
      if (pid < 0) {
        if (errno == EINTR) continue;
        break;
      }
      int r = -1;
      for (int q = 0; q < G; q++)
        if (pids[q] == pid) r = q;
      if (r < 0) continue;
      pids[r] = 0;
      left--;
      if (!WIFEXITED (status)) {
        if (!killed) fprintf (stderr, "Error: GPU worker %d ended abnormally (status 0x%x)\n", r, status);
        __atomic_store_n (&sh->failed, 1, __ATOMIC_RELEASE);
        rc = 1;
        killed = 1;
        for (int q = 0; q < G; q++)
          if (pids[q] > 0) kill (pids[q], SIGKILL);
      } else if (WEXITSTATUS (status)) {
        /* a worker that failed normally leaves through the barriers together with the others (they see
         * the flag and skip their remaining work): by the time the first one is reaped the others are
         * past every barrier or about to be.  One that left early -- out of memory before the first
         * barrier, or out of a collective its peers still sit in -- would leave them waiting for ever:
         * give them a moment, then end them. */
        rc = 1;
        __atomic_store_n (&sh->failed, 1, __ATOMIC_RELEASE);
        for (int wait_ms = 0; wait_ms < 2000 && left > 0; wait_ms += 20) {
          int st2 = 0;
          const pid_t p2 = waitpid (-1, &st2, WNOHANG);
          if (p2 > 0) {
            for (int q = 0; q < G; q++)
              if (pids[q] == p2) {
                pids[q] = 0;
                left--;
              }
          } else usleep (20000);
        }
        if (left > 0) {
          killed = 1;
          for (int q = 0; q < G; q++)
            if (pids[q] > 0) kill (pids[q], SIGKILL);
        }
      }
    }
    if (rc) {
      /* nothing half-written stays behind */
      for (int s = 0; s < 4; s++) {
        if (!job->out_name[s]) continue;
        char tmp[2048];
        snprintf (tmp, sizeof tmp, "%s.tmp", job->out_name[s]);
        unlink (tmp);
      }
    }
  }
  res->n_chunks = sh->n_chunks;
  res->rule_rejected = sh->rule_rejected;
  memcpy (res->message, sh->message, sizeof res->message);
  if (!rc)
    for (unsigned int c = 0; c < sh->n_chunks; c++)
      for (int s = 0; s < 4; s++) {
        res->n_words[s] += sh->chunk[c].n[s];
        res->total_count[s] += sh->chunk[c].t[s];
      }
  if (rc && sh->message[0] && !sh->rule_rejected) fprintf (stderr, "%s\n", sh->message);
  if (G > 1) pthread_barrier_destroy (&sh->bar);
  munmap (sh, bytes);
  return rc;
}
