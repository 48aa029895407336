/*
 * gt4_setops.c -- gt4_write_union / gt4_union / gt4_is_union (include/gt4_set_operations.h) on top
 * of the C ABI in include/gt4hip.h.  Host C; the merges run in the HIP kernels.  No CPU fallback:
 * without a device every entry point fails.
 */
#define _GNU_SOURCE
#include "gt4_set_operations.h"

#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

struct _GT4HipWordList {
  gt4hip_list *dev;  /* the list in HBM; NULL while a file-backed handle has not been uploaded */
  uint64_t num_words;
  uint64_t sum_counts;
  unsigned int word_length;
  uint64_t last_key; /* key of the last record (valid when num_words > 0) */
  /* File-backed handles (round 4): a list whose file is larger than the resident share of the device memory
   * stays mapped instead of being uploaded at once.  gt4_write_union streams such inputs through the device in
   * key-range chunks (gt4_shard.c: loader / merger / writer threads, the plan of the command-line tool); every
   * other entry point uploads the list on first use and fails with a message if it does not fit. */
  GT4ListFile file;
  int file_backed;   /* `file` is open (round 5: also for uploaded lists that came from a file, so that a union which does
                      * not fit beside them can still stream every input from its mapping) */
};

#include "gt4_shard.h"

/* device bytes a handle may take at once and stay resident: $GT4HIP_HBM_LIMIT (tests force the streaming path
 * with it), else a quarter of what the device has free now */
static uint64_t resident_limit (gt4hip_context *ctx)
{
  const char *e = getenv ("GT4HIP_HBM_LIMIT");
  if (e && *e) {
    char *end = NULL;
    double v = strtod (e, &end);
    if (end && (*end == 'K' || *end == 'k')) v *= 1024.0;
    else if (end && (*end == 'M' || *end == 'm')) v *= 1024.0 * 1024.0;
    else if (end && (*end == 'G' || *end == 'g')) v *= 1024.0 * 1024.0 * 1024.0;
    if (v > 0) return (uint64_t) v; /* ("0" or something unparsable: as if unset, like the command-line tool) */
  }
  uint64_t free_b = 0, total_b = 0;
  if (gt4hip_device_memory (ctx, &free_b, &total_b)) return UINT64_MAX; /* (no answer: upload as before rather than map everything) */
  return free_b / 4;
}

/* uploads a file-backed list that is not in HBM yet; 0 = the handle has a device list */
static int ensure_device (gt4hip_context *ctx, GT4HipWordList *l, const char *who)
{
  if (!l) return 1;
  if (l->dev) return 0;
  if (!l->file_backed) return 1;
  const GT4ListFile *lf = &l->file;
  const int rc = lf->index_kmers ? gt4hip_list_upload_index (ctx, lf->index_kmers, lf->header.n_words, lf->index_locations, lf->header.word_length, &l->dev)
                                 : gt4hip_list_upload (ctx, lf->records, lf->header.n_words, lf->header.word_length, &l->dev);
  if (rc) {
    fprintf (stderr, "%s: upload of %s failed: %s\n", who, lf->filename, gt4hip_last_error (ctx));
    l->dev = NULL;
    return 1;
  }
  return 0;
}

static gt4hip_context *g_ctx = NULL;

gt4hip_context *gt4_hip_default_context (void)
{
  if (!g_ctx) {
    const char *dev = getenv ("GT4HIP_DEVICE");
    int rc = gt4hip_create (dev ? atoi (dev) : 0, &g_ctx);
    if (rc) {
      fprintf (stderr, "Error: GPU set operations unavailable: %s\n", gt4hip_last_error (NULL));
      g_ctx = NULL;
    }
  }
  return g_ctx;
}

gt4hip_context *gt4_hip_set_default_context (gt4hip_context *ctx)
{
  gt4hip_context *const old = g_ctx;
  g_ctx = ctx;
  return old;
}

static GT4HipWordList *wrap_uploaded (gt4hip_context *ctx, gt4hip_list *dev, uint64_t n, unsigned int wl, uint64_t sum, int have_sum)
{
  GT4HipWordList *l = (GT4HipWordList *) calloc (1, sizeof *l);
  if (!l) {
    gt4hip_list_free (dev);
    return NULL;
  }
  l->dev = dev;
  l->num_words = n;
  l->word_length = wl;
  if (have_sum) l->sum_counts = sum;
  else if (gt4hip_list_sum_counts (ctx, dev, &l->sum_counts)) l->sum_counts = 0;
  if (n) {
    uint32_t c;
    if (gt4hip_list_get_word (ctx, dev, n - 1, &l->last_key, &c)) l->last_key = 0;
  }
  return l;
}

GT4HipWordList *gt4_hip_word_list_new (const char *listfilename, unsigned int major_version)
{
  gt4hip_context *ctx = gt4_hip_default_context ();
  if (!ctx) return NULL;
  GT4ListFile lf;
  uint32_t code = 0;
  /* a GT4I index is a sorted k-mer list too (gt4_index_map_new's GT4WordSList interface) */
  const int is_index = !gt4_listfile_sniff (listfilename, &code) && code == GT4_INDEX_CODE_VALUE;
  if (is_index ? gt4_indexfile_open (listfilename, major_version, &lf) : gt4_listfile_open (listfilename, major_version, &lf)) return NULL;
  if (12 * lf.header.n_words > resident_limit (ctx) && lf.header.n_words > 0) {
    /* too big to sit in HBM beside the others: stays mapped (see struct _GT4HipWordList) */
    GT4HipWordList *l = (GT4HipWordList *) calloc (1, sizeof *l);
    if (!l) {
      gt4_listfile_close (&lf);
      return NULL;
    }
    l->file = lf;
    l->file_backed = 1;
    l->num_words = lf.header.n_words;
    l->sum_counts = lf.header.total_count;
    l->word_length = lf.header.word_length;
    l->last_key = gt4_listfile_key_at (&lf, lf.header.n_words - 1);
    return l;
  }
  gt4hip_list *dev = NULL;
  int rc = is_index ? gt4hip_list_upload_index (ctx, lf.index_kmers, lf.header.n_words, lf.index_locations, lf.header.word_length, &dev)
                    : gt4hip_list_upload (ctx, lf.records, lf.header.n_words, lf.header.word_length, &dev);
  const uint64_t n = lf.header.n_words, total = lf.header.total_count;
  const unsigned int wl = lf.header.word_length;
  if (rc) {
    gt4_listfile_close (&lf);
    fprintf (stderr, "gt4_hip_word_list_new: upload of %s failed: %s\n", listfilename, gt4hip_last_error (ctx));
    return NULL;
  }
  GT4HipWordList *l = wrap_uploaded (ctx, dev, n, wl, total, 1);
  if (!l) {
    gt4_listfile_close (&lf);
    return NULL;
  }
  /* the mapping stays (costs address space only): a union too big to sit beside the uploaded lists streams every
   * input from its file (gt4_write_union) */
  l->file = lf;
  l->file_backed = 1;
  return l;
}

GT4HipWordList *gt4_hip_word_list_new_from_records (const void *records, uint64_t n_words, unsigned int word_length)
{
  gt4hip_context *ctx = gt4_hip_default_context ();
  if (!ctx) return NULL;
  gt4hip_list *dev = NULL;
  if (gt4hip_list_upload (ctx, records, n_words, word_length, &dev)) {
    fprintf (stderr, "gt4_hip_word_list_new_from_records: %s\n", gt4hip_last_error (ctx));
    return NULL;
  }
  return wrap_uploaded (ctx, dev, n_words, word_length, 0, 0);
}

void gt4_hip_word_list_delete (GT4HipWordList *list)
{
  if (!list) return;
  if (list->dev) gt4hip_list_free (list->dev);
  if (list->file_backed) gt4_listfile_close (&list->file);
  free (list);
}

uint64_t gt4_hip_word_list_num_words (const GT4HipWordList *list) { return list ? list->num_words : 0; }
uint64_t gt4_hip_word_list_sum_counts (const GT4HipWordList *list) { return list ? list->sum_counts : 0; }
unsigned int gt4_hip_word_list_word_length (const GT4HipWordList *list) { return list ? list->word_length : 0; }
const gt4hip_list *gt4_hip_word_list_device (const GT4HipWordList *list) { return list ? list->dev : NULL; } /* (NULL while a file-backed list is not uploaded) */
int gt4_hip_word_list_is_file_backed (const GT4HipWordList *list) { return list && list->file_backed && !list->dev; }

/* ------------------------------------------------------------------ the one-list iterator (src/word-list-sorted.c:59-78) */

#define ITER_BLOCK (1u << 20)

static void iter_remember (GT4HipWordSListIter *it, void *block);

/* makes record it->idx the current one; 1 = ok */
static unsigned int iter_load (GT4HipWordSListIter *it)
{
  GT4HipWordList *l = it->list;
  const unsigned char *rec = NULL;
  if (!l->dev && l->file_backed && l->file.records) {
    rec = l->file.records + 12 * it->idx; /* the mapping itself (src/word-map.h:89-99) */
  } else {
    if (it->idx < it->block_first || it->idx >= it->block_first + it->block_count || !it->block) {
      gt4hip_context *ctx = gt4_hip_default_context ();
      if (!ctx || ensure_device (ctx, l, "gt4_hip_word_slist")) return 0;
      if (!it->block) {
        it->block = malloc ((size_t) ITER_BLOCK * 12u);
        if (it->block) iter_remember (it, it->block);
      }
      if (!it->block) return 0;
      const uint64_t cnt = l->num_words - it->idx < ITER_BLOCK ? l->num_words - it->idx : ITER_BLOCK;
      if (gt4hip_list_download_range (ctx, l->dev, it->idx, cnt, it->block)) {
        fprintf (stderr, "gt4_hip_word_slist: %s\n", gt4hip_last_error (ctx));
        return 0;
      }
      it->block_first = it->idx;
      it->block_count = cnt;
    }
    rec = (const unsigned char *) it->block + 12 * (it->idx - it->block_first);
  }
  memcpy (&it->word, rec, 8);
  memcpy (&it->count, rec + 8, 4);
  return 1;
}

/* Blocks of iterators that are walked again without a release in between (the reference's idiom: get_first_word on
 * the same iterator restarts the walk, src/word-list-sorted.c:59-68, and there is no release there).  The iterator
 * itself may be uninitialised stack memory on its first use, so what it holds cannot be looked at: the blocks handed
 * out are remembered by the iterator's ADDRESS instead. */
#define ITER_SLOTS 64
static struct { GT4HipWordSListIter *it; void *block; } g_iter_blocks[ITER_SLOTS];
static pthread_mutex_t g_iter_lock = PTHREAD_MUTEX_INITIALIZER; /* (iterators on different lists may be walked by different threads) */

/* A walk restarted at this ADDRESS: the block of the walk before goes (freed here). */
static void iter_forget_address (GT4HipWordSListIter *it)
{
  pthread_mutex_lock (&g_iter_lock);
  for (int i = 0; i < ITER_SLOTS; i++)
    if (g_iter_blocks[i].it == it) {
      free (g_iter_blocks[i].block);
      g_iter_blocks[i].it = NULL;
      g_iter_blocks[i].block = NULL;
    }
  pthread_mutex_unlock (&g_iter_lock);
}

/* A block released by its owner -- who may be a COPY of the iterator it was handed to: entries go by the block, not by
 * the address, so that the original address does not free it a second time on its next walk. */
static void iter_forget_block (void *block)
{
  if (!block) return;
  pthread_mutex_lock (&g_iter_lock);
  for (int i = 0; i < ITER_SLOTS; i++)
    if (g_iter_blocks[i].block == block) {
      g_iter_blocks[i].it = NULL;
      g_iter_blocks[i].block = NULL;
    }
  pthread_mutex_unlock (&g_iter_lock);
}

static void iter_remember (GT4HipWordSListIter *it, void *block)
{
  pthread_mutex_lock (&g_iter_lock);
  for (int i = 0; i < ITER_SLOTS; i++)
    if (!g_iter_blocks[i].it) {
      g_iter_blocks[i].it = it;
      g_iter_blocks[i].block = block;
      break;
    }
  /* (more than ITER_SLOTS live iterators with blocks: those beyond must be released by their owners) */
  pthread_mutex_unlock (&g_iter_lock);
}

unsigned int gt4_hip_word_slist_get_first_word (GT4HipWordList *list, GT4HipWordSListIter *it)
{
  if (!list || !it) return 0;
  iter_forget_address (it); /* a walk restarted on this iterator: its block of the walk before goes */
  memset (it, 0, sizeof *it);
  it->list = list;
  it->num_words = list->num_words;
  it->sum_counts = list->sum_counts;
  it->word_length = list->word_length;
  it->idx = 0; /* :64 */
  if (!it->num_words) return 0; /* :65 */
  return iter_load (it);
}

unsigned int gt4_hip_word_slist_get_next_word (GT4HipWordSListIter *it)
{
  if (!it || !it->list) return 0;
  if (it->idx >= it->num_words) return 0; /* :73 */
  it->idx += 1;                           /* :74 */
  if (it->idx >= it->num_words) return 0; /* :75: word and count stay those of the last record */
  return iter_load (it);
}

void gt4_hip_word_slist_iter_release (GT4HipWordSListIter *it)
{
  if (!it) return;
  iter_forget_block (it->block);
  free (it->block);
  it->block = NULL;
  it->block_count = 0;
}

static int write_fully (int fd, const void *buf, size_t len)
{
  const char *p = (const char *) buf;
  while (len) {
    ssize_t w = write (fd, p, len);
    if (w < 0) {
      if (errno == EINTR) continue;
      return 1;
    }
    p += w;
    len -= (size_t) w;
  }
  return 0;
}

#define DOWNLOAD_CHUNK (4u << 20) /* records per device -> host -> file step (48 MiB) */

unsigned int gt4_write_union (GT4HipWordList *arrays[], unsigned int n_arrays, unsigned int cutoff, int ofile, GT4ListHeader *header)
{
  /* src/set-operations.c:49-50 */
  if (n_arrays == 0 || n_arrays > GT4_MAX_SETS || !arrays || !header) return 1;
  gt4hip_context *ctx = gt4_hip_default_context ();
  if (!ctx) return 1;
  /* Inputs that were too big to upload (file-backed handles): when ALL of them are, the union is streamed through
   * the device in key-range chunks -- the command-line tool's pipeline (gt4_shard.c), reading the mapped files and
   * writing `ofile` behind its header -- so neither the inputs nor the output have to fit HBM; glistmaker's
   * collation of up to 32 temporary lists (reference src/glistmaker.c:787-835) is this call. */
  /* Round 5 (ADVICE): decided per CALL.  Every handle that came from a file keeps its mapping, uploaded or not; when
   * all inputs have one, the union streams whenever some input is not resident, or the inputs still to upload plus the
   * worst-case output (plus the intermediate levels of more than eight lists) would not fit what the device has free --
   * and again as the fallback when the resident union runs out of memory after all. */
  unsigned int n_backed = 0, n_files = 0;
  uint64_t in_bytes = 0, missing_bytes = 0;
  for (unsigned int j = 0; j < n_arrays; j++) {
    if (!arrays[j]) return 1;
    n_files += arrays[j]->file_backed != 0;
    n_backed += arrays[j]->file_backed && !arrays[j]->dev;
    in_bytes += 12u * arrays[j]->num_words;
    if (!arrays[j]->dev) missing_bytes += 12u * arrays[j]->num_words;
  }
  int stream = n_backed == n_arrays;
  if (!stream && n_files == n_arrays) {
    uint64_t free_b = 0, total_b = 0;
    const uint64_t need = missing_bytes + in_bytes + (n_arrays > 8 ? in_bytes : 0);
    if (n_backed > 0) stream = 1; /* (a mixed set: the lists that did not fit alone will not fit beside the others) */
    else if (!gt4hip_device_memory (ctx, &free_b, &total_b) && need > free_b - free_b / 5) stream = 1;
  }
  int tried_resident = 0;
stream_it:
  if (stream && n_files == n_arrays && n_arrays <= 1024) {
    GT4ListFile *files = (GT4ListFile *) malloc (n_arrays * sizeof *files);
    if (!files) return 1;
    for (unsigned int j = 0; j < n_arrays; j++) files[j] = arrays[j]->file;
    gt4_list_header_init (header, arrays[0]->word_length);
    GT4ShardJob job;
    GT4ShardResult sres;
    memset (&job, 0, sizeof job);
    job.n_files = n_arrays;
    job.files = files;
    job.word_length = arrays[0]->word_length;
    job.mode = GT4_SHARD_UNION_MULTI;
    job.prm.rule = GT4HIP_RULE_ADD;
    job.prm.cutoff = cutoff;
    job.prm.count_only = ofile ? 0 : 1;
    job.n_ranks = 1;
    /* the worker creates its own contexts: on the device the budget is measured on, with the pooled blocks of this
     * context given back first (they would count as used) */
    gt4hip_trim (ctx);
    job.device_plus_1 = gt4hip_context_device (ctx) + 1;
    job.hbm_limit = resident_limit (ctx) == UINT64_MAX ? 0 : resident_limit (ctx);
    job.debug = getenv ("GT4HIP_VERBOSE") && atoi (getenv ("GT4HIP_VERBOSE"));
    unsigned int bad = 0;
    off_t pos = 0;
    if (ofile) {
      /* placeholder header (:65), records behind it, the real header over it at the end (:123-126) */
      pos = lseek (ofile, 0, SEEK_CUR);
      if (pos < 0) {
        fprintf (stderr, "gt4_write_union: inputs of this size need a seekable output file\n");
        free (files);
        return 1;
      }
      bad |= write_fully (ofile, header, sizeof *header);
      job.out_fd[0] = ofile;
      job.out_base[0] = (uint64_t) pos + sizeof *header;
    }
    if (!bad && gt4_shard_run (&job, &sres)) bad = 1;
    free (files);
    if (bad) return 1;
    header->n_words = sres.n_words[0];
    header->total_count = sres.total_count[0];
    if (ofile) {
      if (pwrite (ofile, header, sizeof *header, pos) != (ssize_t) sizeof *header) return 1;
      if (lseek (ofile, pos + (off_t) sizeof *header + (off_t) (header->n_words * 12u), SEEK_SET) < 0) return 1;
    }
    return 0;
  }
  for (unsigned int j = 0; j < n_arrays; j++)
    if (ensure_device (ctx, arrays[j], "gt4_write_union")) return 1;
  const gt4hip_list **devs = (const gt4hip_list **) malloc (n_arrays * sizeof *devs);
  if (!devs) return 1;
  for (unsigned int j = 0; j < n_arrays; j++) devs[j] = arrays[j]->dev;
  /* the reference takes the word length of the last non-empty list it looked at, or of the last
   * list when all are empty (:54-63); identical for well-formed input where all lengths agree */
  gt4_list_header_init (header, arrays[0]->word_length);
  gt4hip_multi_result res;
  memset (&res, 0, sizeof res);
  int rc = gt4hip_union_multi (ctx, devs, n_arrays, cutoff, GT4HIP_RULE_ADD, 0, ofile ? 0 : 1, &res);
  free (devs);
  if (rc == GT4HIP_ENOMEM && n_files == n_arrays && n_arrays <= 1024 && !tried_resident) {
    tried_resident = 1;
    stream = 1;
    goto stream_it;
  }
  if (rc) {
    fprintf (stderr, "gt4_write_union: %s\n", gt4hip_last_error (ctx));
    return 1;
  }
  header->n_words = res.n_words;
  header->total_count = res.total_count;
  unsigned int bad = 0;
  if (ofile) {
    /* header, then the records (:65, :104-121).  A seekable file takes the library's copy threads
     * (pinned staging, several writers of disjoint extents: gt4hip_list_write_fd) from the current
     * position and is left positioned behind the last record, as sequential writes would leave it;
     * a pipe or socket gets the records streamed device -> host -> fd */
    bad |= write_fully (ofile, header, sizeof *header);
    const uint64_t n = res.n_words;
    const off_t pos = bad ? (off_t) -1 : lseek (ofile, 0, SEEK_CUR);
    if (n && !bad && pos >= 0) {
      if (gt4hip_list_write_fd (ctx, res.out, 0, n, ofile, (uint64_t) pos)) {
        fprintf (stderr, "gt4_write_union: %s\n", gt4hip_last_error (ctx));
        bad = 1;
      } else if (lseek (ofile, pos + (off_t) (n * 12u), SEEK_SET) < 0) bad = 1;
    } else if (n && !bad) {
      void *buf = malloc ((size_t) (n < DOWNLOAD_CHUNK ? n : DOWNLOAD_CHUNK) * 12u);
      if (!buf) bad = 1;
      for (uint64_t first = 0; first < n && !bad; first += DOWNLOAD_CHUNK) {
        const uint64_t cnt = n - first < DOWNLOAD_CHUNK ? n - first : DOWNLOAD_CHUNK;
        if (gt4hip_list_download_range (ctx, res.out, first, cnt, buf)) bad = 1;
        else bad |= write_fully (ofile, buf, (size_t) cnt * 12u);
      }
      free (buf);
    }
  }
  gt4hip_list_free (res.out);
  return bad;
}

#define TABLE_CHUNK (1u << 20) /* table rows per download */
static uint64_t WALK_BLOCK = 1ull << 24;  /* records of the pacing list per key-range block of a walk ...              */
static uint64_t WALK_SINGLE = 1ull << 26; /* ... of lists with more records than this in all (tests: GT4HIP_WALK_BLOCK) */

static unsigned int walk_table (gt4hip_context *ctx, gt4hip_count_table *t, GT4HipWordList *objs[], unsigned int n_objs, int quirk,
                                unsigned int (*callback) (uint64_t, uint32_t *, void *), void *data)
{
  const uint64_t n = t->n_keys;
  unsigned int result = 0;
  if (!n) return 0;
  const uint64_t rows = n < TABLE_CHUNK ? n : TABLE_CHUNK;
  uint64_t *keys = (uint64_t *) malloc ((size_t) rows * 8);
  uint32_t *counts = (uint32_t *) malloc ((size_t) rows * n_objs * 4);
  uint32_t *zeros = (uint32_t *) calloc (n_objs, 4);
  if (!keys || !counts || !zeros) result = 1;
  uint64_t last_union_key = 0;
  if (!result && quirk) {
    /* the union's largest key: after it nothing remains, so no repeated visit there */
    for (unsigned int j = 0; j < n_objs; j++)
      if (objs[j]->num_words && objs[j]->last_key > last_union_key) last_union_key = objs[j]->last_key;
  }
  for (uint64_t first = 0; first < n && !result; first += TABLE_CHUNK) {
    const uint64_t cnt = n - first < TABLE_CHUNK ? n - first : TABLE_CHUNK;
    if (gt4hip_table_download (ctx, t, first, cnt, keys, counts)) {
      fprintf (stderr, "gt4_union: %s\n", gt4hip_last_error (ctx));
      result = 1;
      break;
    }
    for (uint64_t i = 0; i < cnt && !result; i++) {
      result = callback (keys[i], counts + i * n_objs, data);
      if (result || !quirk || keys[i] == last_union_key) continue;
      /* src/set-operations.c:166-170: a list that has just run out still feeds its last word into
       * `next`, so the reference visits that word a second time, with all counts 0 */
      int exhausted_here = 0;
      for (unsigned int j = 0; j < n_objs; j++) exhausted_here |= objs[j]->num_words && objs[j]->last_key == keys[i];
      if (exhausted_here) {
        memset (zeros, 0, (size_t) n_objs * 4);
        result = callback (keys[i], zeros, data);
      }
    }
  }
  free (keys);
  free (counts);
  free (zeros);
  return result;
}

static unsigned int table_walk (GT4HipWordList *objs[], unsigned int n_objs, int probe,
                                unsigned int (*callback) (uint64_t, uint32_t *, void *), void *data)
{
  if (n_objs == 0 || n_objs > GT4_MAX_SETS || !objs || !callback) return 1; /* src/set-operations.c:140-141 */
  gt4hip_context *ctx = gt4_hip_default_context ();
  if (!ctx) return 1;
  for (unsigned int j = 0; j < n_objs; j++)
    if (ensure_device (ctx, objs[j], probe ? "gt4_is_union" : "gt4_union")) return 1;
  {
    const char *e = getenv ("GT4HIP_WALK_BLOCK"); /* tests: small blocks on small lists */
    if (e && atoll (e) > 0) {
      WALK_BLOCK = (uint64_t) atoll (e);
      WALK_SINGLE = 0;
    }
  }
  const gt4hip_list **devs = (const gt4hip_list **) malloc (n_objs * sizeof *devs);
  if (!devs) return 1;
  for (unsigned int j = 0; j < n_objs; j++) devs[j] = objs[j]->dev;
  /* Long lists are walked in KEY-RANGE BLOCKS (the operations are key-local): the table of a block is built, handed
   * to the callback row by row, freed, then the next block's -- the table never holds more than a block, and a
   * callback that stops the walk (reference src/set-operations.c:176-178) stops the device work with it instead of
   * paying for the whole union first.  Block boundaries: every WALK_BLOCK-th key of the list that paces the walk
   * (gt4_is_union: list 0, whose keys are the rows; gt4_union: the longest), located in the other lists by lower
   * bounds on the device. */
  uint64_t total = 0;
  unsigned int pace = 0;
  for (unsigned int j = 0; j < n_objs; j++) {
    total += objs[j]->num_words;
    if (!probe && objs[j]->num_words > objs[pace]->num_words) pace = j;
  }
  const char *const who = probe ? "gt4_is_union" : "gt4_union";
  const uint64_t n_pace = objs[pace]->num_words;
  const uint64_t n_blocks = total > WALK_SINGLE && n_pace > WALK_BLOCK ? (n_pace + WALK_BLOCK - 1) / WALK_BLOCK : 1;
  unsigned int result = 0;
  int rc = 0;
  if (n_blocks == 1) {
    gt4hip_count_table t;
    rc = probe ? gt4hip_probe_table (ctx, devs, n_objs, &t) : gt4hip_union_table (ctx, devs, n_objs, &t);
    if (!rc) {
      result = walk_table (ctx, &t, objs, n_objs, !probe, callback, data);
      gt4hip_table_free (&t);
    }
  } else {
    uint64_t *lo = (uint64_t *) calloc (n_objs, sizeof *lo), *hi = (uint64_t *) calloc (n_objs, sizeof *hi);
    gt4hip_list **views = (gt4hip_list **) calloc (n_objs, sizeof *views);
    if (!lo || !hi || !views) rc = 1;
    for (uint64_t b = 0; b < n_blocks && !rc && !result; b++) {
      /* [lo, hi) of every list: records with a key below the pacing list's key at (b + 1) * WALK_BLOCK */
      if (b + 1 == n_blocks) {
        for (unsigned int j = 0; j < n_objs; j++) hi[j] = objs[j]->num_words;
      } else {
        uint64_t cut_key = 0;
        uint32_t c = 0;
        rc = gt4hip_list_get_word (ctx, devs[pace], (b + 1) * WALK_BLOCK, &cut_key, &c);
        for (unsigned int j = 0; j < n_objs && !rc; j++) {
          if (j == pace) hi[j] = (b + 1) * WALK_BLOCK;
          else rc = gt4hip_list_lower_bound (ctx, devs[j], cut_key, &hi[j]);
        }
      }
      for (unsigned int j = 0; j < n_objs && !rc; j++) rc = gt4hip_list_slice (ctx, devs[j], lo[j], hi[j] - lo[j], &views[j]);
      if (!rc) {
        gt4hip_count_table t;
        rc = probe ? gt4hip_probe_table (ctx, (const gt4hip_list *const *) views, n_objs, &t) : gt4hip_union_table (ctx, (const gt4hip_list *const *) views, n_objs, &t);
        if (!rc) {
          result = walk_table (ctx, &t, objs, n_objs, !probe, callback, data);
          gt4hip_table_free (&t);
        }
      }
      for (unsigned int j = 0; j < n_objs; j++) {
        if (views && views[j]) gt4hip_list_free (views[j]);
        if (views) views[j] = NULL;
        lo[j] = hi[j];
      }
    }
    free (lo);
    free (hi);
    free (views);
  }
  free (devs);
  if (rc) {
    fprintf (stderr, "%s: %s\n", who, gt4hip_last_error (ctx));
    return 1;
  }
  return result;
}

unsigned int gt4_union (GT4HipWordList *objs[], unsigned int n_objs, unsigned int (*callback) (uint64_t, uint32_t *, void *), void *data)
{
  return table_walk (objs, n_objs, 0, callback, data);
}

unsigned int gt4_is_union (GT4HipWordList *objs[], unsigned int n_objs, unsigned int (*callback) (uint64_t, uint32_t *, void *), void *data)
{
  return table_walk (objs, n_objs, 1, callback, data);
}

/* ------------------------------------------------------------------ glistquery's searches */

unsigned int gt4_word2string (char *b, uint64_t word, unsigned int wordlength)
{
  static const char alphabet[4] = { 'A', 'C', 'G', 'T' };
  for (unsigned int i = 0; i < wordlength; i++) {
    b[wordlength - i - 1] = alphabet[word & 3];
    word >>= 2;
  }
  b[wordlength] = 0;
  return wordlength;
}

unsigned int gt4_search_lists_multi (GT4HipWordList *query, GT4HipWordList *lists[], unsigned int n_lists,
                                     unsigned int (*callback) (uint64_t, unsigned int, uint32_t, void *), void *data)
{
  if (!query || !lists || !n_lists || n_lists >= GT4_MAX_SETS || !callback) return 1;
  gt4hip_context *ctx = gt4_hip_default_context ();
  if (!ctx) return 1;
  const unsigned int n = n_lists + 1;
  if (ensure_device (ctx, query, "gt4_search_lists_multi")) return 1;
  for (unsigned int j = 0; j < n_lists; j++)
    if (ensure_device (ctx, lists[j], "gt4_search_lists_multi")) return 1;
  const gt4hip_list **devs = (const gt4hip_list **) malloc (n * sizeof *devs);
  if (!devs) return 1;
  devs[0] = query->dev;
  for (unsigned int j = 0; j < n_lists; j++) devs[j + 1] = lists[j]->dev;
  /* two tables over the query's words: the lists' counts, and which lists hold the word at all */
  gt4hip_count_table tc, tp;
  int rc = gt4hip_probe_table_ex (ctx, devs, n, 0, &tc);
  if (!rc) {
    rc = gt4hip_probe_table_ex (ctx, devs, n, 1, &tp);
    if (rc) gt4hip_table_free (&tc);
  }
  free (devs);
  if (rc) {
    fprintf (stderr, "gt4_search_lists_multi: %s\n", gt4hip_last_error (ctx));
    return 1;
  }
  unsigned int result = 0;
  const uint64_t total = tc.n_keys, rows = total < TABLE_CHUNK ? total : TABLE_CHUNK;
  uint64_t *keys = (uint64_t *) malloc ((size_t) (rows ? rows : 1) * 8);
  uint32_t *counts = (uint32_t *) malloc ((size_t) (rows ? rows : 1) * n * 4), *present = (uint32_t *) malloc ((size_t) (rows ? rows : 1) * n * 4);
  if (!keys || !counts || !present) result = 1;
  for (uint64_t first = 0; first < total && !result; first += TABLE_CHUNK) {
    const uint64_t cnt = total - first < TABLE_CHUNK ? total - first : TABLE_CHUNK;
    if (gt4hip_table_download (ctx, &tc, first, cnt, keys, counts) || gt4hip_table_download (ctx, &tp, first, cnt, NULL, present)) {
      fprintf (stderr, "gt4_search_lists_multi: %s\n", gt4hip_last_error (ctx));
      result = 1;
      break;
    }
    for (uint64_t i = 0; i < cnt && !result; i++)
      for (unsigned int j = 1; j < n && !result; j++)
        if (present[i * n + j]) result = callback (keys[i], j - 1, counts[i * n + j], data);
  }
  free (keys);
  free (counts);
  free (present);
  gt4hip_table_free (&tc);
  gt4hip_table_free (&tp);
  return result;
}

unsigned int gt4_search_list_zipper (GT4HipWordList *list, GT4HipWordList *query, unsigned int (*callback) (uint64_t, uint32_t, void *), void *data)
{
  if (!list || !query || !callback) return 1;
  gt4hip_context *ctx = gt4_hip_default_context ();
  if (!ctx) return 1;
  /* the words of `query` that `list` holds, with the query's counts: an intersection under rule
   * FIRST with every matched key kept (cutoff 0 lets zero counts through the input test; the
   * reference's loop has no count test at all) -- except keys whose count in `query` is 0, which
   * the reference prints and the intersection's "count != 0" test drops: those come from the
   * presence table instead, so use the table form when the query holds zero counts */
  if (ensure_device (ctx, query, "gt4_search_list_zipper") || ensure_device (ctx, list, "gt4_search_list_zipper")) return 1;
  const gt4hip_list *devs[2] = { query->dev, list->dev };
  gt4hip_count_table tp;
  if (gt4hip_probe_table_ex (ctx, devs, 2, 1, &tp)) {
    fprintf (stderr, "gt4_search_list_zipper: %s\n", gt4hip_last_error (ctx));
    return 1;
  }
  unsigned int result = 0;
  const uint64_t total = tp.n_keys, rows = total < TABLE_CHUNK ? total : TABLE_CHUNK;
  uint64_t *keys = (uint64_t *) malloc ((size_t) (rows ? rows : 1) * 8);
  uint32_t *present = (uint32_t *) malloc ((size_t) (rows ? rows : 1) * 2 * 4);
  unsigned char *recs = (unsigned char *) malloc ((size_t) (rows ? rows : 1) * 12);
  if (!keys || !present || !recs) result = 1;
  for (uint64_t first = 0; first < total && !result; first += TABLE_CHUNK) {
    const uint64_t cnt = total - first < TABLE_CHUNK ? total - first : TABLE_CHUNK;
    if (gt4hip_table_download (ctx, &tp, first, cnt, keys, present) || gt4hip_list_download_range (ctx, query->dev, first, cnt, recs)) {
      fprintf (stderr, "gt4_search_list_zipper: %s\n", gt4hip_last_error (ctx));
      result = 1;
      break;
    }
    for (uint64_t i = 0; i < cnt && !result; i++) {
      if (!present[2 * i + 1]) continue;
      uint32_t c;
      memcpy (&c, recs + 12 * i + 8, 4);
      result = callback (keys[i], c, data);
    }
  }
  free (keys);
  free (present);
  free (recs);
  gt4hip_table_free (&tp);
  return result;
}
