/*
 * gt4hip_io.hip -- file <-> HBM transfers of list records.
 *
 * The reference reads its lists through mmap (gt4_mmap, src/utils.c:35-64, with a read-ahead
 * "scout" thread, :72-99) or 3 KiB read()s (word-list-stream.c:85-125), and writes results with two
 * fwrite calls per record (glistcompare.c:491-496).  Here a list body moves between a file
 * descriptor (or host memory) and HBM in 8 MiB pieces through pinned staging buffers that belong to
 * the context and are set up ONCE: every copy thread owns two buffers and one HIP stream, preads a
 * piece into one buffer while the other one is in flight to the device (or, for results, copies
 * device -> pinned and pwrites the piece at its file offset while the next one is in flight).
 * Pieces are dealt round-robin to the threads; nothing is shared between them but the job record.
 *
 * Host-only code (no kernels); compiled with hipcc like the rest of the C-ABI implementation.
 */
#include "gt4hip_host.h"

#include <errno.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr size_t PIECE_DEFAULT = 8u << 20; /* bytes per staging buffer (GT4HIP_IO_PIECE_MB) */
constexpr int MAX_THREADS = 64;

enum JobKind { JOB_NONE = 0, JOB_FD_TO_DEV, JOB_MEM_TO_DEV, JOB_DEV_TO_FD, JOB_DEV_TO_MEM, JOB_EXIT };

constexpr int MAX_SEG = 4;

struct Job {
  int kind;
  int fd;
  off_t file_off;
  const char *src_mem;
  char *dst_mem;
  char *dev;
  size_t bytes;
  /* JOB_DEV_TO_FD may carry several segments (the outputs of one operation, each to its own file):
   * the copy threads are dealt to the segments, so that different threads write different files --
   * writers of ONE file serialise on its inode lock */
  int n_seg;
  int seg_fd[MAX_SEG];
  off_t seg_off[MAX_SEG];
  char *seg_dev[MAX_SEG];
  size_t seg_bytes[MAX_SEG];
};

struct Worker {
  gt4hip_io *io;
  int index;
  pthread_t thread;
  void *pinned[2];
  hipStream_t stream;
  int err;           /* 0 ok, 1 read/write failed (sys_errno, file_off_failed say what and where), 2 HIP failed */
  int sys_errno;     /* errno of the failing pread / pwrite on THIS thread; -1: the file ended early */
  long long fail_off;
};

}  // namespace

struct gt4hip_io {
  int device;
  int n_threads;       /* copy threads that exist */
  int read_threads;    /* ... that file -> HBM jobs use: the page cache hands data out faster than it takes it in */
  size_t piece;        /* bytes per staging buffer */
  bool use_mmap;       /* results go into the files through shared mappings (GT4HIP_IO_MMAP=0: pwrite) */
  Worker w[MAX_THREADS];
  pthread_mutex_t mu;
  pthread_cond_t cv_start, cv_done;
  Job job;
  unsigned long generation; /* bumped per job */
  int running;              /* workers still busy with the current job */
};

namespace {

/* 0, or the errno of the failing call; -1: the file ended early (no errno at all) -- kept per copy
 * thread, because errno is thread-local and the message is composed by the calling thread */
int full_pread (int fd, void *buf, size_t len, off_t off)
{
  char *p = (char *) buf;
  while (len) {
    const ssize_t r = pread (fd, p, len, off);
    if (r < 0) {
      if (errno == EINTR) continue;
      return errno ? errno : EIO;
    }
    if (r == 0) return -1; /* file shorter than its header promised */
    p += r;
    off += r;
    len -= (size_t) r;
  }
  return 0;
}

int full_pwrite (int fd, const void *buf, size_t len, off_t off)
{
  const char *p = (const char *) buf;
  while (len) {
    const ssize_t r = pwrite (fd, p, len, off);
    if (r < 0) {
      if (errno == EINTR) continue;
      return errno ? errno : EIO;
    }
    p += r;
    off += r;
    len -= (size_t) r;
  }
  return 0;
}

/* One piece of a job: where it lies on the device and in the file (or in host memory). */
struct Piece {
  char *dev;
  int fd;
  off_t foff;
  size_t len;
  size_t mem_off;
};

/* piece g of the job, in segment order (a job without segments is one segment); false: past the end */
bool piece_at (const Job &j, size_t piece_bytes, size_t g, Piece *out)
{
  const int n = j.n_seg > 0 ? j.n_seg : 1;
  for (int s = 0; s < n; s++) {
    const size_t bytes = j.n_seg > 0 ? j.seg_bytes[s] : j.bytes;
    const size_t np = (bytes + piece_bytes - 1) / piece_bytes;
    if (g < np) {
      const size_t off = g * piece_bytes;
      out->dev = (j.n_seg > 0 ? j.seg_dev[s] : j.dev) + off;
      out->fd = j.n_seg > 0 ? j.seg_fd[s] : j.fd;
      out->foff = (j.n_seg > 0 ? j.seg_off[s] : j.file_off) + (off_t) off;
      out->len = bytes - off < piece_bytes ? bytes - off : piece_bytes;
      out->mem_off = off;
      return true;
    }
    g -= np;
  }
  return false;
}

/* Result bytes into a file THROUGH A SHARED MAPPING of the extent: pwrite takes the file's inode lock,
 * so the writers of one file wait for each other (the 36 GB union of the 2 x 2e9 pair went out at the
 * rate of ONE copying thread); page faults on a shared mapping do not.  The file is grown first (one
 * byte written at the extent's end: microseconds under the lock), never shrunk -- several threads and
 * processes fill disjoint extents of one file.  Falls back to pwrite where the file cannot be mapped. */
int write_extent (int fd, const void *buf, size_t len, off_t off, bool use_mmap)
{
  if (use_mmap && len) {
    const long pg = sysconf (_SC_PAGESIZE);
    const off_t base = off / pg * pg;
    const size_t delta = (size_t) (off - base);
    struct stat st;
    if (fstat (fd, &st) == 0 && S_ISREG (st.st_mode)) {
      if (st.st_size < off + (off_t) len) {
        const char z = 0;
        if (const int e = full_pwrite (fd, &z, 1, off + (off_t) len - 1)) return e;
      }
      void *m = mmap (NULL, len + delta, PROT_READ | PROT_WRITE, MAP_SHARED, fd, base);
      if (m != MAP_FAILED) {
        memcpy ((char *) m + delta, buf, len);
        munmap (m, len + delta);
        return 0;
      }
    }
  }
  return full_pwrite (fd, buf, len, off);
}

/* Which pieces this thread takes.  File -> HBM jobs and single files: pieces index, index + T, ...
 * HBM -> several files (the outputs of one operation): the threads are DEALT TO THE FILES in proportion
 * to their sizes (at least one each), and a file's threads take its pieces in turn -- writers of one
 * file wait for each other on its inode lock and on the page cache's per-file allocation lock, so
 * every thread serving every file (measured, 2 x 2e9 -u -i -d: 15 s against 6.5 s) only queues them up */
struct Share {
  size_t first, stride; /* global piece numbers first, first + stride, ... */
  size_t lo, hi;        /* ... inside [lo, hi): the pieces of this thread's file */
};

Share share_of (const Job &j, size_t piece_bytes, size_t index, size_t T)
{
  Share sh = { index, T, 0, (size_t) -1 };
  if (j.kind != JOB_DEV_TO_FD || j.n_seg < 2 || T < (size_t) j.n_seg) return sh;
  size_t np[MAX_SEG], total = 0, thr[MAX_SEG], used = 0;
  for (int s = 0; s < j.n_seg; s++) {
    np[s] = (j.seg_bytes[s] + piece_bytes - 1) / piece_bytes;
    total += np[s];
  }
  for (int s = 0; s < j.n_seg; s++) {
    thr[s] = total ? np[s] * T / total : 0;
    if (thr[s] < 1) thr[s] = 1;
    used += thr[s];
  }
  /* the rounding's leftovers (or excess) go to (come from) the largest file */
  int big = 0;
  for (int s = 1; s < j.n_seg; s++)
    if (np[s] > np[big]) big = s;
  while (used > T && thr[big] > 1) {
    thr[big]--;
    used--;
  }
  while (used < T) {
    thr[big]++;
    used++;
  }
  size_t t0 = 0, p0 = 0;
  for (int s = 0; s < j.n_seg; s++) {
    if (index < t0 + thr[s]) {
      sh.first = p0 + (index - t0);
      sh.stride = thr[s];
      sh.lo = p0;
      sh.hi = p0 + np[s];
      return sh;
    }
    t0 += thr[s];
    p0 += np[s];
  }
  sh.first = sh.hi = 0; /* (more threads than shares: nothing to do) */
  return sh;
}

void run_job (Worker *w, const Job &j)
{
  gt4hip_io *const io = w->io;
  const size_t PIECE = io->piece;
  const bool to_dev = j.kind == JOB_FD_TO_DEV || j.kind == JOB_MEM_TO_DEV;
  const size_t T = (size_t) (to_dev ? io->read_threads : io->n_threads);
  w->err = 0;
  if ((size_t) w->index >= T) return;
  const Share sh = share_of (j, PIECE, (size_t) w->index, T);
  Piece pc;
  if (to_dev) {
    int slot = 0;
    bool busy[2] = { false, false };
    hipEvent_t ev[2];
    hipEventCreateWithFlags (&ev[0], hipEventDisableTiming);
    hipEventCreateWithFlags (&ev[1], hipEventDisableTiming);
    for (size_t g = sh.first; g < sh.hi && piece_at (j, PIECE, g, &pc) && !w->err; g += sh.stride) {
      if (busy[slot] && hipEventSynchronize (ev[slot]) != hipSuccess) w->err = 2;
      if (j.kind == JOB_FD_TO_DEV) {
        if (const int e = full_pread (pc.fd, w->pinned[slot], pc.len, pc.foff)) {
          w->err = 1;
          w->sys_errno = e;
          w->fail_off = (long long) pc.foff;
        }
      } else {
        memcpy (w->pinned[slot], j.src_mem + pc.mem_off, pc.len);
      }
      if (!w->err && hipMemcpyAsync (pc.dev, w->pinned[slot], pc.len, hipMemcpyHostToDevice, w->stream) != hipSuccess) w->err = 2;
      hipEventRecord (ev[slot], w->stream);
      busy[slot] = true;
      slot ^= 1;
    }
    if (hipStreamSynchronize (w->stream) != hipSuccess) w->err = 2;
    hipEventDestroy (ev[0]);
    hipEventDestroy (ev[1]);
    return;
  }
  /* device -> pinned (async) -> file / memory: the copy of the next piece runs while this one is written */
  size_t g = sh.first;
  int slot = 0;
  Piece cur;
  bool have = g < sh.hi && piece_at (j, PIECE, g, &cur);
  if (have && hipMemcpyAsync (w->pinned[slot], cur.dev, cur.len, hipMemcpyDeviceToHost, w->stream) != hipSuccess) w->err = 2;
  while (have && !w->err) {
    if (hipStreamSynchronize (w->stream) != hipSuccess) w->err = 2;
    const int done_slot = slot;
    const Piece done = cur;
    g += sh.stride;
    have = g < sh.hi && piece_at (j, PIECE, g, &cur);
    if (have) {
      slot ^= 1;
      if (hipMemcpyAsync (w->pinned[slot], cur.dev, cur.len, hipMemcpyDeviceToHost, w->stream) != hipSuccess) w->err = 2;
    }
    if (!w->err) {
      if (j.kind == JOB_DEV_TO_FD) {
        if (const int e = write_extent (done.fd, w->pinned[done_slot], done.len, done.foff, io->use_mmap)) {
          w->err = 1;
          w->sys_errno = e;
          w->fail_off = (long long) done.foff;
        }
      } else {
        memcpy (j.dst_mem + done.mem_off, w->pinned[done_slot], done.len);
      }
    }
  }
  hipStreamSynchronize (w->stream);
}

void *worker_main (void *arg)
{
  Worker *w = (Worker *) arg;
  gt4hip_io *io = w->io;
  hipSetDevice (io->device);
  unsigned long seen = 0;
  for (;;) {
    pthread_mutex_lock (&io->mu);
    while (io->generation == seen) pthread_cond_wait (&io->cv_start, &io->mu);
    seen = io->generation;
    const Job j = io->job;
    pthread_mutex_unlock (&io->mu);
    if (j.kind == JOB_EXIT) return NULL;
    run_job (w, j);
    pthread_mutex_lock (&io->mu);
    if (--io->running == 0) pthread_cond_signal (&io->cv_done);
    pthread_mutex_unlock (&io->mu);
  }
}

int io_get (gt4hip_context *ctx, gt4hip_io **out)
{
  if (ctx->io) {
    *out = ctx->io;
    return GT4HIP_OK;
  }
  gt4hip_io *io = (gt4hip_io *) calloc (1, sizeof (gt4hip_io));
  if (!io) return gt4hip_fail (ctx, GT4HIP_ENOMEM, "host allocation failed");
  io->device = ctx->device;
  /* copy threads: GT4HIP_IO_THREADS in all (default 8: more writers only queue up in front of the page cache, measured), of which file -> HBM jobs use
   * GT4HIP_IO_READ_THREADS (default 8: eight already read a page-cached file at the PCIe rate) */
  int T = 8, TR = 8;
  const char *e = getenv ("GT4HIP_IO_THREADS");
  if (e && atoi (e) > 0) T = atoi (e);
  if (T > MAX_THREADS) T = MAX_THREADS;
  e = getenv ("GT4HIP_IO_READ_THREADS");
  if (e && atoi (e) > 0) TR = atoi (e);
  if (TR > T) TR = T;
  io->read_threads = TR;
  io->piece = PIECE_DEFAULT;
  e = getenv ("GT4HIP_IO_PIECE_MB");
  if (e && atoi (e) > 0 && atoi (e) <= 256) io->piece = (size_t) atoi (e) << 20;
  e = getenv ("GT4HIP_IO_MMAP");
  io->use_mmap = e && atoi (e); /* measured slower than pwrite once more than a few threads fault pages of one file in: off by default */
  pthread_mutex_init (&io->mu, NULL);
  pthread_cond_init (&io->cv_start, NULL);
  pthread_cond_init (&io->cv_done, NULL);
  int made = 0;
  bool bad = false;
  for (; made < T && !bad; made++) {
    Worker *w = &io->w[made];
    w->io = io;
    w->index = made;
    if (hipHostMalloc (&w->pinned[0], io->piece, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc (&w->pinned[1], io->piece, hipHostMallocDefault) != hipSuccess ||
        hipStreamCreateWithFlags (&w->stream, hipStreamNonBlocking) != hipSuccess)
      bad = true;
  }
  if (bad) {
    (void) hipGetLastError ();
    for (int i = 0; i < made; i++) {
      if (io->w[i].pinned[0]) hipHostFree (io->w[i].pinned[0]);
      if (io->w[i].pinned[1]) hipHostFree (io->w[i].pinned[1]);
      if (io->w[i].stream) hipStreamDestroy (io->w[i].stream);
    }
    free (io);
    return gt4hip_fail (ctx, GT4HIP_ENOMEM, "pinned staging buffers could not be allocated");
  }
  io->n_threads = T;
  for (int i = 0; i < T; i++) pthread_create (&io->w[i].thread, NULL, worker_main, &io->w[i]);
  ctx->io = io;
  *out = io;
  return GT4HIP_OK;
}

int io_run (gt4hip_context *ctx, const Job &job, const char *what)
{
  gt4hip_io *io = NULL;
  int rc = io_get (ctx, &io);
  if (rc) return rc;
  /* everything enqueued on the context's own stream (the kernels that produced / will consume the
   * records) is ordered against the copy streams by a full synchronisation on both sides */
  HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
  pthread_mutex_lock (&io->mu);
  io->job = job;
  io->running = io->n_threads;
  io->generation++;
  pthread_cond_broadcast (&io->cv_start);
  while (io->running) pthread_cond_wait (&io->cv_done, &io->mu);
  pthread_mutex_unlock (&io->mu);
  for (int i = 0; i < io->n_threads; i++) {
    if (io->w[i].err == 1)
      return gt4hip_fail (ctx, GT4HIP_EIO, "%s: file I/O failed near byte %lld: %s", what, io->w[i].fail_off,
                          io->w[i].sys_errno < 0 ? "the file is shorter than its header promises" : strerror (io->w[i].sys_errno));
    if (io->w[i].err == 2) {
      (void) hipGetLastError ();
      return gt4hip_fail (ctx, GT4HIP_EHIP, "%s: a HIP copy failed", what);
    }
  }
  return GT4HIP_OK;
}

}  // namespace

void gt4hip_io_destroy (gt4hip_context *ctx)
{
  gt4hip_io *io = ctx->io;
  if (!io) return;
  pthread_mutex_lock (&io->mu);
  io->job.kind = JOB_EXIT;
  io->generation++;
  pthread_cond_broadcast (&io->cv_start);
  pthread_mutex_unlock (&io->mu);
  for (int i = 0; i < io->n_threads; i++) {
    pthread_join (io->w[i].thread, NULL);
    hipHostFree (io->w[i].pinned[0]);
    hipHostFree (io->w[i].pinned[1]);
    hipStreamDestroy (io->w[i].stream);
  }
  pthread_mutex_destroy (&io->mu);
  pthread_cond_destroy (&io->cv_start);
  pthread_cond_destroy (&io->cv_done);
  free (io);
  ctx->io = NULL;
}

/* below this size a plain hipMemcpy is as fast and the staging threads are not worth waking */
static const size_t IO_THRESHOLD = 32u << 20;

extern "C" int gt4hip_list_load_fd (gt4hip_context *ctx, gt4hip_list *list, int fd, uint64_t file_offset, uint64_t n_words)
{
  if (!ctx || !list || n_words > list->capacity) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  list->n_words = n_words;
  if (!n_words) return GT4HIP_OK;
  Job j;
  memset (&j, 0, sizeof j);
  j.kind = JOB_FD_TO_DEV;
  j.fd = fd;
  j.file_off = (off_t) file_offset;
  j.dev = (char *) list->dev;
  j.bytes = (size_t) n_words * GT4HIP_RECORD_BYTES;
  return io_run (ctx, j, "gt4hip_list_load_fd");
}

extern "C" int gt4hip_list_upload_fd (gt4hip_context *ctx, int fd, uint64_t file_offset, uint64_t n_words, uint32_t word_length, gt4hip_list **out)
{
  if (!ctx || !out) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  gt4hip_list *l = NULL;
  int rc = gt4hip_list_new (ctx, n_words, word_length, &l);
  if (rc) return rc;
  rc = gt4hip_list_load_fd (ctx, l, fd, file_offset, n_words);
  if (rc) {
    gt4hip_list_free (l);
    return rc;
  }
  *out = l;
  return GT4HIP_OK;
}

extern "C" int gt4hip_list_load (gt4hip_context *ctx, gt4hip_list *list, const void *host_records, uint64_t n_words)
{
  if (!ctx || !list || n_words > list->capacity || (n_words && !host_records)) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  list->n_words = n_words;
  if (!n_words) return GT4HIP_OK;
  const size_t bytes = (size_t) n_words * GT4HIP_RECORD_BYTES;
  if (bytes < IO_THRESHOLD) {
    HIPCHK (ctx, hipMemcpyAsync (list->dev, host_records, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
    return GT4HIP_OK;
  }
  Job j;
  memset (&j, 0, sizeof j);
  j.kind = JOB_MEM_TO_DEV;
  j.src_mem = (const char *) host_records;
  j.dev = (char *) list->dev;
  j.bytes = bytes;
  return io_run (ctx, j, "gt4hip_list_load");
}

extern "C" int gt4hip_list_write_fd (gt4hip_context *ctx, const gt4hip_list *list, uint64_t first, uint64_t count, int fd, uint64_t file_offset)
{
  if (!ctx || !list || first > list->n_words || count > list->n_words - first) return GT4HIP_EINVAL;
  if (!count) return GT4HIP_OK;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  Job j;
  memset (&j, 0, sizeof j);
  j.kind = JOB_DEV_TO_FD;
  j.fd = fd;
  j.file_off = (off_t) file_offset;
  j.dev = (char *) list->dev + first * GT4HIP_RECORD_BYTES;
  j.bytes = (size_t) count * GT4HIP_RECORD_BYTES;
  return io_run (ctx, j, "gt4hip_list_write_fd");
}

extern "C" int gt4hip_lists_write_fd (gt4hip_context *ctx, uint32_t n, const gt4hip_list *const lists[], const uint64_t first[], const uint64_t count[],
                                      const int fds[], const uint64_t file_offsets[])
{
  if (!ctx || !n || n > MAX_SEG || !lists || !first || !count || !fds || !file_offsets) return GT4HIP_EINVAL;
  HIPCHK (ctx, hipSetDevice (ctx->device));
  Job j;
  memset (&j, 0, sizeof j);
  j.kind = JOB_DEV_TO_FD;
  for (uint32_t i = 0; i < n; i++) {
    if (!lists[i] || first[i] > lists[i]->n_words || count[i] > lists[i]->n_words - first[i]) return GT4HIP_EINVAL;
    if (!count[i]) continue;
    j.seg_fd[j.n_seg] = fds[i];
    j.seg_off[j.n_seg] = (off_t) file_offsets[i];
    j.seg_dev[j.n_seg] = (char *) lists[i]->dev + first[i] * GT4HIP_RECORD_BYTES;
    j.seg_bytes[j.n_seg] = (size_t) count[i] * GT4HIP_RECORD_BYTES;
    j.n_seg++;
  }
  if (!j.n_seg) return GT4HIP_OK;
  return io_run (ctx, j, "gt4hip_lists_write_fd");
}

/* used by gt4hip_list_download_range for large ranges */
int gt4hip_io_download (gt4hip_context *ctx, const void *dev, void *host, size_t bytes)
{
  Job j;
  memset (&j, 0, sizeof j);
  j.kind = JOB_DEV_TO_MEM;
  j.dst_mem = (char *) host;
  j.dev = (char *) dev;
  j.bytes = bytes;
  return io_run (ctx, j, "gt4hip_list_download");
}
