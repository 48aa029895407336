/* io_prefault_probe.c -- PROBE, not product: what bounds the output side of the file -> HBM -> file
 * pipeline -- the page cache taking NEW pages of one file -- and whether allocating them ahead of the writers helps.
 *   gcc -O2 -pthread tools/exp/io_prefault_probe.c -o tools/exp/io_prefault_probe
 *   tools/exp/io_prefault_probe <dir> <GiB per file> <writer threads>
 * Cases, each on fresh files in <dir>, GB/s over the bytes written:
 *   A  pwrite, T threads, one file                    (what the product does per output file)
 *   B  fallocate of the whole file, one call          (page allocation + zeroing alone)
 *   C  pwrite, T threads, into the file B allocated   (copy alone: pages are there)
 *   D  fallocate by T threads on disjoint ranges of one file
 *   E  B and then C overlapped: one thread allocates 256 MiB ahead of T writers (the proposal)
 *   F  A on three files at once (T threads each)
 *   G  MADV_POPULATE_WRITE on a shared mapping, one thread; then memcpy by T threads
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static const size_t PIECE = 8u << 20;
static char *src;

struct Job { int fd; size_t bytes; int T, idx; int mode; volatile size_t *ahead; char *map; };

static void *writer (void *a)
{
  struct Job *j = (struct Job *) a;
  const size_t np = j->bytes / PIECE;
  for (size_t g = (size_t) j->idx; g < np; g += (size_t) j->T) {
    const off_t off = (off_t) (g * PIECE);
    if (j->mode == 0) {
      if (j->ahead) while (*j->ahead < (size_t) off + PIECE) sched_yield ();
      size_t done = 0;
      while (done < PIECE) {
        ssize_t r = pwrite (j->fd, src + done, PIECE - done, off + (off_t) done);
        if (r < 0) { perror ("pwrite"); exit (1); }
        done += (size_t) r;
      }
    } else if (j->mode == 1) {
      if (fallocate (j->fd, 0, off, (off_t) PIECE)) { perror ("fallocate"); exit (1); }
    } else {
      if (j->mode == 3) while (*j->ahead < (size_t) off + PIECE) sched_yield ();
      memcpy (j->map + off, src, PIECE);
    }
  }
  return NULL;
}

static double run_threads (int fd, size_t bytes, int T, int mode, volatile size_t *ahead, char *map)
{
  pthread_t th[64];
  struct Job jobs[64];
  const double t0 = now ();
  for (int i = 0; i < T; i++) {
    jobs[i] = (struct Job) { fd, bytes, T, i, mode, ahead, map };
    pthread_create (&th[i], NULL, writer, &jobs[i]);
  }
  for (int i = 0; i < T; i++) pthread_join (th[i], NULL);
  return now () - t0;
}

struct Alloc { int fd; size_t bytes; volatile size_t ahead; };
static void *allocator (void *a)
{
  struct Alloc *al = (struct Alloc *) a;
  const size_t step = 256u << 20;
  for (size_t off = 0; off < al->bytes; off += step) {
    const size_t len = al->bytes - off < step ? al->bytes - off : step;
    if (fallocate (al->fd, 0, (off_t) off, (off_t) len)) { perror ("fallocate"); exit (1); }
    al->ahead = off + len;
  }
  return NULL;
}

static int fresh (const char *dir, const char *name, char *path)
{
  sprintf (path, "%s/%s", dir, name);
  unlink (path);
  int fd = open (path, O_RDWR | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) { perror (path); exit (1); }
  return fd;
}

int main (int argc, char **argv)
{
  const char *dir = argc > 1 ? argv[1] : "/dev/shm";
  const size_t gib = argc > 2 ? (size_t) atoi (argv[2]) : 8;
  const int T = argc > 3 ? atoi (argv[3]) : 8;
  const size_t bytes = gib << 30;
  char path[512], p2[512], p3[512];
  src = malloc (PIECE);
  memset (src, 0x5a, PIECE);
  printf ("dir %s, %zu GiB per file, %d threads\n", dir, gib, T);
  int fd = fresh (dir, "probe_a", path);
  double t = run_threads (fd, bytes, T, 0, NULL, NULL);
  printf ("A pwrite x%d, fresh file                 %6.2f GB/s\n", T, bytes / t / 1e9);
  close (fd); unlink (path);

  fd = fresh (dir, "probe_b", path);
  double t0 = now ();
  if (fallocate (fd, 0, 0, (off_t) bytes)) perror ("fallocate");
  t = now () - t0;
  printf ("B fallocate, one call                    %6.2f GB/s\n", bytes / t / 1e9);
  t = run_threads (fd, bytes, T, 0, NULL, NULL);
  printf ("C pwrite x%d into allocated pages        %6.2f GB/s\n", T, bytes / t / 1e9);
  close (fd); unlink (path);

  fd = fresh (dir, "probe_d", path);
  t = run_threads (fd, bytes, T, 1, NULL, NULL);
  printf ("D fallocate x%d on disjoint ranges       %6.2f GB/s\n", T, bytes / t / 1e9);
  close (fd); unlink (path);

  fd = fresh (dir, "probe_e", path);
  struct Alloc al = { fd, bytes, 0 };
  pthread_t at;
  t0 = now ();
  pthread_create (&at, NULL, allocator, &al);
  run_threads (fd, bytes, T, 0, &al.ahead, NULL);
  pthread_join (at, NULL);
  t = now () - t0;
  printf ("E allocator thread ahead of x%d writers  %6.2f GB/s\n", T, bytes / t / 1e9);
  close (fd); unlink (path);

  {
    int f1 = fresh (dir, "probe_f1", path), f2 = fresh (dir, "probe_f2", p2), f3 = fresh (dir, "probe_f3", p3);
    pthread_t th[3][64];
    struct Job jobs[3][64];
    int fds[3] = { f1, f2, f3 };
    t0 = now ();
    for (int f = 0; f < 3; f++)
      for (int i = 0; i < T; i++) {
        jobs[f][i] = (struct Job) { fds[f], bytes / 2, T, i, 0, NULL, NULL };
        pthread_create (&th[f][i], NULL, writer, &jobs[f][i]);
      }
    for (int f = 0; f < 3; f++)
      for (int i = 0; i < T; i++) pthread_join (th[f][i], NULL);
    t = now () - t0;
    printf ("F pwrite x%d on each of three files      %6.2f GB/s (all three together)\n", T, 3.0 * (bytes / 2) / t / 1e9);
    close (f1); close (f2); close (f3); unlink (path); unlink (p2); unlink (p3);
  }

  fd = fresh (dir, "probe_g", path);
  if (ftruncate (fd, (off_t) bytes)) perror ("ftruncate");
  char *m = mmap (NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (m != MAP_FAILED) {
    t0 = now ();
    int rc = madvise (m, bytes, MADV_POPULATE_WRITE);
    t = now () - t0;
    printf ("G MADV_POPULATE_WRITE, one call          %6.2f GB/s%s\n", bytes / t / 1e9, rc ? " (FAILED)" : "");
    t = run_threads (fd, bytes, T, 2, NULL, m);
    printf ("G memcpy x%d into the populated mapping  %6.2f GB/s\n", T, bytes / t / 1e9);
    munmap (m, bytes);
  }
  close (fd); unlink (path);

  /* H: fallocate, then memcpy through a shared mapping that is NOT populated (minor faults on pages that exist) */
  fd = fresh (dir, "probe_h", path);
  t0 = now ();
  if (fallocate (fd, 0, 0, (off_t) bytes)) perror ("fallocate");
  double ta = now () - t0;
  m = mmap (NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (m != MAP_FAILED) {
    t = run_threads (fd, bytes, T, 2, NULL, m);
    printf ("H fallocate %.2f GB/s, then memcpy x%d through an unpopulated mapping %6.2f GB/s\n", bytes / ta / 1e9, T, bytes / t / 1e9);
    munmap (m, bytes);
  }
  close (fd); unlink (path);

  /* I: allocator thread ahead (fallocate in 256 MiB steps), writers memcpy through the mapping behind it */
  fd = fresh (dir, "probe_i", path);
  if (ftruncate (fd, (off_t) bytes)) perror ("ftruncate");
  m = mmap (NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (m != MAP_FAILED) {
    struct Alloc al2 = { fd, bytes, 0 };
    t0 = now ();
    pthread_create (&at, NULL, allocator, &al2);
    {
      pthread_t th[64];
      struct Job jobs[64];
      for (int i = 0; i < T; i++) {
        jobs[i] = (struct Job) { fd, bytes, T, i, 3, &al2.ahead, m };
        pthread_create (&th[i], NULL, writer, &jobs[i]);
      }
      for (int i = 0; i < T; i++) pthread_join (th[i], NULL);
    }
    pthread_join (at, NULL);
    t = now () - t0;
    printf ("I allocator thread ahead, x%d writers memcpy through the mapping %6.2f GB/s\n", T, bytes / t / 1e9);
    munmap (m, bytes);
  }
  close (fd); unlink (path);
  return 0;
}
