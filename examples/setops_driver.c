/*
 * setops_driver.c -- example / test host program for include/gt4_set_operations.h.
 *
 * The same command line as oracle/ref_setops_driver.c (which links the REFERENCE's
 * set-operations.c), so the two can be run side by side on the same .list files:
 *
 *   setops_driver write_union CUTOFF OUT.list L1 L2 ...   -> writes OUT.list, prints NUnique/NTotal
 *   setops_driver union L1 L2 ...                         -> one line per callback: key\tc0\tc1...
 *   setops_driver is_union L1 L2 ...                      -> same, gt4_is_union
 *   setops_driver union_stop N L1 L2 ...                  -> callback returns 7 on its N-th call
 *
 * and glistquery's multi-list forms, printed as the reference's glistquery prints them
 * (src/glistquery.c:82-106, :776-812, :702-717):
 *
 *   setops_driver dump L1 L2 ...            = glistquery L1 L2 ...             (KMER\tc0\tc1...)
 *   setops_driver dump_is_union L1 L2 ...   = glistquery L1 L2 ... --is_union
 *   setops_driver search_multi Q L1 L2 ...  = glistquery L1 L2 ... -l Q         (KMER\t0:c\t1:c...)
 *   setops_driver zipper L Q                = glistquery L -l Q                 (KMER\tcount)
 */
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "gt4_set_operations.h"

static unsigned int n_lists_g;
static unsigned long long stop_after = 0, calls = 0;

static unsigned int print_cb (uint64_t word, uint32_t *counts, void *data)
{
  (void) data;
  printf ("%llu", (unsigned long long) word);
  for (unsigned int j = 0; j < n_lists_g; j++) printf ("\t%u", counts[j]);
  printf ("\n");
  calls += 1;
  return (stop_after && calls == stop_after) ? 7 : 0;
}

static unsigned int wlen_g;

static unsigned int dump_cb (uint64_t word, uint32_t *counts, void *data)
{
  char b[64];
  (void) data;
  gt4_word2string (b, word, wlen_g);
  fputs (b, stdout);
  for (unsigned int j = 0; j < n_lists_g; j++) printf ("\t%u", counts[j]);
  printf ("\n");
  return 0;
}

static uint64_t multi_last = 0;
static int multi_open = 0;

static unsigned int multi_cb (uint64_t word, unsigned int list, uint32_t count, void *data)
{
  (void) data;
  if (!multi_open || word != multi_last) {
    char b[64];
    if (multi_open) printf ("\n");
    gt4_word2string (b, word, wlen_g);
    fputs (b, stdout);
    multi_open = 1;
    multi_last = word;
  }
  printf ("\t%u:%u", list, count);
  return 0;
}

static unsigned int zipper_cb (uint64_t word, uint32_t count, void *data)
{
  char b[64];
  (void) data;
  gt4_word2string (b, word, wlen_g);
  printf ("%s\t%u\n", b, count);
  return 0;
}

int main (int argc, const char **argv)
{
  GT4HipWordList *objs[64];
  unsigned int n = 0, r;
  if (argc < 3) return 2;
  const int first = !strcmp (argv[1], "write_union") ? 4 : (!strcmp (argv[1], "union_stop") ? 3 : 2);
  for (int i = first; i < argc && n < 64; i++) {
    objs[n] = gt4_hip_word_list_new (argv[i], 4);
    if (!objs[n]) return 3;
    n++;
  }
  n_lists_g = n;
  wlen_g = n ? gt4_hip_word_list_word_length (objs[0]) : 0;
  if (!strcmp (argv[1], "dump") || !strcmp (argv[1], "dump_is_union")) {
    r = !strcmp (argv[1], "dump") ? gt4_union (objs, n, dump_cb, NULL) : gt4_is_union (objs, n, dump_cb, NULL);
    for (unsigned int j = 0; j < n; j++) gt4_hip_word_list_delete (objs[j]);
    return (int) r;
  }
  if (!strcmp (argv[1], "search_multi") && n >= 2) {
    n_lists_g = n - 1;
    r = gt4_search_lists_multi (objs[0], objs + 1, n - 1, multi_cb, NULL);
    if (multi_open) printf ("\n");
    return (int) r;
  }
  if (!strcmp (argv[1], "zipper") && n == 2) return (int) gt4_search_list_zipper (objs[0], objs[1], zipper_cb, NULL);
  if (!strcmp (argv[1], "write_union")) {
    GT4ListHeader h;
    int fd = creat (argv[3], 0644);
    if (fd < 0) return 4;
    r = gt4_write_union (objs, n, (unsigned int) strtoul (argv[2], NULL, 10), fd, &h);
    close (fd);
    printf ("NUnique\t%llu\nNTotal\t%llu\n", (unsigned long long) h.n_words, (unsigned long long) h.total_count);
  } else if (!strcmp (argv[1], "union")) {
    r = gt4_union (objs, n, print_cb, NULL);
  } else if (!strcmp (argv[1], "union_stop")) {
    stop_after = strtoull (argv[2], NULL, 10);
    r = gt4_union (objs, n, print_cb, NULL);
  } else if (!strcmp (argv[1], "is_union")) {
    r = gt4_is_union (objs, n, print_cb, NULL);
  } else {
    return 2;
  }
  printf ("result\t%u\n", r);
  for (unsigned int j = 0; j < n; j++) gt4_hip_word_list_delete (objs[j]);
  return 0;
}
