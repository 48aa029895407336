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
