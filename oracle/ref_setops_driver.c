/*
 * ref_setops_driver.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A tiny command-line driver (this repo's own code) that is linked against the
 * REFERENCE's set-operations.c / word-map.c, compiled where they lie under
 * /root/reference/src by oracle/Makefile into oracle/_ref/ref_setops.  It lets
 * the tests run the reference's exported gt4_write_union / gt4_union /
 * gt4_is_union (src/set-operations.h:34,38,39) on real .list files and capture
 * what they produce, to pin oracle/gt4_oracle.c.
 *
 *   ref_setops write_union CUTOFF OUT.list L1 L2 ...   -> writes OUT.list, prints NUnique/NTotal
 *   ref_setops union L1 L2 ...                         -> one line per callback: key\tc0\tc1...
 *   ref_setops is_union L1 L2 ...                      -> same, gt4_is_union
 *   ... union_stop N L1 L2 ...                         -> callback returns 7 on its N-th call
 */
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "set-operations.h"
#include "word-map.h"

int debug = 0;

static unsigned int n_lists_g;
static unsigned long long stop_after = 0, calls = 0;

static unsigned int
print_cb (uint64_t word, uint32_t *counts, void *data)
{
  unsigned int j;
  (void) data;
  printf ("%llu", (unsigned long long) word);
  for (j = 0; j < n_lists_g; j++) printf ("\t%u", counts[j]);
  printf ("\n");
  calls += 1;
  if (stop_after && calls == stop_after) return 7;
  return 0;
}

int
main (int argc, const char **argv)
{
  AZObject *objs[64];
  unsigned int n = 0, r;
  int first;
  if (argc < 3) return 2;
  if (!strcmp (argv[1], "write_union")) first = 4;
  else if (!strcmp (argv[1], "union_stop")) first = 3;
  else first = 2;
  for (int i = first; i < argc && n < 64; i++) {
    objs[n] = (AZObject *) gt4_word_map_new (argv[i], 4, 0, 0);
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
  return 0;
}
