/*
 * shard_plan_harness.c -- CPU-only unit harness for the chunk planner of the sharded command-line
 * path (genometester4_amd/csrc/gt4_shard.c: make_plan).  TEST INFRASTRUCTURE: includes the product
 * source and stubs the device library out (nothing here merges anything); built and run by
 * tests/test_shard_plan.py.
 *
 *   shard_plan_harness <n_ranks> <hbm_limit> <mode> <ops> <file>...
 *
 * prints "chunks C" and one line per file with its C + 1 cut indices.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>

/* the planner needs none of the device entry points: satisfy the linker */
#include "gt4hip.h"
int gt4hip_create (int d, gt4hip_context **c) { (void) d; (void) c; return 2; }
void gt4hip_destroy (gt4hip_context *c) { (void) c; }
const char *gt4hip_last_error (const gt4hip_context *c) { (void) c; return "stub"; }
int gt4hip_device_count (void) { return 0; }
const char *gt4hip_device_info (const gt4hip_context *c) { (void) c; return ""; }
int gt4hip_device_memory (gt4hip_context *c, uint64_t *f, uint64_t *t) { (void) c; (void) f; (void) t; return 1; }
int gt4hip_set_option (gt4hip_context *c, const char *n, int64_t v) { (void) c; (void) n; (void) v; return 0; }
int gt4hip_list_upload_fd (gt4hip_context *c, int fd, uint64_t o, uint64_t n, uint32_t w, gt4hip_list **l) { (void) c; (void) fd; (void) o; (void) n; (void) w; (void) l; return 1; }
int gt4hip_list_upload_index (gt4hip_context *c, const void *k, uint64_t n, uint64_t nl, uint32_t w, gt4hip_list **l) { (void) c; (void) k; (void) n; (void) nl; (void) w; (void) l; return 1; }
int gt4hip_list_alloc (gt4hip_context *c, uint64_t n, uint32_t w, gt4hip_list **l) { (void) c; (void) n; (void) w; (void) l; return 1; }
void gt4hip_list_free (gt4hip_list *l) { (void) l; }
int gt4hip_lists_write_fd (gt4hip_context *c, uint32_t n, const gt4hip_list *const l[], const uint64_t f[], const uint64_t cn[], const int fd[], const uint64_t o[]) { (void) c; (void) n; (void) l; (void) f; (void) cn; (void) fd; (void) o; return 1; }
int gt4hip_compare (gt4hip_context *c, const gt4hip_list *a, const gt4hip_list *b, const gt4hip_compare_params *p, gt4hip_compare_result *r) { (void) c; (void) a; (void) b; (void) p; (void) r; return 1; }
int gt4hip_union_multi (gt4hip_context *c, const gt4hip_list *const l[], uint32_t n, uint32_t cu, int32_t ru, uint32_t o, int32_t co, gt4hip_multi_result *r) { (void) c; (void) l; (void) n; (void) cu; (void) ru; (void) o; (void) co; (void) r; return 1; }
int gt4hip_intersect_multi (gt4hip_context *c, const gt4hip_list *const l[], uint32_t n, uint32_t cu, int32_t ru, uint32_t o, int32_t co, gt4hip_multi_result *r) { (void) c; (void) l; (void) n; (void) cu; (void) ru; (void) o; (void) co; (void) r; return 1; }
int gt4hip_comm_unique_id (void *id) { (void) id; return 1; }
const char *gt4hip_comm_last_error (void) { return "stub"; }
int gt4hip_comm_create (gt4hip_context *c, const void *id, int n, int r, gt4hip_comm **o) { (void) c; (void) id; (void) n; (void) r; (void) o; return 1; }
void gt4hip_comm_destroy (gt4hip_comm *c) { (void) c; }
int gt4hip_comm_gatherv (gt4hip_comm *c, const gt4hip_list *l, const uint64_t cn[], int r, gt4hip_list *g) { (void) c; (void) l; (void) cn; (void) r; (void) g; return 1; }

#include "../../genometester4_amd/csrc/gt4_shard.c"

int main (int argc, char **argv)
{
  if (argc < 6) return 2;
  static GT4ListFile files[64];
  GT4ShardJob job;
  memset (&job, 0, sizeof job);
  job.n_ranks = atoi (argv[1]);
  const uint64_t limit = strtoull (argv[2], NULL, 10);
  job.mode = atoi (argv[3]);
  job.prm.ops = (uint32_t) atoi (argv[4]);
  job.n_files = (unsigned int) (argc - 5);
  for (unsigned int f = 0; f < job.n_files; f++)
    if (gt4_listfile_open (argv[5 + f], 4, &files[f])) return 3;
  job.files = files;
  job.word_length = files[0].header.word_length;
  Plan p;
  if (make_plan (&job, limit, &p)) return 4;
  printf ("chunks %u\n", p.n_chunks);
  for (unsigned int f = 0; f < job.n_files; f++) {
    for (unsigned int c = 0; c <= p.n_chunks; c++) printf ("%llu%c", (unsigned long long) plan_cut (&p, f, c), c == p.n_chunks ? '\n' : ' ');
  }
  return 0;
}
