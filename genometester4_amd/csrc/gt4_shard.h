/*
 * gt4_shard.h -- key-range sharded execution of the glistcompare operations for the C host:
 * several GPUs of one node (one worker process per GPU) and/or inputs larger than the device
 * memory (key-range chunks streamed through one GPU), SURVEY 8e + 8f N1.
 *
 * Replaces, for the multi-list job, the tree of glistcompare processes exchanging .list files on
 * disk that scripts/MakeUnion.pl:31-95 drives, and for big inputs the reference's reliance on
 * mmap paging (src/utils.c:35-64) / 3 KiB read()s (src/word-list-stream.c:85-125).
 */
#pragma once

#include <stdint.h>

#include "gt4_listfile.h"
#include "gt4hip.h"

enum { GT4_SHARD_PAIR = 0, GT4_SHARD_UNION_MULTI = 1, GT4_SHARD_INTERSECT_MULTI = 2 };

typedef struct {
  /* inputs (mapped by the caller; strictly ascending keys) */
  unsigned int n_files;
  const GT4ListFile *files;
  unsigned int word_length;
  /* operation */
  int mode;                      /* GT4_SHARD_*                                                  */
  gt4hip_compare_params prm;     /* PAIR: ops, rule, cutoff, subtract, count_override, count_only */
  /* outputs: final file names per stream (union, intrsec, diff1, diff2; N-way: slot 0), NULL = not
   * produced; written as <name>.tmp then renamed */
  const char *out_name[4];
  unsigned int out_mode;         /* creation mode of the output files                            */
  /* ... or, single worker only (n_ranks = 1), a file the CALLER has opened (gt4_write_union's `ofile`,
   * include/gt4_set_operations.h): out_fd[s] > 0 with out_name[s] = NULL -- the records go to byte
   * out_base[s] + 12 * (records before); no header, no temporary, no rename, the descriptor stays open */
  int out_fd[4];
  uint64_t out_base[4];
  /* plan */
  int n_ranks;                   /* worker processes = GPUs (1: no fork, runs in the caller)     */
  uint64_t hbm_limit;            /* device bytes a worker may hold in flight; 0: 70 % of what is free */
  int auto_budget;               /* hbm_limit 0 and inputs that fit the device: the budget by mode (see worker_main) */
  int device_plus_1;             /* single worker: the device to run on + 1 (0: $GT4HIP_DEVICE, else device 0) -- the caller's context may live elsewhere */
  int gather_rccl;               /* 0: every rank pwrites its extents; 1: RCCL gatherv to rank 0 */
  int debug;
} GT4ShardJob;

typedef struct {
  uint64_t n_words[4];
  uint64_t total_count[4];
  unsigned int n_chunks;
  int rule_rejected;             /* N-way: the library refused the rule (message in `message`)   */
  char message[512];
} GT4ShardResult;

/* Runs the job.  Returns 0 on success (outputs renamed into place, totals in `res`), 1 on failure
 * (message on stderr, temporary files removed).  With n_ranks > 1 the caller must not have touched
 * the HIP runtime yet: the workers are forked here, before any HIP call. */
int gt4_shard_run (const GT4ShardJob *job, GT4ShardResult *res);

/* Index of the first record of a mapped list / index file with key >= `key` (host binary search). */
uint64_t gt4_listfile_lower_bound (const GT4ListFile *lf, uint64_t key);
uint64_t gt4_listfile_key_at (const GT4ListFile *lf, uint64_t idx);
