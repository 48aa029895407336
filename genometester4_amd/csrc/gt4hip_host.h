/* gt4hip_host.h -- host-side internals shared by the C-ABI implementation files
 * (gt4hip_api.hip, gt4hip_io.hip, gt4hip_comm.hip, gt4hip_nway.hip). */
#ifndef GT4HIP_HOST_H
#define GT4HIP_HOST_H

#include "../../include/gt4hip.h"
#include "gt4hip_internal.h"

#include <stddef.h>
#include <utility>
#include <vector>

struct gt4hip_io; /* gt4hip_io.hip: pinned staging + copy threads, created on first use */

struct gt4hip_context {
  int device;
  hipStream_t stream;
  hipEvent_t ev[4];
  int n_cus;
  int two_pass;
  int64_t grid_override;
  uint32_t spin_limit;       /* option "spin_limit": bound of the single-pass kernel's waits (0 = default) */
  int dynamic;     /* option "dynamic": tiles of the single-pass kernels dealt by a ticket counter: 0 automatic, 1 always, -1 never (round-robin) */
  int scan_group;  /* option "scan_group": 0 automatic, 1 always the scanner group, -1 always one wavefront per stream */
  int force_geom; /* options "geom1" / "geom0": force the large / small geometry for every call (experiments); 0 = automatic */
  uint64_t single_pass_fallbacks; /* calls that had to be rerun on the two-pass path */
  /* freed list storage kept for reuse: hipMalloc / hipFree of tens of GB cost far more than the
   * merges themselves (an 8-way union tree allocates seven outputs per call) */
  std::vector<std::pair<void *, size_t>> *pool;
  int pool_enabled;
  size_t pool_bytes;         /* bytes the pool holds right now */
  size_t pool_cap;           /* most it may hold (option "pool_cap_mb"; default: half of the device memory) */
  /* workspace, grown on demand */
  uint64_t *part;
  size_t part_bytes;
  unsigned long long *desc;
  size_t desc_bytes;
  unsigned long long *block_sums;
  size_t block_sums_bytes;
  gt4::PairControl *ctl;          /* device */
  gt4::PairControl *ctl_host;     /* pinned */
  unsigned long long *scratch;      /* device, 4 x u64 */
  unsigned long long *scratch_host; /* pinned */
  /* N-way tile kernel (gt4hip_nway.hip) */
  uint64_t *kway_part;       /* tile boundaries, [tiles + 1][8] */
  size_t kway_part_bytes;
  void *kway_cnt;            /* samples of every list per bracket of 64 tiles */
  size_t kway_cnt_bytes;
  uint64_t *kway_part2;      /* the boundaries with the tiles that would not fit LDS cut in two (the table the tile kernel reads) */
  size_t kway_part2_bytes;
  void *kway_need;           /* tiles each nominal tile becomes (u32), block sums (u32) */
  size_t kway_need_bytes;
  uint64_t kway_splits;      /* counter "kway_splits": tiles cut in two by the last N-way call */
  int kway_enabled;          /* option "kway": 0 = always the pairwise tree, 1 = the one-pass kernel unless the keys are clustered, 2 = always, also for two lists, 3 = always (three lists and more) */
  int kway_max;              /* option "kway_max": lists per launch of the tile kernel, 32 (default: unless the lists share most keys), 8, or 33 (= 32 whatever the keys) */
  uint64_t kway_width;       /* counter "kway_width": lists per launch the last N-way union took (8 or 32) */
  uint64_t kway_shared_x100; /* counter "kway_shared_x100": 100 x the mean number of lists a probed key lies in (last union of more than eight lists) */
  int64_t kway_g;            /* option "kway_g": samples per tile (0 = automatic) */
  int64_t kway_vt;           /* option "kway_vt": positions per thread in a merge pass, at least (0 = default) */
  uint64_t kway_overflows;   /* calls that fell back to the tree because a tile would not fit LDS */
  uint64_t kway_calls;       /* N-way unions done by the one-pass kernel */
  uint64_t kway_declined;    /* ... handed to the pairwise tree because the keys are clustered (option "kway" = 1) */
  double table_ms;           /* the last gt4hip_union_table, wall time of the call */
  double sort_ms, fold_ms;   /* the last gt4hip_device_words_to_list: radix sort and fold (HIP events) */
  double nway_kernel_ms;     /* the last one-pass launch's kernel time (HIP events on the library's stream) and tiles */
  uint64_t nway_tiles;
  int last_multi_one_pass;   /* the last gt4hip_union_multi was done by the one-pass tile kernel (counter "nway_one_pass") */
  gt4hip_io *io;            /* file <-> HBM staging (gt4hip_io.hip), NULL until first used */
  char err[512];
  char info[256];
};

struct gt4hip_list {
  gt4hip_context *ctx;
  void *dev;
  size_t bytes; /* size of the allocation behind dev when owned */
  uint64_t n_words;
  uint64_t capacity;
  uint32_t word_length;
  int owns;
};


int gt4hip_fail (gt4hip_context *ctx, int code, const char *fmt, ...);
/* every device allocation of the library: gives the pooled blocks back and retries when the driver is out of memory */
hipError_t gt4hip_dev_alloc (gt4hip_context *ctx, void **p, size_t bytes);
int gt4hip_list_new (gt4hip_context *ctx, uint64_t capacity, uint32_t word_length, gt4hip_list **out);
void gt4hip_io_destroy (gt4hip_context *ctx);
int gt4hip_nway_union (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t k, uint32_t rule, uint32_t cutoff, uint32_t ovr,
                       uint32_t filter, bool count_only, gt4hip_list *out, uint64_t *n_words, uint64_t *total_count, double *device_ms,
                       int *used);
int gt4hip_block_alloc (gt4hip_context *ctx, size_t bytes, void **dev, void **owner);
void gt4hip_block_free (void *owner);
int gt4hip_table_alloc (gt4hip_context *ctx, gt4hip_count_table *table, uint64_t n, uint32_t n_lists);
/* ragged tables (gt4hip_count_table.ragged): the index of `tiles` tiles -- rows before every tile (compact) and where
 * the tile's rows lie in the arrays (padded), tiles + 1 device u64 each */
int gt4hip_table_set_ragged (gt4hip_context *ctx, gt4hip_count_table *table, uint64_t tiles);
void *gt4hip_table_compact_bases (gt4hip_count_table *table);
void *gt4hip_table_padded_bases (gt4hip_count_table *table);
int gt4hip_nway_table (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t k, const uint32_t cols[], gt4hip_count_table *table, int probe,
                       int presence, int *used);
int gt4hip_io_download (gt4hip_context *ctx, const void *dev, void *host, size_t bytes);

#define HIPCHK(ctx, call)                                                                               \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) return gt4hip_fail ((ctx), e_ == hipErrorOutOfMemory ? GT4HIP_ENOMEM : GT4HIP_EHIP, \
                                              "%s failed: %s", #call, hipGetErrorString (e_));          \
  } while (0)

#endif
