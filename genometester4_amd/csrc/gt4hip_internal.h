/* gt4hip_internal.h -- shared between the kernel file and the C-ABI implementation. */
#ifndef GT4HIP_INTERNAL_H
#define GT4HIP_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

/* The one compile-time switch of the device code: the diagnostics build (`make prof`, -DGT4_PROFILE_PHASES=<thread whose
 * phases are stamped>) keeps what PROF (...) encloses -- phase stamps, scanner and resolve statistics; the product build
 * drops it. */
#ifdef GT4_PROFILE_PHASES
#define PROF(...) __VA_ARGS__
#else
#define PROF(...)
#endif

namespace gt4 {

/* Internal rule code on top of the reference's enum Rules (src/glistcompare.c:45-54):
 * the running minimum of intersect_multi, which restarts whenever it is 0
 * (`if (!freq || c < freq) freq = c`, src/glistcompare.c:669). */
constexpr uint32_t RULE_MINZ = 8;

/* How an output stream decides to keep a key once its count is computed. */
enum Filter : uint32_t {
  FILTER_REFERENCE = 0, /* include_in_{union,intersection,complement}, src/glistcompare.c:459-489   */
  FILTER_RAW = 1,       /* keep every key of the stream's domain (intermediate N-way levels)        */
  FILTER_RESULT = 2     /* keep iff count >= cutoff (union_multi/intersect_multi, :574, :682)       */
};

struct PairParams {
  uint32_t ops;            /* bit s: stream s is produced (0 union, 1 intrsec, 2 diff1, 3 diff2) */
  uint32_t rule[4];        /* resolved rule per stream (never DEFAULT)                           */
  uint32_t cutoff;
  uint32_t subtract;       /* diff1 only                                                         */
  uint32_t count_override;
  uint32_t filter;
  uint32_t spin_limit;     /* bound of every inter-workgroup wait (0: the default, ~seconds); tests set it low */
  uint32_t scan_group;     /* 0: one scanner wavefront per stream; 1: summers + chainer (launches with many rows) */
  uint32_t dynamic;        /* 0: tiles dealt round-robin; 1: by a ticket counter (ctl->ticket), three tiles ahead */
};

struct PairOutputs {
  uint32_t *rec[4];        /* packed 12-byte records as dwords; may be null in count mode */
};

/* Control block in device memory, zeroed before every launch. */
struct PairControl {
  unsigned long long n_words[4];
  unsigned long long total_count[4];
  unsigned int ticket;     /* dynamic dealing: next tile */
  unsigned int error;      /* non-zero: a bounded spin gave up / consistency check tripped */
  unsigned int role;       /* first workgroup to arrive becomes the scanner */
  unsigned int pad;
  unsigned long long phase_cycles[24];
  unsigned long long resolve_stats[8]; /* diagnostic builds: sampled resolve calls, spins, -, agg/carry not ready at first look; scanner rounds, rows retired on the first look, rows */ /* diagnostic builds (-DGT4_PROFILE_PHASES): shader cycles per phase, summed over workgroups */
};

enum MergeMode : int {
  MODE_COUNT = 0,     /* totals only (--count_only), also pass 1 of the two-pass path: writes tile counts */
  MODE_LOOKBACK = 1,  /* single pass: a scanner wavefront chains tile totals into output offsets          */
  MODE_OFFSETS = 2    /* pass 2 of the two-pass path: tile offsets already scanned                       */
};

/* Geometry of the merge kernel: workgroups of 512 threads (geom 0: count-only
 * calls) or 1024 threads (geom 1: calls that materialise records), MERGE_VT positions per thread --
 * 6 for the single-output intersection (merge_ipt in gt4hip_kernels.hip).  A tile holds
 * threads x positions - 64 records (the pair fix-up makes it +-1): in the workgroup's position
 * space the B records start at the next multiple of 64 after the A records, so that no 64-position
 * chunk mixes the two lists, and both record ranges fit in 16-byte chunks. */
constexpr int MERGE_VT = 4;
constexpr int MERGE_TILE_SLACK = 64;

uint64_t merge_tile_records (int geom, uint32_t ops);
hipError_t launch_partition (hipStream_t s, const uint32_t *A, uint64_t nA, const uint32_t *B, uint64_t nB,
                             uint64_t num_tiles, uint64_t tile_records, uint64_t *part);
hipError_t launch_pair_merge (hipStream_t s, int geom, int mode, int grid, const uint32_t *A, uint64_t nA,
                              const uint32_t *B, uint64_t nB, const uint64_t *part, uint64_t num_tiles,
                              const PairParams &p, const PairOutputs &o, unsigned long long *desc,
                              PairControl *ctl);
hipError_t launch_scan_tiles (hipStream_t s, unsigned long long *desc, uint64_t num_tiles,
                              unsigned long long *block_sums);
hipError_t launch_generate (hipStream_t s, uint32_t *rec, uint64_t n, uint64_t stride, uint64_t seed,
                            uint64_t count_seed, uint32_t max_count, uint64_t mult, uint64_t add);
hipError_t launch_sum_counts (hipStream_t s, const uint32_t *rec, uint64_t n, unsigned long long *sum);
hipError_t launch_check_sorted (hipStream_t s, const uint32_t *rec, uint64_t n, unsigned int *bad);
hipError_t launch_lower_bound (hipStream_t s, const uint32_t *rec, uint64_t n, uint64_t key,
                               unsigned long long *idx);
hipError_t launch_extract_column (hipStream_t s, const uint32_t *rec, uint64_t n, uint32_t *counts, uint32_t n_lists, uint32_t column);
hipError_t launch_extract_keys (hipStream_t s, const uint32_t *rec, uint64_t n, unsigned long long *keys);
hipError_t launch_decode_index (hipStream_t s, const unsigned long long *kmers, uint64_t n, uint64_t num_locations, uint32_t *rec);

int merge_blocks_per_cu (int geom, int mode, uint32_t ops, const PairParams *p = nullptr);

}  // namespace gt4

#endif
