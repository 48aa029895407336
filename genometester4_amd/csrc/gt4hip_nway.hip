/*
 * gt4hip_nway.hip -- N-way union of up to eight sorted lists in ONE pass over HBM.
 *
 * What it restates: union_multi (reference src/glistcompare.c:500-603; hot loop :545-591) and
 * gt4_write_union (src/set-operations.c:40-129): for every distinct key ascending, the count is
 * the sum / maximum / override over the lists that hold the key, kept iff count >= cutoff.
 *
 * Why a second kernel: the pairwise tree of k_pair_merge moves every record log2(N) times through
 * HBM (246 GB for eight 5e8-record lists whose algorithmic traffic is 78 GB), and a rank search per
 * record and level is what the pair kernel spends its instructions on.  Here a workgroup owns one
 * TILE of the merged key sequence -- a key range cut out of all the lists at once -- and orders the
 * tile's records WITHOUT any search, at a cost per record that does not depend on the number of lists:
 *
 *   1. every lane holds one record (fetched one tile ahead with 12-byte buffer loads, 64 consecutive
 *      records of ONE list per wave-instruction);
 *   2. a monotone bucket number from the key by interpolation inside the tile's key range (two
 *      buckets per position on average), one LDS atomic per record counts the bucket and hands the
 *      record its arrival number; a workgroup scan turns the counts into bucket starts;
 *   3. the keys are stored grouped by bucket; a record's rank inside its bucket is the number of
 *      smaller keys there -- a handful of INDEPENDENT 8-byte LDS reads (no dependent chain), all
 *      lanes running the same number of steps;
 *   4. position = bucket start + rank = number of smaller keys in the tile.  Equal keys of different
 *      lists get the same position: the counts are FOLDED there by an LDS atomic (add: u32 wrap as the
 *      reference's unsigned sum; max);
 *   5. (union, count-only: GT4_NWAY_LEAD) the first record to set its position's bit in a bitmap LEADS
 *      the position: it reads the folded count, applies the cutoff, and its place in the staging area is
 *      the number of bits below its own (popcount prefix, worked out by every wavefront for itself);
 *      (merged samples, count tables) the key is stored once per position, a byte marks the position
 *      live, positions in order: live ones are the tile's output, ballots and prefix sums compact them;
 *      the staging area is written out during the next tile at the offset the chained scan of tile
 *      totals (gt4hip_device.h) has published by then.
 *   Every per-record step is straight-line code: all LDS reads of a thread's records before the first
 *   wait, lanes without a (kept) record store into trash rows instead of branching around the store.
 *
 * The interpolation is only a heuristic for SPEED: a tile whose keys cluster (a bucket with more than
 * NWAY_TRY0 keys) is bucketed again by rank in its longest run (NWAY_LIMIT), and one that defeats that
 * too takes a bounded fallback -- the records go back to LDS as sorted runs and every record adds up
 * its lower bounds in all the runs (binary searches) -- and continues at step 4.
 *
 * The same tile kernel also builds glistquery's count tables (gt4_union / gt4_is_union /
 * search_lists_multi: src/set-operations.c:131-228, src/glistquery.c:776-812): NWAY_TABLE (all distinct keys,
 * a column per list; one launch, every tile's rows where its records start: a ragged table), NWAY_PROBE (the keys
 * of list 0).
 *
 *   K5 k_nway_sample          every S-th key of every list -> "sample lists" (1/S of the data)
 *   K6 k_nway_sample_counts,  tile boundaries: the merged samples' every G-th key; a merged sample carries its
 *      k_nway_bracket_bases,  list, so the samples of list i in front of a boundary are a prefix count and the
 *      k_nway_partition_rows  cut lies in the S records behind them (all records with a key <= the boundary key
 *                             go left), plus the tile's key range, interpolation constants and whether its samples
 *                             look clustered; k_nway_partition: the same by searches over brackets of 64 tiles (the
 *                             topmost level, which has no merged samples; option "kway_vt" = 97)
 *      k_nway_need, _need_scan, a tile that would exceed the LDS capacity is cut in two at the middle key of its longest run
 *      k_nway_emit            (more than two pieces: the host retries with fewer samples per tile, down to the number for
 *                             which it cannot happen)
 *   K7 k_nway_merge           the tile kernel; NWAY_DUPS keeps every record and remembers its list (it is how the
 *                             sample lists themselves are merged, one level up: the recursion ends when a level
 *                             fits one tile), NWAY_UNION / NWAY_COUNT fold equal keys and apply the rule,
 *                             NWAY_TABLE / NWAY_PROBE write count tables
 *      k_nway_base_sums, _scan, rows before every tile and where its rows lie (the ragged count table's index)
 *      k_nway_tile_bases,
 *      k_nway_padded_bases
 */
#define GT4_RESOLVE_LOOKBACK 0
#include "gt4hip_device.h"
#include "gt4hip_host.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#define GT4_KM 8
#define GT4_KM_NS km8
#define GT4_KM_ROWS "gt4hip_nway_rows8.h"
#include "gt4hip_nway_body.h"
#undef GT4_KM
#undef GT4_KM_NS
#undef GT4_KM_ROWS
#define GT4_KM 32
#define GT4_KM_NS km32
#define GT4_KM_ROWS "gt4hip_nway_rows32.h"
#include "gt4hip_nway_body.h"
#undef GT4_KM
#undef GT4_KM_NS

using namespace gt4;

/* N-way union of 2..8 non-empty lists in one pass.  *used = 0 when the call must take the pairwise
 * tree instead (the single-pass chain gave up on a shared device).  `out`: capacity >= sum of the
 * lists (unless count_only). */
int gt4hip_nway_union (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t k, uint32_t rule, uint32_t cutoff, uint32_t ovr,
                       uint32_t filter, bool count_only, gt4hip_list *out, uint64_t *n_words, uint64_t *total_count, double *device_ms,
                       int *used)
{
  if (k > 8) return km32::nway_run (ctx, lists, k, rule, cutoff, ovr, filter, count_only, out, n_words, total_count, device_ms, used, NULL, NULL, false);
  return km8::nway_run (ctx, lists, k, rule, cutoff, ovr, filter, count_only, out, n_words, total_count, device_ms, used, NULL, NULL, false);
}

/* The count table of 2..32 non-empty lists (all their distinct keys ascending; column cols[i] = list i's
 * count of the key, 0 where it has none) by ONE launch of the tile kernel: keys and counts written where every
 * tile's records start (a ragged table, allocated for the lists' records; GT4HIP_ENOMEM: the caller builds it by merges).  table->n_lists columns (those no list
 * is given for stay 0).  *used = 0: nothing was done, the caller builds the table by merges. */
int gt4hip_nway_table (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t k, const uint32_t cols[], gt4hip_count_table *table, int probe,
                       int presence, int *used)
{
  /* probe: rows = the records of lists[0] (which must be the first, non-empty); presence: 1 instead of the count */
  uint64_t n = 0, t = 0;
  double ms = 0;
  *used = 0;
  if (k > 32) return GT4HIP_OK; /* (count tables of more than 32 lists: the caller builds them by merges) */
  const int rc = k > 8 ? km32::nway_run (ctx, lists, k, probe && presence ? 7 : 1, 0, 1, FILTER_RAW, true, NULL, &n, &t, &ms, used, table, cols, probe != 0)
                       : km8::nway_run (ctx, lists, k, probe && presence ? 7 : 1, 0, 1, FILTER_RAW, true, NULL, &n, &t, &ms, used, table, cols, probe != 0);
  if (rc || !*used) gt4hip_table_free (table);
  return rc;
}
