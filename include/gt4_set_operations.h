/*
 * gt4_set_operations.h -- the three entry points of the reference's set-operations.h with their
 * exact signatures (reference src/set-operations.h:34, 38, 39), backed by the MI355X engine.
 *
 * The only change is the list handle: where the reference takes `AZObject *` objects implementing
 * GT4WordSList, these take `GT4HipWordList *` (a sorted list resident in HBM).  glistmaker's
 * collation step (src/glistmaker.c:333, :814) and glistquery's multi-list dump
 * (src/glistquery.c:95-106) link against them unchanged otherwise; see INTEGRATION.md.
 */
#ifndef GT4_SET_OPERATIONS_H
#define GT4_SET_OPERATIONS_H

#include <stdint.h>

#include "gt4_listfile.h"
#include "gt4hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GT4_MAX_SETS 4096 /* src/set-operations.h:29 */

typedef struct _GT4HipWordList GT4HipWordList;

/* The device context every GT4HipWordList of this process lives in (created on first use on
 * device $GT4HIP_DEVICE, default 0).  NULL (and a message on stderr) when there is no GPU. */
gt4hip_context *gt4_hip_default_context (void);
/* Hosts that manage their own contexts (several GPUs, several threads) install the one the
 * GT4HipWordList functions of the calling process shall use from now on; NULL forgets it (the
 * previous one is not destroyed: whoever created it does that).  Returns the previous one. */
gt4hip_context *gt4_hip_set_default_context (gt4hip_context *ctx);

/* gt4_word_map_new (src/word-map.c:165-241) for the GPU path: map, validate, upload.  A GT4I index
 * file is accepted as the sorted k-mer list it contains (gt4_index_map_new, src/index-map.c:317-373).
 * NULL on failure (diagnostic on stderr).
 * A file larger than the resident share of the device memory ($GT4HIP_HBM_LIMIT bytes, K / M / G suffixes
 * accepted; default: a quarter of what the device has free) is NOT uploaded: the handle stays FILE-BACKED (the file
 * mapped, as the reference's GT4WordMap is).  gt4_write_union over file-backed handles streams the inputs through
 * the device in key-range chunks and writes `ofile` as it goes -- inputs and output of any size, e.g. glistmaker's
 * collation of up to 32 temporary lists (reference src/glistmaker.c:787-835); every other entry point uploads a
 * file-backed list on first use (and fails with a message if it does not fit). */
GT4HipWordList *gt4_hip_word_list_new (const char *listfilename, unsigned int major_version);
/* Wraps packed records already in host memory (copied to HBM). */
GT4HipWordList *gt4_hip_word_list_new_from_records (const void *records, uint64_t n_words, unsigned int word_length);
void gt4_hip_word_list_delete (GT4HipWordList *list);
/* GT4WordSListInstance fields, src/word-list-sorted.h:45-57 */
uint64_t gt4_hip_word_list_num_words (const GT4HipWordList *list);
uint64_t gt4_hip_word_list_sum_counts (const GT4HipWordList *list);
unsigned int gt4_hip_word_list_word_length (const GT4HipWordList *list);
const gt4hip_list *gt4_hip_word_list_device (const GT4HipWordList *list); /* NULL while a file-backed list has not been uploaded */
int gt4_hip_word_list_is_file_backed (const GT4HipWordList *list);

/* The iterator of the reference's sorted word lists -- GT4WordSListInstance with get_first_word / get_next_word
 * (src/word-list-sorted.h:42-57, src/word-list-sorted.c:59-78) -- over a GT4HipWordList, for callers that walk ONE
 * list record by record (glistquery's statistics, src/glistquery.c:560-640): the records come to the host in
 * blocks of 2^20 (from HBM, or straight from the mapping of a file-backed handle).  Same contract: both return 1
 * while `word` / `count` hold a record and 0 at the end or on error; get_next_word on the last record returns 0
 * WITHOUT touching word / count (callers test idx < num_words, as the reference's do). */
typedef struct {
  uint64_t num_words, sum_counts; /* of the list */
  uint64_t idx;                   /* index of the current record */
  uint64_t word;
  uint32_t count;
  unsigned int word_length;
  GT4HipWordList *list;           /* private from here on */
  void *block;
  uint64_t block_first, block_count;
} GT4HipWordSListIter;
unsigned int gt4_hip_word_slist_get_first_word (GT4HipWordList *list, GT4HipWordSListIter *it);
unsigned int gt4_hip_word_slist_get_next_word (GT4HipWordSListIter *it);
void gt4_hip_word_slist_iter_release (GT4HipWordSListIter *it); /* frees the block (the list stays the caller's) */

/* Combines N lists into one union (counts added, u32 wrap), keeps keys whose sum >= cutoff,
 * writes header + records to `ofile` when it is non-zero, always fills *header.
 * Returns 0 on success (reference src/set-operations.c:40-129). */
unsigned int gt4_write_union (GT4HipWordList *arrays[], unsigned int n_arrays, unsigned int cutoff, int ofile, GT4ListHeader *header);

/* Executes callback for each distinct key ascending with counts[j] = count in list j or 0.
 * A non-zero callback result stops the walk and is returned (src/set-operations.c:131-183).
 * Reproduces the reference's observable quirk: after the last key of a list, while other lists
 * remain, the callback is invoked once more for that key with all-zero counts. */
unsigned int gt4_union (GT4HipWordList *objs[], unsigned int n_objs, unsigned int (*callback) (uint64_t, uint32_t *, void *), void *data);

/* Walks the keys of list 0 only and reports the other lists' counts for them
 * (src/set-operations.c:185-228). */
unsigned int gt4_is_union (GT4HipWordList *objs[], unsigned int n_objs, unsigned int (*callback) (uint64_t, uint32_t *, void *), void *data);

/* glistquery's list-against-lists searches, as merges on the device:
 *
 * gt4_search_lists_multi -- search_lists_multi, reference src/glistquery.c:776-812: for every word of
 * `query` ascending, callback (word, j, count) for every list j (ascending) that holds the word; words
 * found in no list are skipped.  A non-zero callback result stops the walk and is returned.
 *
 * gt4_search_list_zipper -- search_list_zipper, src/glistquery.c:702-717: callback (word, count in
 * `query`) for every word of `query` that `list` holds, ascending. */
unsigned int gt4_search_lists_multi (GT4HipWordList *query, GT4HipWordList *lists[], unsigned int n_lists,
                                     unsigned int (*callback) (uint64_t word, unsigned int list, uint32_t count, void *data), void *data);
unsigned int gt4_search_list_zipper (GT4HipWordList *list, GT4HipWordList *query,
                                     unsigned int (*callback) (uint64_t word, uint32_t count, void *data), void *data);
/* word2string, src/sequence.c:103-114: 2 bits per base, A C G T, first base in the highest bits.
 * `b` needs wordlength + 1 bytes; returns wordlength. */
unsigned int gt4_word2string (char *b, uint64_t word, unsigned int wordlength);

#ifdef __cplusplus
}
#endif
#endif
