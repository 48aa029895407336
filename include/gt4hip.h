/*
 * gt4hip.h -- C ABI of the MI355X (gfx950) sorted k-mer list set-operation engine.
 *
 * This is the drop-in boundary between a C host -- the glistcompare CLI in
 * genometester4_amd/csrc/gt4_glistcompare_cli.c, or GenomeTester4's own glistcompare.c / glistmaker.c /
 * glistquery.c with the binding shown in INTEGRATION.md -- and the hand-written HIP kernels.
 * Plain C types only; nothing here throws, exits or prints.  Every function returns
 * GT4HIP_OK (0) or a GT4HIP_E* code; gt4hip_last_error() gives the message.
 *
 * A "list" is an array of packed 12-byte records (u64 key LE + u32 count LE), strictly ascending
 * by key, resident in HBM -- the record layout of a GenomeTester4 .list file body
 * (reference src/word-map.h:89-99, src/word-list.h:61-72).
 *
 * Threading: one host thread per context.  The context owns its HIP stream and workspace;
 * callers never manage streams.  Inputs are borrowed for the duration of a call (the reference
 * maps them PROT_READ, src/utils.c:54); outputs belong to the caller and are released with
 * gt4hip_list_free().
 *
 * There is NO CPU fallback behind this interface: without a usable gfx950 device every entry
 * point fails with GT4HIP_ENODEVICE.
 */
#ifndef GT4HIP_H
#define GT4HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GT4HIP_RECORD_BYTES 12u

enum {
  GT4HIP_OK = 0,
  GT4HIP_EINVAL = 1,      /* bad argument                                   */
  GT4HIP_ENODEVICE = 2,   /* no usable GPU / HIP runtime failure at init     */
  GT4HIP_ENOMEM = 3,      /* device or host allocation failed                */
  GT4HIP_ERULE = 4,       /* rule not allowed for this operation (the reference returns 1:
                             src/glistcompare.c:518-523, :622-627)           */
  GT4HIP_EHIP = 5,        /* a HIP call or kernel failed                     */
  GT4HIP_EWORDLEN = 6,    /* lists of different word length                  */
  GT4HIP_EINTERNAL = 7,   /* in-kernel consistency check tripped             */
  GT4HIP_ECALLBACK = 8,   /* (internal) walk stopped by the caller's callback */
  GT4HIP_EIO = 9,         /* reading or writing a file descriptor failed       */
  GT4HIP_ECOMM = 10       /* RCCL could not be loaded / a collective failed    */
};

/* enum Rules of the reference, src/glistcompare.c:45-54 (same numeric values) */
enum {
  GT4HIP_RULE_DEFAULT = 0,
  GT4HIP_RULE_ADD = 1,
  GT4HIP_RULE_SUBTRACT = 2,
  GT4HIP_RULE_MIN = 3,
  GT4HIP_RULE_MAX = 4,
  GT4HIP_RULE_FIRST = 5,
  GT4HIP_RULE_SECOND = 6,
  GT4HIP_RULE_NUMBER = 7
};

/* the four outputs of compare_wordmaps, src/glistcompare.c:814-834; also the index order */
enum {
  GT4HIP_OP_UNION = 1,   /* index 0: <out>_<k>_union.list   */
  GT4HIP_OP_INTRSEC = 2, /* index 1: <out>_<k>_intrsec.list */
  GT4HIP_OP_DIFF1 = 4,   /* index 2: <out>_<k>_0_diff1.list */
  GT4HIP_OP_DIFF2 = 8    /* index 3: <out>_<k>_0_diff2.list */
};

typedef struct gt4hip_context gt4hip_context;
typedef struct gt4hip_list gt4hip_list;

/* ---------------------------------------------------------------- (i) init / teardown */

/* Binds a context to HIP device `device` (>= 0), creating its stream and workspace. */
int gt4hip_create (int device, gt4hip_context **ctx);
void gt4hip_destroy (gt4hip_context *ctx);
/* Message of the last failure on this context (or of the last failed gt4hip_create when ctx is
 * NULL).  Valid until the next call on the same context. */
const char *gt4hip_last_error (const gt4hip_context *ctx);
const char *gt4hip_strerror (int code);
/* Number of HIP devices visible; 0 when there is none or the runtime is unusable. */
int gt4hip_device_count (void);
/* The device a context lives on, and: give the context's pooled (freed, kept for reuse) blocks back to the driver. */
int gt4hip_context_device (const gt4hip_context *ctx);
int gt4hip_trim (gt4hip_context *ctx);
/* Free and total bytes of the context's device memory right now (hipMemGetInfo). */
int gt4hip_device_memory (gt4hip_context *ctx, uint64_t *free_bytes, uint64_t *total_bytes);
/* "name|gcnArch|CUs|HBM bytes" of the context's device, for logs. */
const char *gt4hip_device_info (const gt4hip_context *ctx);

/* ---------------------------------------------------------------- lists in HBM */

/* Copies n_words packed records from host memory (pageable or pinned) into a new HBM list.
 * Replaces gt4_word_map_new's mmap (src/word-map.c:165-241) as the way a list becomes readable. */
int gt4hip_list_upload (gt4hip_context *ctx, const void *host_records, uint64_t n_words,
                        uint32_t word_length, gt4hip_list **out);
/* The same from the k-mer table of a GT4I index file (what gt4_index_map_new exposes through the
 * sorted-list interface, reference src/index-map.c:123-175): n_words 16-byte (word, first location)
 * entries; the count of entry i is entry i+1's first location minus its own, the last one's
 * num_locations minus its own, truncated to 32 bits.  Decoded on the device. */
int gt4hip_list_upload_index (gt4hip_context *ctx, const void *host_kmers, uint64_t n_words, uint64_t num_locations,
                              uint32_t word_length, gt4hip_list **out);
/* Wraps records already in device memory (4-byte aligned, as the body of a .list file and every
 * gt4hip_list_slice are); the storage is not freed by gt4hip_list_free. */
int gt4hip_list_wrap (gt4hip_context *ctx, void *device_records, uint64_t n_words,
                      uint32_t word_length, gt4hip_list **out);
/* Uninitialised list with room for `capacity` records (n_words = capacity until set). */
int gt4hip_list_alloc (gt4hip_context *ctx, uint64_t capacity, uint32_t word_length, gt4hip_list **out);
/* A view of records [first, first+count) of `list` (shares storage; used for key-range shards). */
int gt4hip_list_slice (gt4hip_context *ctx, const gt4hip_list *list, uint64_t first, uint64_t count,
                       gt4hip_list **out);
/* File <-> HBM transfers (replace gt4_mmap + scout, src/utils.c:35-99, and the
 * fwrite / write output loops, src/glistcompare.c:491-496, :579-582).  The body of a list moves in
 * 8 MiB pieces through pinned staging buffers owned by the context (set up once, on first use) on
 * GT4HIP_IO_THREADS (default 8) copy threads, each with its own HIP stream: pread -> pinned -> HBM
 * and HBM -> pinned -> pwrite overlap piece by piece.  `fd` needs no particular file position.
 *   _upload_fd  new list from n_words records at byte `file_offset` of `fd`
 *   _load_fd    the same into an existing list (capacity >= n_words; sets n_words)
 *   _load       the same from host memory (large pageable buffers go through the staging threads)
 *   _write_fd   records [first, first+count) of `list` to `fd` at byte `file_offset` (pwrite: several
 *               writers -- threads or processes -- may fill disjoint extents of one file) */
int gt4hip_list_upload_fd (gt4hip_context *ctx, int fd, uint64_t file_offset, uint64_t n_words, uint32_t word_length,
                           gt4hip_list **out);
int gt4hip_list_load_fd (gt4hip_context *ctx, gt4hip_list *list, int fd, uint64_t file_offset, uint64_t n_words);
int gt4hip_list_load (gt4hip_context *ctx, gt4hip_list *list, const void *host_records, uint64_t n_words);
int gt4hip_list_write_fd (gt4hip_context *ctx, const gt4hip_list *list, uint64_t first, uint64_t count, int fd,
                          uint64_t file_offset);
/* Up to four lists to four files in one go (the outputs of one compare_wordmaps pass): the copy threads
 * are dealt to the files, because several writers of ONE file serialise on its inode lock. */
int gt4hip_lists_write_fd (gt4hip_context *ctx, uint32_t n, const gt4hip_list *const lists[], const uint64_t first[],
                           const uint64_t count[], const int fds[], const uint64_t file_offsets[]);
/* Copies the records back to host memory (n_words * 12 bytes). */
int gt4hip_list_download (gt4hip_context *ctx, const gt4hip_list *list, void *host_records);
/* Copies records [first, first+count) back to host memory. */
int gt4hip_list_download_range (gt4hip_context *ctx, const gt4hip_list *list, uint64_t first,
                                uint64_t count, void *host_records);
/* Lists return their storage to their context's pool: free every list BEFORE gt4hip_destroy of the
 * context it came from (a list must not outlive its context). */
void gt4hip_list_free (gt4hip_list *list);

uint64_t gt4hip_list_n_words (const gt4hip_list *list);      /* GT4WordSListInstance.num_words   */
uint32_t gt4hip_list_word_length (const gt4hip_list *list);  /* GT4WordSListInstance.word_length */
void *gt4hip_list_device_ptr (const gt4hip_list *list);
int gt4hip_list_set_n_words (gt4hip_list *list, uint64_t n_words);
/* Sum of counts (GT4WordSListInstance.sum_counts), computed on the device. */
int gt4hip_list_sum_counts (gt4hip_context *ctx, const gt4hip_list *list, uint64_t *sum);
/* 1 if keys are strictly ascending (the precondition of every merge below), else 0. */
int gt4hip_list_is_sorted (gt4hip_context *ctx, const gt4hip_list *list, int *sorted);
/* Index of the first record with key >= `key` (binary search on the device; shard splitters). */
int gt4hip_list_lower_bound (gt4hip_context *ctx, const gt4hip_list *list, uint64_t key, uint64_t *index);
/* GT4WordSArray get_word(idx), src/word-array-sorted.h:41-49 */
int gt4hip_list_get_word (gt4hip_context *ctx, const gt4hip_list *list, uint64_t idx, uint64_t *word,
                          uint32_t *count);

/* ---------------------------------------------------------------- (ii) pair operation */

/* Arguments of compare_wordmaps (src/glistcompare.c:789-790), minus the file naming. */
typedef struct {
  uint32_t ops;            /* mask of GT4HIP_OP_*: which of the four outputs to produce      */
  int32_t rule;            /* GT4HIP_RULE_*; DEFAULT resolves per output as the reference does */
  uint32_t cutoff;         /* -c / --cutoff (default 1)                                       */
  int32_t subtract;        /* -du: diff1 keeps keys with equal counts (src/glistcompare.c:480) */
  uint32_t count_override; /* the integer given to -r (RULE_NUMBER)                           */
  int32_t count_only;      /* --count_only: totals only, no records materialised              */
} gt4hip_compare_params;

typedef struct {
  uint64_t n_words[4];     /* records per output (header n_words), 0 for outputs not requested */
  uint64_t total_count[4]; /* sum of counts per output (header total_count)                    */
  /* In: NULL, or a caller-provided list (gt4hip_list_alloc) with enough capacity to receive the
   * output -- ZERO the struct before the call (memset) unless outputs are provided: a garbage
   * pointer here is taken for a caller's list.  Out: the output records (a new list when NULL was passed; NULL for count_only and
   * for outputs not requested).  Worst-case capacities: union na+nb, intrsec min(na,nb),
   * diff1 na, diff2 nb. */
  gt4hip_list *out[4];
  double merge_kernel_ms;  /* device time of the merge kernel alone (HIP events on the stream)  */
  double device_ms;        /* device time of everything the call enqueued                        */
  uint64_t merge_tiles;    /* tiles the merge kernel processed                                   */
} gt4hip_compare_result;

/* One pass over the merged key sequence of a and b producing up to four outputs
 * (compare_wordmaps hot loop, src/glistcompare.c:843-905). */
int gt4hip_compare (gt4hip_context *ctx, const gt4hip_list *a, const gt4hip_list *b,
                    const gt4hip_compare_params *params, gt4hip_compare_result *result);

/* ---------------------------------------------------------------- (iii) N-way operations */

typedef struct {
  uint64_t n_words;
  uint64_t total_count;
  gt4hip_list *out;        /* as gt4hip_compare_result.out: optional in, result out */
  double device_ms;
  uint64_t records_read;    /* records the pairwise merges of this call read (all levels) ...      */
  uint64_t records_written; /* ... and wrote: 12 bytes each, the HBM traffic the call really caused */
} gt4hip_multi_result;

/* union_multi, src/glistcompare.c:500-603: rule in {DEFAULT(=ADD), ADD, MAX, NUMBER}, cutoff
 * applied to the RESULTING count.  Empty lists are skipped. */
int gt4hip_union_multi (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n_lists,
                        uint32_t cutoff, int32_t rule, uint32_t count_override, int32_t count_only,
                        gt4hip_multi_result *result);
/* intersect_multi, src/glistcompare.c:605-717: rule in {DEFAULT(=MIN), MIN, MAX, ADD, NUMBER}. */
int gt4hip_intersect_multi (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n_lists,
                            uint32_t cutoff, int32_t rule, uint32_t count_override, int32_t count_only,
                            gt4hip_multi_result *result);

/* Per-key count table of an N-way union: for every distinct key ascending, counts[j] = count in
 * list j or 0 (what gt4_union hands to its callback, src/set-operations.c:161-179).  Up to 32 non-empty lists:
 * one launch of the N-way tile kernel (a ragged table, see below; nine and more lists: its 32-list instance -- 32 lists of
 * 2e7 entries, 3.4e8 rows: 13.6 ms against 540 ms by merges, profiles/round5); more than 32, option "kway_max" = 8 beyond
 * eight, or a ragged table that does not fit the device: by merges -- the N-way union gives the keys, and each column is
 * one more streaming merge of the key list with list j (rule SECOND keeps list j's count, absent keys get 0).
 * keys_out: n_keys u64; counts_out: n_keys * n_lists u32, row-major.  Both are device buffers
 * owned by the result; release with gt4hip_table_free. */
typedef struct {
  uint64_t n_keys;
  uint32_t n_lists;
  void *device_keys;
  void *device_counts;
  void *owner[2]; /* library-private: the pooled device blocks behind the two arrays */
  /* Non-NULL: the table is RAGGED.  gt4hip_union_table of up to 32 lists writes the table in ONE launch of the
   * N-way tile kernel: a tile of the merged key sequence puts its rows where its RECORDS start (it has at most as many
   * distinct keys as records), so the two arrays are allocated for the lists' records and hold unused rows behind
   * every tile's.  gt4hip_table_download gathers the rows asked for; gt4hip_table_compact turns the table into the
   * contiguous form (n_keys rows, ragged = NULL) for callers that read the device arrays themselves. */
  void *ragged;
} gt4hip_count_table;
int gt4hip_union_table (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n_lists,
                        gt4hip_count_table *table);
/* The same table restricted to the keys of lists[0] (what gt4_is_union walks,
 * src/set-operations.c:207-226): n_keys = n_words of lists[0], column 0 = its own counts. */
int gt4hip_probe_table (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n_lists,
                        gt4hip_count_table *table);
/* The same with `presence` != 0: column j holds 1 where list j contains the key and 0 where it does
 * not (a list may hold a key with count 0): what search_lists_multi needs beside the counts
 * (reference src/glistquery.c:776-812). */
int gt4hip_probe_table_ex (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n_lists, int presence,
                           gt4hip_count_table *table);
/* Makes a ragged table contiguous on the device (a gather into new arrays; a no-op for a contiguous one). */
int gt4hip_table_compact (gt4hip_context *ctx, gt4hip_count_table *table);
/* Copies rows [first, first+count) of the table to host memory. */
int gt4hip_table_download (gt4hip_context *ctx, const gt4hip_count_table *table, uint64_t first,
                           uint64_t count, uint64_t *host_keys, uint32_t *host_counts);
void gt4hip_table_free (gt4hip_count_table *table);

/* ---------------------------------------------------------------- (e) key-range shards across GPUs */

/* Every set operation above is key-local, so the key space can be cut into contiguous ranges: shard
 * g merges only the records of its range (cut out of every input with a lower_bound), and the shards'
 * outputs concatenated in shard order are the sorted result.  One process per GPU.
 * First key of shard g of n_shards equal-width ranges of the 4^word_length key space (2^64 for
 * k = 32); shard g covers [first(g), first(g+1)), the last one everything from first(n_shards-1). */
uint64_t gt4hip_shard_first_key (uint32_t word_length, uint32_t n_shards, uint32_t g);
/* SAMPLED cuts for lists resident in HBM (equal-width ranges balance only uniformly spread keys: the packed word IS
 * the sequence, reference src/sequence.c:116-130): first_keys[g], g < n_shards, is the first key of shard g
 * (first_keys[0] = 0) such that the shards hold about the same number of INPUT records of the n lists together
 * (every (total / 65536)-th key of every list, merged).  Deterministic: ranks holding the same lists get the same cuts. */
int gt4hip_shard_cuts (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t n, uint32_t n_shards, uint64_t *first_keys);

/* The exchange step: gatherv of the shards' records on `root` over RCCL (xGMI inside a node), as
 * grouped ncclSend / ncclRecv (RCCL has no native gatherv).  librccl.so is loaded by the first call
 * of this group only.  The communicator id is made by ONE rank (gt4hip_comm_unique_id) and handed
 * to the others by whatever the host has (shared memory after fork, a file, MPI, torch.distributed). */
#define GT4HIP_COMM_ID_BYTES 128
typedef struct gt4hip_comm gt4hip_comm;
int gt4hip_comm_unique_id (void *id_out);
int gt4hip_comm_create (gt4hip_context *ctx, const void *id, int n_ranks, int rank, gt4hip_comm **comm);
void gt4hip_comm_destroy (gt4hip_comm *comm);
int gt4hip_comm_rank (const gt4hip_comm *comm);
int gt4hip_comm_size (const gt4hip_comm *comm);
/* Message of the last failed gt4hip_comm_unique_id (no context yet at that point). */
const char *gt4hip_comm_last_error (void);
/* The other exchange of a sharded step: every rank's (n_words, total_count) to every rank (header totals, output
 * offsets): one ncclAllGather of two 64-bit words per rank on the library's stream, one synchronisation.
 * totals[2 r], totals[2 r + 1] = rank r's pair (2 * gt4hip_comm_size words). */
int gt4hip_comm_allgather_totals (gt4hip_comm *c, uint64_t n_words, uint64_t total_count, uint64_t *totals);
/* ... n <= 8 words per rank (the totals of every output of a pair operation at once): all[n r + i] = word i of rank r */
int gt4hip_comm_allgather_u64 (gt4hip_comm *c, const uint64_t *mine, uint32_t n, uint64_t *all);

/* counts[r] = records rank r contributes (every rank passes the same array: the all-gathered header
 * totals).  Rank r sends the first counts[r] records of `local`; on `root`, `gathered` (capacity >=
 * the sum) receives them in rank order and its n_words is set; other ranks pass NULL.  Blocks until
 * this rank's part is done. */
int gt4hip_comm_gatherv (gt4hip_comm *comm, const gt4hip_list *local, const uint64_t counts[], int root,
                         gt4hip_list *gathered);

/* ---------------------------------------------------------------- glistmaker's table step */

/* Sorts n_words packed 64-bit k-mer words (device memory) ascending, in place: the reference's
 * wordtable_sort (src/word-table.c:217-231; hybridInPlaceRadixSort256, src/utils.c:127-198) as an LSD
 * radix sort over the 2 * word_length significant bits (8- and 9-bit digits; one histogram kernel, then one
 * chained-scan scatter kernel per digit).  Needs n_words * 8 bytes of scratch + 4 KB of scan state per
 * 8192 words (0.5 GB per 10^9 words).  n_words < 2^56 and fewer than 2^32 tiles of 8192 words (n_words <
 * 2^45); more is GT4HIP_EINVAL.  The scans' waits are bounded (option "spin_limit"): one that gives up
 * -- a device shared with a stuck process -- makes the call return GT4HIP_EHIP. */
int gt4hip_sort_words (gt4hip_context *ctx, void *device_words, uint64_t n_words, uint32_t word_length);
/* Sort + wordtable_find_frequencies (src/word-table.c:233-260): host words (any order, repeats
 * allowed) -> a new list of (word, number of occurrences) records, ascending -- what glistmaker writes
 * to its temporary lists before gt4_write_union collates them (src/glistmaker.c:914-924, :333, :814). */
int gt4hip_words_to_list (gt4hip_context *ctx, const uint64_t *host_words, uint64_t n_words, uint32_t word_length,
                          gt4hip_list **out);
/* The same for words already in device memory.  The words are scratch from the call on: their buffer
 * holds the sorted words or the last pass's input afterwards, whichever the number of passes leaves. */
int gt4hip_device_words_to_list (gt4hip_context *ctx, void *device_words, uint64_t n_words, uint32_t word_length,
                                 gt4hip_list **out);

/* ---------------------------------------------------------------- synthetic lists (bench) */

/* Fills `list` (capacity >= n) with n strictly ascending keys < 4^word_length and counts in
 * [1, max_count]: record i gets key i*stride + (hash(seed,i) mod stride) with
 * stride = floor(keyspace / n); counts from hash(seed+1,i).  Same (seed, n, word_length) =>
 * same list on every device (tests regenerate it on the CPU). */
int gt4hip_generate (gt4hip_context *ctx, gt4hip_list *list, uint64_t n, uint64_t seed, uint32_t max_count);
/* General form: key_i = (i*stride + hash(key_seed,i) mod stride) * mult + add with
 * stride = floor(keyspace / mult / n) and add < mult, counts from hash(count_seed,i).  Lists made
 * with different `add` under one `mult` are disjoint; lists sharing key_seed share keys.  The
 * bench builds A = S u PA, B = S' u PB from three disjoint residue classes this way, which fixes
 * |A n B| = |S| exactly.  gt4hip_generate is (seed, seed+1, mult 1, add 0). */
int gt4hip_generate_ex (gt4hip_context *ctx, gt4hip_list *list, uint64_t n, uint64_t key_seed, uint64_t count_seed,
                        uint32_t max_count, uint64_t mult, uint64_t add);

/* Blocks until everything enqueued on the context's stream has finished. */
int gt4hip_synchronize (gt4hip_context *ctx);

/* Tuning / debugging knobs (not part of the reference surface):
 *   "two_pass" = 1     count + scan + write instead of the single-pass kernel
 *   "pool" = 0         release freed list storage to the driver instead of pooling it
 *   "pool_cap_mb" = n  most the pool may hold (default: half of the device memory); every device
 *                      allocation that fails gives the pooled blocks back and retries
 *   "grid" = n         workgroups of the merge kernel (0: one per resident slot)
 *   "kway" = 0 / 1 / 2 / 3  N-way unions of three and more lists by the pairwise tree of the pair kernel / by
 *                      the one-pass tile kernel (gt4hip_nway.hip) unless a probe of the keys or the tiles' own
 *                      samples show them clustered -- stretches of adjacent keys between wide gaps, which the
 *                      tile kernel orders two to three times slower: the tree is faster then (the default;
 *                      counter "kway_declined") / always by the tile kernel, two-list unions of the N-way
 *                      entry points too / always by the tile kernel (three lists and more); count tables
 *                      follow the same switch (and are never declined).
 *                      "kway_g": samples per tile of its first partition attempt; "kway_vt" (tests):
 *                      97 tile boundaries by searches over whole brackets, 98 every tile bucketed by
 *                      its pivot run, 99 every tile on the search path
 *   "spin_limit" = n   bound of the single-pass kernel's inter-workgroup waits (0: default, ~seconds)
 *   "dynamic" = 1 / -1 tiles of the single-pass kernel always / never dealt by a ticket counter
 *                      (0: automatic -- the record-writing kernels except a complement alone)
 *   "scan_group" = 1 / -1  the scanner as a group of wavefronts always / never (0: by launch size)
 *   "geom0" / "geom1"  force the 512- / 1024-thread geometry (experiments). */
int gt4hip_set_option (gt4hip_context *ctx, const char *name, int64_t value);
/* Diagnostic counters of a context.  "single_pass_fallbacks": calls whose single-pass merge gave up a
 * bounded wait (a worker not resident: shared device) and were rerun on the two-pass path;
 * "kway_declined": N-way unions handed to the pairwise tree because their keys are clustered;
 * "kway_calls" / "kway_overflows": N-way unions (and count tables) done by the one-pass tile kernel /
 * partitions repeated with fewer samples per tile because a tile would not have fit LDS;
 * "nway_kernel_us", "nway_tiles": the last N-way call's tile kernel; "nway_one_pass": 1 when the last
 * gt4hip_union_multi took the one-pass tile kernel, 0 when it took the pairwise tree; "sort_us", "fold_us", "table_us":
 * the last gt4hip_device_words_to_list / gt4hip_union_table call. */
int gt4hip_get_counter (gt4hip_context *ctx, const char *name, uint64_t *value);

#ifdef __cplusplus
}
#endif
#endif
