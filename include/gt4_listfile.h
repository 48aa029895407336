/*
 * gt4_listfile.h -- GenomeTester4 `.list` files on the host (C).
 *
 * Drop-in for the file-format side of the hot path: the on-disk header of
 * reference src/word-list.h:40-72, its initialiser (src/word-list.c:33-44) and the validating
 * mmap reader of gt4_word_map_new (src/word-map.c:165-241).  A file is a 48-byte (v4.0: 40-byte)
 * little-endian header followed by packed 12-byte records (u64 key + u32 count), ascending by key.
 */
#ifndef GT4_LISTFILE_H
#define GT4_LISTFILE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 'G'<<24 | 'T'<<16 | '4'<<8 | 'C' and ...'I' (src/word-list.c:31, src/word-index.h) */
#define GT4_LIST_CODE_VALUE 0x47543443u
#define GT4_INDEX_CODE_VALUE 0x47543449u

#define GT4_VERSION_MAJOR 4
#define GT4_VERSION_MINOR 2
#define GT4_VERSION_MICRO 16
#define GT4_VERSION_QUALIFIER "stable"

/* struct _GT4ListHeader_4_4, src/word-list.h:61-72: 48 bytes, no padding */
typedef struct {
  uint32_t code;
  uint32_t version_major;
  uint32_t version_minor;
  uint32_t word_length;
  uint64_t n_words;
  uint64_t total_count;
  uint64_t list_start;
  uint32_t word_bytes;
  uint32_t count_bytes;
} GT4ListHeader;

/* gt4_list_header_init, src/word-list.c:33-44: zero, magic, version 4.2, list_start 48, 8 + 4 */
void gt4_list_header_init (GT4ListHeader *hdr, unsigned int word_length);

/* A list file mapped read-only (the GT4WordMap of the GPU path: records are then uploaded). */
typedef struct {
  char *filename;
  const unsigned char *file_map; /* whole file, PROT_READ / MAP_PRIVATE (src/utils.c:54) */
  uint64_t file_size;
  GT4ListHeader header;          /* normalised as src/word-map.c:198-209 */
  const unsigned char *records;  /* file_map + header.list_start; NULL for a GT4I index file */
  /* GT4I index files (gt4_indexfile_open): the k-mer table of 16-byte (word, first location) entries;
   * the count of entry i is the distance to the next entry's first location, the last one's to
   * index_locations (imap_get_count, src/index-map.c:129-139).  NULL / 0 for list files. */
  const unsigned char *index_kmers;
  uint64_t index_locations;
} GT4ListFile;

/* struct _GT4IndexHeader, src/index-map.h:69-83: 72 bytes, little-endian, no padding */
typedef struct {
  uint32_t code;
  uint32_t version_major;
  uint32_t version_minor;
  uint32_t word_length;
  uint64_t num_words;
  uint64_t num_locations;
  uint32_t n_file_bits;
  uint32_t n_subseq_bits;
  uint32_t n_pos_bits;
  uint32_t filler0;
  uint64_t files_start;
  uint64_t kmers_start;
  uint64_t locations_start;
} GT4IndexHeader;

enum {
  GT4_LISTFILE_OK = 0,
  GT4_LISTFILE_EOPEN = 1,    /* cannot open / map                                   */
  GT4_LISTFILE_EMAGIC = 2,   /* invalid file tag (src/word-map.c:181)               */
  GT4_LISTFILE_EVERSION = 3, /* incompatible major version (src/word-map.c:185)     */
  GT4_LISTFILE_ESIZE = 4     /* file size too small (src/word-map.c:211-215) -- and headers that pass that test while
                              * the records cannot lie in the file at stride 12: list_start > size or
                              * n_words > (size - list_start) / 12, checked by division (the reference's product wraps) */
};

/* Reads the 4-byte tag of a file (glistcompare's format sniff, src/glistcompare.c:256-263).
 * Returns 0 and the tag, or 1 when the file cannot be opened / is shorter than 4 bytes. */
int gt4_listfile_sniff (const char *path, uint32_t *code);

/* Maps and validates `path`.  On failure prints the reference's diagnostic to stderr
 * ("gt4_word_map_new: ...") and returns a GT4_LISTFILE_E* code. */
int gt4_listfile_open (const char *path, unsigned int major_version, GT4ListFile *out);
void gt4_listfile_close (GT4ListFile *lf);

/* Maps a GT4I index file (gt4_index_map_new, src/index-map.c:317-373) as a sorted k-mer list: fills
 * header.word_length / n_words (= num_words) / total_count (= num_locations), index_kmers and
 * index_locations; records stays NULL.  The reference checks tag and major version only
 * ("gt4_index_map_new: ..." diagnostics, reproduced); a k-mer table that does not fit in the file
 * is rejected here with GT4_LISTFILE_ESIZE instead of being read out of bounds. */
int gt4_indexfile_open (const char *path, unsigned int major_version, GT4ListFile *out);

/* Incremental writer: placeholder header, records, back-patched header (the reference's
 * fopen/fwrite/fseek sequence, src/glistcompare.c:816-834, :907-915, or write/pwrite, :538-595). */
typedef struct {
  int fd;
  GT4ListHeader header;
} GT4ListWriter;

/* Creates (truncates) `path` with `mode` and writes the placeholder header.  Returns 0 / 1. */
int gt4_listwriter_begin (GT4ListWriter *w, const char *path, unsigned int word_length, unsigned int mode);
/* Appends n packed records (does not touch the header totals).  Returns 0 / 1. */
int gt4_listwriter_append (GT4ListWriter *w, const void *records, uint64_t n);
/* Back-patches n_words / total_count and closes.  Returns 0 / 1. */
int gt4_listwriter_finish (GT4ListWriter *w, uint64_t n_words, uint64_t total_count);
/* Closes without finishing (caller unlinks). */
void gt4_listwriter_abort (GT4ListWriter *w);

#ifdef __cplusplus
}
#endif
#endif
