/*
 * gt4_listfile.c -- host-side `.list` file access for the GPU glistcompare path (see
 * include/gt4_listfile.h).  Plain C; no GPU code here.
 */
#define _GNU_SOURCE
#include "gt4_listfile.h"

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

void gt4_list_header_init (GT4ListHeader *hdr, unsigned int word_length)
{
  memset (hdr, 0, sizeof *hdr);
  hdr->code = GT4_LIST_CODE_VALUE;
  hdr->version_major = GT4_VERSION_MAJOR;
  hdr->version_minor = GT4_VERSION_MINOR;
  hdr->word_length = word_length;
  hdr->list_start = sizeof (GT4ListHeader);
  hdr->word_bytes = 8;
  hdr->count_bytes = 4;
}

int gt4_listfile_sniff (const char *path, uint32_t *code)
{
  int fd = open (path, O_RDONLY);
  if (fd < 0) return 1;
  uint32_t c = 0;
  ssize_t got = read (fd, &c, 4);
  close (fd);
  if (got != 4) return 1;
  *code = c;
  return 0;
}

int gt4_listfile_open (const char *path, unsigned int major_version, GT4ListFile *out)
{
  memset (out, 0, sizeof *out);
  int fd = open (path, O_RDONLY);
  struct stat st;
  if (fd < 0 || fstat (fd, &st) < 0) {
    if (fd >= 0) close (fd);
    fprintf (stderr, "gt4_word_map_new: could not mmap file %s\n", path);
    return GT4_LISTFILE_EOPEN;
  }
  const uint64_t size = (uint64_t) st.st_size;
  const unsigned char *map = NULL;
  if (size) {
    map = (const unsigned char *) mmap (NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) map = NULL;
  }
  close (fd);
  if (!map || size < 16) {
    if (map) munmap ((void *) map, size);
    fprintf (stderr, "gt4_word_map_new: could not mmap file %s\n", path);
    return GT4_LISTFILE_EOPEN;
  }
  /* the first 16 bytes are common to every header version */
  uint32_t code, major, minor;
  memcpy (&code, map, 4);
  memcpy (&major, map + 4, 4);
  memcpy (&minor, map + 8, 4);
  if (code != GT4_LIST_CODE_VALUE) {
    fprintf (stderr, "gt4_word_map_new: invalid file tag (%x, should be %x)\n", code, GT4_LIST_CODE_VALUE);
    munmap ((void *) map, size);
    return GT4_LISTFILE_EMAGIC;
  }
  if (major != major_version) {
    fprintf (stderr, "gt4_word_map_new: incompatible major version %u (required %u)\n", major, major_version);
    munmap ((void *) map, size);
    return GT4_LISTFILE_EVERSION;
  }
  GT4ListHeader h;
  memset (&h, 0, sizeof h);
  const size_t avail = size < sizeof h ? (size_t) size : sizeof h;
  if (minor == 0) {
    /* 4.0: 40-byte header, records right after it */
    memcpy (&h, map, avail < 40 ? avail : 40);
    h.list_start = 40;
    h.word_bytes = 8;
    h.count_bytes = 4;
  } else if (minor <= 2) {
    /* 4.1 / 4.2: list_start from the file, 8 + 4 byte records implied */
    memcpy (&h, map, avail < 40 ? avail : 40);
    h.word_bytes = 8;
    h.count_bytes = 4;
  } else {
    memcpy (&h, map, avail);
  }
  /* The reference's own test (src/word-map.c:211-215) multiplies in wrapping 64-bit arithmetic by the FILE's
   * word_bytes + count_bytes (minor > 2) while every reader -- its own included, src/word-map.h:89-99 -- strides 12:
   * a header with n_words = 2^40 and 0 + 0 bytes per record, or with n_words = 2^64 / 12 + 1, passes it and is then
   * read far beyond the mapping.  Here: the reference's test first (same message, same number, where it fires), then
   * the bound that actually protects the readers, by division: list_start inside the file and n_words records of 12
   * bytes behind it. */
  const uint64_t ref_need = h.list_start + h.n_words * (uint64_t) (h.word_bytes + h.count_bytes);
  const int fits = h.list_start <= size && h.n_words <= (size - h.list_start) / 12u;
  if (size < ref_need || !fits) {
    uint64_t need = ref_need;
    if (size >= ref_need) { /* (the reference would have gone on: say what the readers need, saturating) */
      need = h.n_words > (UINT64_MAX - h.list_start) / 12u ? UINT64_MAX : h.list_start + 12u * h.n_words;
      if (h.list_start > size && need < h.list_start) need = h.list_start;
    }
    fprintf (stderr, "gt4_word_map_new: file size too small (%llu, should be at least %llu)\n", (unsigned long long) size,
             (unsigned long long) need);
    munmap ((void *) map, size);
    return GT4_LISTFILE_ESIZE;
  }
  out->filename = strdup (path);
  out->file_map = map;
  out->file_size = size;
  out->header = h;
  out->records = map + h.list_start;
  return GT4_LISTFILE_OK;
}

int gt4_indexfile_open (const char *path, unsigned int major_version, GT4ListFile *out)
{
  memset (out, 0, sizeof *out);
  int fd = open (path, O_RDONLY);
  struct stat st;
  if (fd < 0 || fstat (fd, &st) < 0) {
    if (fd >= 0) close (fd);
    fprintf (stderr, "gt4_index_map_new: could not mmap file %s\n", path);
    return GT4_LISTFILE_EOPEN;
  }
  const uint64_t size = (uint64_t) st.st_size;
  const unsigned char *map = NULL;
  if (size) {
    map = (const unsigned char *) mmap (NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) map = NULL;
  }
  close (fd);
  if (!map || size < sizeof (GT4IndexHeader)) {
    if (map) munmap ((void *) map, size);
    fprintf (stderr, "gt4_index_map_new: could not mmap file %s\n", path);
    return GT4_LISTFILE_EOPEN;
  }
  GT4IndexHeader h;
  memcpy (&h, map, sizeof h);
  if (h.code != GT4_INDEX_CODE_VALUE) {
    fprintf (stderr, "gt4_index_map_new: invalid file tag (%x, should be %x)\n", h.code, GT4_INDEX_CODE_VALUE);
    munmap ((void *) map, size);
    return GT4_LISTFILE_EMAGIC;
  }
  if (h.version_major != major_version) {
    fprintf (stderr, "gt4_index_map_new: incompatible major version %u (required %u)\n", h.version_major, major_version);
    munmap ((void *) map, size);
    return GT4_LISTFILE_EVERSION;
  }
  if (h.kmers_start > size || h.num_words > (size - h.kmers_start) / 16) {
    fprintf (stderr, "gt4_index_map_new: file size too small (%llu) for %llu k-mers at %llu\n", (unsigned long long) size,
             (unsigned long long) h.num_words, (unsigned long long) h.kmers_start);
    munmap ((void *) map, size);
    return GT4_LISTFILE_ESIZE;
  }
  out->filename = strdup (path);
  out->file_map = map;
  out->file_size = size;
  gt4_list_header_init (&out->header, h.word_length);
  out->header.n_words = h.num_words;
  out->header.total_count = h.num_locations;
  out->records = NULL;
  out->index_kmers = map + h.kmers_start;
  out->index_locations = h.num_locations;
  return GT4_LISTFILE_OK;
}

void gt4_listfile_close (GT4ListFile *lf)
{
  if (!lf) return;
  if (lf->file_map) munmap ((void *) lf->file_map, lf->file_size);
  free (lf->filename);
  memset (lf, 0, sizeof *lf);
}

static int write_all (int fd, const void *buf, size_t len)
{
  const char *p = (const char *) buf;
  while (len) {
    ssize_t w = write (fd, p, len);
    if (w < 0) {
      if (errno == EINTR) continue;
      return 1;
    }
    p += w;
    len -= (size_t) w;
  }
  return 0;
}

int gt4_listwriter_begin (GT4ListWriter *w, const char *path, unsigned int word_length, unsigned int mode)
{
  gt4_list_header_init (&w->header, word_length);
  w->fd = open (path, O_WRONLY | O_CREAT | O_TRUNC, (mode_t) mode);
  if (w->fd < 0) return 1;
  return write_all (w->fd, &w->header, sizeof w->header);
}

int gt4_listwriter_append (GT4ListWriter *w, const void *records, uint64_t n)
{
  return write_all (w->fd, records, (size_t) n * 12u);
}

int gt4_listwriter_finish (GT4ListWriter *w, uint64_t n_words, uint64_t total_count)
{
  w->header.n_words = n_words;
  w->header.total_count = total_count;
  int bad = pwrite (w->fd, &w->header, sizeof w->header, 0) != (ssize_t) sizeof w->header;
  bad |= close (w->fd) != 0;
  w->fd = -1;
  return bad;
}

void gt4_listwriter_abort (GT4ListWriter *w)
{
  if (w->fd >= 0) close (w->fd);
  w->fd = -1;
}
