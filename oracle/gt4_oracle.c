/*
 * gt4_oracle.c -- TEST INFRASTRUCTURE ONLY (see gt4_oracle.h).
 *
 * Scalar restatement of the reference merge loops over packed in-memory record
 * arrays.  Every function names the reference lines it restates.  Nothing in
 * the product path may call into this file.
 */
#include "gt4_oracle.h"

#include <string.h>

#define REC GT4O_RECORD_BYTES
#define ALL_ONES 0xffffffffffffffffULL

/* record i of a packed list: key at +0, count at +8 (reference src/word-map.h:89-99) */
static inline uint64_t rec_key (const uint8_t *l, uint64_t i)
{
  uint64_t k;
  memcpy (&k, l + REC * i, 8);
  return k;
}

static inline uint32_t rec_cnt (const uint8_t *l, uint64_t i)
{
  uint32_t c;
  memcpy (&c, l + REC * i + 8, 4);
  return c;
}

/* one output sink: optional record buffer + running header totals */
typedef struct {
  uint8_t *buf;
  uint64_t n;
  uint64_t sum;
} sink;

static inline void sink_put (sink *s, uint64_t key, uint32_t freq)
{
  if (s->buf) {
    memcpy (s->buf + REC * s->n, &key, 8);
    memcpy (s->buf + REC * s->n + 8, &freq, 4);
  }
  s->n += 1;
  s->sum += freq;
}

void gt4o_header_init (gt4o_header *h, uint32_t word_length)
{
  /* src/word-list.c:33-44 with VERSION 4.2 (src/version.h:27-28) */
  memset (h, 0, sizeof *h);
  h->code = 0x47543443u; /* 'G'<<24|'T'<<16|'4'<<8|'C', src/word-list.c:31 */
  h->version_major = 4;
  h->version_minor = 2;
  h->word_length = word_length;
  h->list_start = 48;
  h->word_bytes = 8;
  h->count_bytes = 4;
}

int gt4o_header_parse (const uint8_t *file, uint64_t file_size, gt4o_header *out)
{
  gt4o_header raw;
  memset (&raw, 0, sizeof raw);
  memcpy (&raw, file, file_size < 48 ? file_size : 48);
  if (raw.code != 0x47543443u) return 1;       /* src/word-map.c:181 */
  if (raw.version_major != 4) return 2;        /* src/word-map.c:185 */
  memset (out, 0, sizeof *out);
  if (raw.version_minor == 0) {                /* src/word-map.c:198-202 */
    memcpy (out, &raw, 40);
    out->list_start = 40;
    out->word_bytes = 8;
    out->count_bytes = 4;
  } else if (raw.version_minor <= 2) {         /* src/word-map.c:203-206 */
    memcpy (out, &raw, 40);
    out->word_bytes = 8;
    out->count_bytes = 4;
  } else {                                     /* src/word-map.c:207-209 */
    memcpy (out, &raw, 48);
  }
  /* src/word-map.c:211-215 */
  if (file_size < out->list_start + out->n_words * (uint64_t) (out->word_bytes + out->count_bytes)) return 3;
  return 0;
}

int gt4o_index_decode (const uint8_t *file, uint64_t file_size, uint32_t *word_length, uint64_t *num_words,
                       uint64_t *num_locations, uint8_t *records)
{
  /* struct _GT4IndexHeader, src/index-map.h:69-83 (72 bytes, little-endian) */
  uint32_t code, major, wlen;
  uint64_t nw, nloc, kmers_start;
  if (file_size < 72) return 3;
  memcpy (&code, file, 4);
  memcpy (&major, file + 4, 4);
  memcpy (&wlen, file + 12, 4);
  memcpy (&nw, file + 16, 8);
  memcpy (&nloc, file + 24, 8);
  memcpy (&kmers_start, file + 56, 8);
  if (code != 0x47543449u) return 1;           /* src/index-map.c:339 */
  if (major != 4) return 2;                    /* src/index-map.c:344 */
  if (kmers_start > file_size || nw > (file_size - kmers_start) / 16) return 3;
  *word_length = wlen;
  *num_words = nw;
  *num_locations = nloc;
  if (!records) return 0;
  const uint8_t *k = file + kmers_start;
  for (uint64_t i = 0; i < nw; i++) {
    uint64_t word, loc, next;
    memcpy (&word, k + 16 * i, 8);             /* imap_get_word, :123-127 */
    memcpy (&loc, k + 16 * i + 8, 8);
    if (i + 1 == nw) next = nloc;              /* imap_get_count, :129-139 */
    else memcpy (&next, k + 16 * (i + 1) + 8, 8);
    const uint32_t count = (uint32_t) (next - loc);
    memcpy (records + 12 * i, &word, 8);
    memcpy (records + 12 * i + 8, &count, 4);
  }
  return 0;
}

uint32_t gt4o_calculate_freq (uint32_t f1, uint32_t f2, int rule, uint32_t count_override)
{
  /* src/glistcompare.c:433-455 */
  if (rule == GT4O_RULE_ADD) return f1 + f2;
  if (rule == GT4O_RULE_SUBTRACT) return f1 > f2 ? f1 - f2 : 0;
  if (rule == GT4O_RULE_MIN) return f1 < f2 ? f1 : f2;
  if (rule == GT4O_RULE_MAX) return f1 > f2 ? f1 : f2;
  if (rule == GT4O_RULE_FIRST) return f1;
  if (rule == GT4O_RULE_SECOND) return f2;
  if (rule == GT4O_RULE_NUMBER) return count_override;
  return 0;
}

/* src/glistcompare.c:459-466 */
static int union_pred (uint32_t f1, uint32_t f2, int rule, uint32_t cutoff, uint32_t ovr, uint32_t *freq)
{
  if (f1 < cutoff && f2 < cutoff) return 0;
  *freq = gt4o_calculate_freq (f1, f2, rule == GT4O_RULE_DEFAULT ? GT4O_RULE_ADD : rule, ovr);
  return *freq != 0;
}

/* src/glistcompare.c:468-475 */
static int intrsec_pred (uint32_t f1, uint32_t f2, int rule, uint32_t cutoff, uint32_t ovr, uint32_t *freq)
{
  if (f1 < cutoff || f2 < cutoff) return 0;
  *freq = gt4o_calculate_freq (f1, f2, rule == GT4O_RULE_DEFAULT ? GT4O_RULE_MIN : rule, ovr);
  return *freq != 0;
}

/* src/glistcompare.c:477-489 */
static int complement_pred (uint32_t f1, uint32_t f2, int rule, uint32_t cutoff, uint32_t ovr, int subtract, uint32_t *freq)
{
  if (subtract) {
    if (f1 != f2 || f1 < cutoff) return 0;
    *freq = f1;
    return 1;
  }
  if (f1 < cutoff || f2 >= cutoff) return 0;
  *freq = gt4o_calculate_freq (f1, f2, rule == GT4O_RULE_DEFAULT ? GT4O_RULE_SUBTRACT : rule, ovr);
  return *freq != 0;
}

int gt4o_compare (const uint8_t *a, uint64_t na, const uint8_t *b, uint64_t nb,
                  unsigned ops, int rule, uint32_t cutoff, int subtract,
                  uint32_t ovr, uint8_t *out[4], gt4o_stat stat[4])
{
  /* src/glistcompare.c:843-905: one pass over the merged key sequence */
  sink s[4];
  uint64_t i = 0, j = 0;
  int want_u = (ops & GT4O_OP_UNION) != 0, want_i = (ops & GT4O_OP_INTRSEC) != 0;
  int want_d = (ops & GT4O_OP_DIFF1) != 0, want_dd = (ops & GT4O_OP_DIFF2) != 0;
  for (int k = 0; k < 4; k++) {
    s[k].buf = out ? out[k] : 0;
    s[k].n = 0;
    s[k].sum = 0;
  }
  while (i < na || j < nb) {
    uint32_t freq = 0;
    int have_a = i < na, have_b = j < nb;
    uint64_t ka = have_a ? rec_key (a, i) : 0, kb = have_b ? rec_key (b, j) : 0;
    if (have_a && have_b && ka == kb) {
      /* :845-872 key in both lists */
      uint32_t f1 = rec_cnt (a, i), f2 = rec_cnt (b, j);
      if (want_u && union_pred (f1, f2, rule, cutoff, ovr, &freq)) sink_put (&s[0], ka, freq);
      if (want_i && intrsec_pred (f1, f2, rule, cutoff, ovr, &freq)) sink_put (&s[1], ka, freq);
      if (want_d && complement_pred (f1, f2, rule, cutoff, ovr, subtract, &freq)) sink_put (&s[2], ka, freq);
      if (want_dd && complement_pred (f2, f1, rule, cutoff, ovr, 0, &freq)) sink_put (&s[3], kb, freq);
      i++;
      j++;
    } else if (have_a && (!have_b || ka < kb)) {
      /* :873-888 key only in the first list */
      uint32_t f1 = rec_cnt (a, i);
      if (want_u && union_pred (f1, 0, rule, cutoff, ovr, &freq)) sink_put (&s[0], ka, freq);
      if (want_d && complement_pred (f1, 0, rule, cutoff, ovr, subtract, &freq)) sink_put (&s[2], ka, freq);
      i++;
    } else {
      /* :889-904 key only in the second list */
      uint32_t f2 = rec_cnt (b, j);
      if (want_u && union_pred (0, f2, rule, cutoff, ovr, &freq)) sink_put (&s[0], kb, freq);
      if (want_dd && complement_pred (f2, 0, rule, cutoff, ovr, 0, &freq)) sink_put (&s[3], kb, freq);
      j++;
    }
  }
  for (int k = 0; k < 4; k++) {
    int wanted = (ops >> k) & 1;
    stat[k].n_words = wanted ? s[k].n : 0;
    stat[k].total_count = wanted ? s[k].sum : 0;
  }
  return 0;
}

#define MAX_LISTS 4096 /* GT4_MAX_SETS, src/set-operations.h:29 */

/* shared N-way union walk: src/glistcompare.c:545-591 and src/set-operations.c:77-116.
 * A flat scan of every live head per distinct key (the reference keeps no heap). */
static void nway_union (const uint8_t *const lists[], const uint64_t n[], unsigned n_lists,
                        uint32_t cutoff, int rule, uint32_t ovr, sink *s)
{
  static __thread uint64_t pos[MAX_LISTS];
  unsigned live = 0;
  uint64_t word = ALL_ONES;
  for (unsigned j = 0; j < n_lists; j++) {
    pos[j] = 0;
    if (n[j]) {
      live++;
      if (rec_key (lists[j], 0) < word) word = rec_key (lists[j], 0);
    }
  }
  while (live) {
    uint64_t next = ALL_ONES;
    uint32_t freq = 0;
    for (unsigned j = 0; j < n_lists; j++) {
      if (pos[j] >= n[j]) continue;
      if (rec_key (lists[j], pos[j]) == word) {
        uint32_t c = rec_cnt (lists[j], pos[j]);
        if (rule == GT4O_RULE_ADD) freq += c;
        else if (rule == GT4O_RULE_MAX) { if (c > freq) freq = c; }
        else freq = ovr;
        pos[j]++;
        if (pos[j] >= n[j]) {
          live--;
          continue;
        }
      }
      if (rec_key (lists[j], pos[j]) < next) next = rec_key (lists[j], pos[j]);
    }
    if (freq >= cutoff) sink_put (s, word, freq); /* :574, cutoff on the RESULT */
    word = next;
  }
}

int gt4o_union_multi (const uint8_t *const lists[], const uint64_t n[], unsigned n_lists,
                      uint32_t cutoff, int rule, uint32_t ovr, uint8_t *out, gt4o_stat *stat)
{
  sink s = { out, 0, 0 };
  /* src/glistcompare.c:518-523 */
  if (rule == GT4O_RULE_DEFAULT) rule = GT4O_RULE_ADD;
  else if (rule != GT4O_RULE_ADD && rule != GT4O_RULE_MAX && rule != GT4O_RULE_NUMBER) return 1;
  if (n_lists > MAX_LISTS) return 1;
  nway_union (lists, n, n_lists, cutoff, rule, ovr, &s);
  stat->n_words = s.n;
  stat->total_count = s.sum;
  return 0;
}

int gt4o_write_union (const uint8_t *const lists[], const uint64_t n[], unsigned n_lists,
                      uint32_t cutoff, uint8_t *out, gt4o_stat *stat)
{
  sink s = { out, 0, 0 };
  /* src/set-operations.c:49-50 */
  if (n_lists == 0 || n_lists > MAX_LISTS) return 1;
  nway_union (lists, n, n_lists, cutoff, GT4O_RULE_ADD, 0, &s);
  stat->n_words = s.n;
  stat->total_count = s.sum;
  return 0;
}

int gt4o_intersect_multi (const uint8_t *const lists[], const uint64_t n[], unsigned n_lists,
                          uint32_t cutoff, int rule, uint32_t ovr, uint8_t *out, gt4o_stat *stat)
{
  static __thread uint64_t pos[MAX_LISTS];
  sink s = { out, 0, 0 };
  int done = 0;
  uint64_t word = 0;
  /* src/glistcompare.c:622-627 */
  if (rule == GT4O_RULE_DEFAULT) rule = GT4O_RULE_MIN;
  else if (rule != GT4O_RULE_ADD && rule != GT4O_RULE_MIN && rule != GT4O_RULE_MAX && rule != GT4O_RULE_NUMBER) return 1;
  if (n_lists > MAX_LISTS) return 1;
  stat->n_words = 0;
  stat->total_count = 0;
  for (unsigned j = 0; j < n_lists; j++) {
    pos[j] = 0;
    if (!n[j]) done = 1; /* :633-636 any empty list empties the result */
  }
  while (!done) {
    uint32_t freq = 0;
    unsigned n_equal = 0;
    /* :651-653 raise the candidate to the largest head */
    for (unsigned j = 0; j < n_lists; j++) {
      uint64_t k = rec_key (lists[j], pos[j]);
      if (k > word) word = k;
    }
    /* :655-678 advance laggards; restart when a head overshoots */
    for (unsigned j = 0; j < n_lists && !done; j++) {
      while (rec_key (lists[j], pos[j]) < word) {
        if (++pos[j] >= n[j]) {
          done = 1;
          break;
        }
      }
      if (done) break;
      if (rec_key (lists[j], pos[j]) > word) {
        word = rec_key (lists[j], pos[j]);
        break;
      }
      n_equal++;
      uint32_t c = rec_cnt (lists[j], pos[j]);
      if (rule == GT4O_RULE_MIN) { if (!freq || c < freq) freq = c; } /* :669 */
      else if (rule == GT4O_RULE_MAX) { if (c > freq) freq = c; }
      else if (rule == GT4O_RULE_ADD) freq += c;
      else freq = ovr;
    }
    if (done) break;
    if (n_equal == n_lists) {
      if (freq >= cutoff) sink_put (&s, word, freq); /* :682 */
      /* :696-703 step every list past the emitted key */
      for (unsigned j = 0; j < n_lists; j++) {
        if (++pos[j] >= n[j]) {
          done = 1;
          break;
        }
        if (rec_key (lists[j], pos[j]) > word) word = rec_key (lists[j], pos[j]);
      }
    }
  }
  stat->n_words = s.n;
  stat->total_count = s.sum;
  return 0;
}

unsigned int gt4o_union (const uint8_t *const lists[], const uint64_t n[], unsigned n_lists,
                         gt4o_callback cb, void *data)
{
  /* src/set-operations.c:153-180 */
  static __thread uint64_t pos[MAX_LISTS];
  static __thread uint32_t counts[MAX_LISTS];
  unsigned live = 0;
  uint64_t word = ALL_ONES;
  if (n_lists == 0 || n_lists > MAX_LISTS) return 1;
  for (unsigned j = 0; j < n_lists; j++) {
    pos[j] = 0;
    if (n[j]) {
      live++;
      if (rec_key (lists[j], 0) < word) word = rec_key (lists[j], 0);
    }
  }
  while (live) {
    uint64_t next = ALL_ONES;
    for (unsigned j = 0; j < n_lists; j++) {
      counts[j] = 0;
      if (pos[j] >= n[j]) continue;
      if (rec_key (lists[j], pos[j]) == word) {
        counts[j] = rec_cnt (lists[j], pos[j]);
        pos[j]++;
        if (pos[j] >= n[j]) {
          /* :166-170 the exhausted iterator keeps its last word (src/word-list-sorted.c:73-75),
           * which still feeds `next`: while other lists remain the reference therefore
           * re-visits `word` once more with all-zero counts.  Verified against
           * oracle/_ref/ref_setops; restated as-is. */
          live--;
          if (word < next) next = word;
          continue;
        }
      }
      if (rec_key (lists[j], pos[j]) < next) next = rec_key (lists[j], pos[j]);
    }
    unsigned int r = cb (word, counts, data);
    if (r) return r;
    word = next;
  }
  return 0;
}

unsigned int gt4o_is_union (const uint8_t *const lists[], const uint64_t n[], unsigned n_lists,
                            gt4o_callback cb, void *data)
{
  /* src/set-operations.c:207-226: walk list 0, probe the others */
  static __thread uint64_t pos[MAX_LISTS];
  static __thread uint32_t counts[MAX_LISTS];
  if (n_lists == 0 || n_lists > MAX_LISTS) return 1;
  for (unsigned j = 0; j < n_lists; j++) pos[j] = 0;
  for (uint64_t i = 0; i < n[0]; i++) {
    uint64_t word = rec_key (lists[0], i);
    counts[0] = rec_cnt (lists[0], i);
    for (unsigned j = 1; j < n_lists; j++) {
      counts[j] = 0;
      while (pos[j] < n[j] && rec_key (lists[j], pos[j]) < word) pos[j]++;
      if (pos[j] < n[j] && rec_key (lists[j], pos[j]) == word) counts[j] = rec_cnt (lists[j], pos[j]);
    }
    unsigned int r = cb (word, counts, data);
    if (r) return r;
  }
  return 0;
}
