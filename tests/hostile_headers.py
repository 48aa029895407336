"""Hostile `.list` / GT4I headers (Missing 4): files whose header passes the reference's own size test
(`/root/reference/src/word-map.c:211-215`: a wrapping 64-bit product with the FILE's word_bytes + count_bytes) or
defeats it, while every reader strides 12 (`src/word-map.h:89-99`).  The reference crashes on several of them; the
product must refuse each with "file size too small" and exit code 1, without a signal and without a sanitizer report.

`cases()` -> [(file name, bytes, substring the refusal must contain)]."""
import struct

LIST_CODE = 0x47543443
INDEX_CODE = 0x47543449


def list_header(minor=2, k=16, n_words=0, total=0, list_start=48, word_bytes=8, count_bytes=4, major=4):
    return struct.pack("<IIIIQQQII", LIST_CODE, major, minor, k, n_words, total, list_start, word_bytes, count_bytes)


def index_header(k=16, num_words=0, num_locations=0, kmers_start=72, files_start=72, locations_start=72, major=4):
    return struct.pack("<IIIIQQIIIIQQQ", INDEX_CODE, major, 0, k, num_words, num_locations, 8, 8, 16, 0, files_start, kmers_start, locations_start)


def _rec(n):
    return b"".join(struct.pack("<QI", 3 * i + 1, 1) for i in range(n))


def cases():
    small = "file size too small"
    out = [
        # v4.4 header that declares records of 0 + 0 bytes: the reference's product is 48, the readers would stride 12 x 2^40
        ("h_zero_record_bytes.list", list_header(minor=4, n_words=1 << 40, word_bytes=0, count_bytes=0) + _rec(2), small),
        # n_words = 2^64 / 12 + 1: 12 x n_words wraps to 8, the reference's test passes
        ("h_wrapping_product.list", list_header(minor=2, n_words=(1 << 64) // 12 + 1) + _rec(4), small),
        ("h_wrapping_product_v44.list", list_header(minor=4, n_words=(1 << 64) // 12 + 1) + _rec(4), small),
        # list_start beyond the end of the file, nothing behind it
        ("h_list_start_beyond_eof.list", list_header(minor=2, n_words=0, list_start=1 << 20) + _rec(1), small),
        ("h_list_start_huge.list", list_header(minor=4, n_words=1, list_start=(1 << 64) - 12) + _rec(1), small),
        # all ones
        ("h_n_words_all_ones.list", list_header(minor=2, n_words=(1 << 64) - 1) + _rec(3), small),
        ("h_n_words_all_ones_v40.list", list_header(minor=0, n_words=(1 << 64) - 1)[:40] + _rec(3), small),
        # v4.4 record bytes smaller than the stride: 2 x (4 + 4) = 16 bytes "fit", 2 x 12 do not
        ("h_short_record_bytes.list", list_header(minor=4, n_words=2, word_bytes=4, count_bytes=4) + b"\0" * 16, small),
        # truncated headers: < 16 bytes cannot be mapped as a header at all; < 40 / < 48 leave the totals short
        ("h_truncated_12.list", list_header()[:12], "could not mmap"),
        ("h_truncated_30.list", list_header(minor=2, n_words=5)[:30], small),
        ("h_truncated_44.list", list_header(minor=4, n_words=5)[:44], small),
        # GT4I twins
        ("h_index_num_words_all_ones.index", index_header(num_words=(1 << 64) - 1) + b"\0" * 64, small),
        ("h_index_wrapping_product.index", index_header(num_words=(1 << 60) + 1) + b"\0" * 64, small),
        ("h_index_kmers_beyond_eof.index", index_header(num_words=1, kmers_start=1 << 30) + b"\0" * 64, small),
        ("h_index_truncated.index", index_header()[:40], "could not mmap"),
    ]
    return out


def good_list(n=8, k=16):
    return list_header(minor=2, k=k, n_words=n, total=n) + _rec(n)
