"""The one-list iterator of include/gt4_set_operations.h (gt4_hip_word_slist_get_first_word / _get_next_word: the
reference's GT4WordSListInstance contract, src/word-list-sorted.h:42-57, src/word-list-sorted.c:59-78) through
ctypes on libgt4hip.so: every record in order, the end-of-list behaviour the reference's callers rely on (the last
call returns 0 and leaves word / count alone; idx == num_words), an empty list, a list longer than one host block,
and a file-backed handle (records straight from the mapping)."""
import ctypes as C
import os

import numpy as np
import pytest

from genometester4_amd import capi
from genometester4_amd.listio import make_records, write_list

pytestmark = pytest.mark.gpu


class Iter(C.Structure):
    _fields_ = [("num_words", C.c_uint64), ("sum_counts", C.c_uint64), ("idx", C.c_uint64), ("word", C.c_uint64), ("count", C.c_uint32),
                ("word_length", C.c_uint), ("list", C.c_void_p), ("block", C.c_void_p), ("block_first", C.c_uint64), ("block_count", C.c_uint64)]


def _lib():
    L = capi.lib()
    L.gt4_hip_word_list_new.restype = C.c_void_p
    L.gt4_hip_word_list_new.argtypes = [C.c_char_p, C.c_uint]
    L.gt4_hip_word_list_delete.argtypes = [C.c_void_p]
    L.gt4_hip_word_list_is_file_backed.argtypes = [C.c_void_p]
    L.gt4_hip_word_slist_get_first_word.argtypes = [C.c_void_p, C.POINTER(Iter)]
    L.gt4_hip_word_slist_get_next_word.argtypes = [C.POINTER(Iter)]
    L.gt4_hip_word_slist_iter_release.argtypes = [C.POINTER(Iter)]
    return L


def _walk(L, handle):
    it = Iter()
    got = []
    ok = L.gt4_hip_word_slist_get_first_word(handle, C.byref(it))
    while ok:
        got.append((it.word, it.count, it.idx))
        ok = L.gt4_hip_word_slist_get_next_word(C.byref(it))
    state = (it.idx, it.word, it.count, it.num_words, it.sum_counts, it.word_length)
    assert L.gt4_hip_word_slist_get_next_word(C.byref(it)) == 0  # exhausted stays exhausted
    L.gt4_hip_word_slist_iter_release(C.byref(it))
    return got, state


@pytest.mark.parametrize("n,limit", [(0, None), (1, None), (1000, None), (2_500_000, None), (300_000, "64K")])
def test_iterator_walks_every_record(tmp_path, n, limit, monkeypatch):
    rng = np.random.default_rng(n + 1)
    keys = np.unique(rng.integers(0, 1 << 40, size=n + n // 50, dtype=np.uint64))[:n]
    rec = make_records(keys, rng.integers(0, 9, size=len(keys), dtype=np.uint32))
    path = os.path.join(tmp_path, "l.list")
    write_list(path, rec, 20)
    if limit:
        monkeypatch.setenv("GT4HIP_HBM_LIMIT", limit)
    L = _lib()
    h = L.gt4_hip_word_list_new(path.encode(), 4)
    assert h
    assert bool(L.gt4_hip_word_list_is_file_backed(h)) == bool(limit)
    got, state = _walk(L, h)
    L.gt4_hip_word_list_delete(h)
    n = len(rec)
    assert len(got) == n
    if n:
        assert [g[0] for g in got] == rec["key"].tolist() and [g[1] for g in got] == rec["count"].tolist()
        assert [g[2] for g in got] == list(range(n))
        # src/word-list-sorted.c:74-75: idx runs one past the last record, word / count keep the last record
        assert state[:3] == (n, int(rec["key"][-1]), int(rec["count"][-1]))
    assert state[3:] == (n, int(rec["count"].astype(np.uint64).sum()), 20)
