"""ctypes binding to oracle/libgt4oracle.so -- the CPU checker.  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from genometester4_amd.listio import RECORD_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
REF_GLISTCOMPARE = os.path.join(REF_DIR, "glistcompare")
REF_SETOPS = os.path.join(REF_DIR, "ref_setops")

OP_UNION, OP_INTRSEC, OP_DIFF1, OP_DIFF2 = 1, 2, 4, 8
RULES = dict(default=0, add=1, subtract=2, min=3, max=4, first=5, second=6, number=7)


class Stat(C.Structure):
    _fields_ = [("n_words", C.c_uint64), ("total_count", C.c_uint64)]


CALLBACK = C.CFUNCTYPE(C.c_uint, C.c_uint64, C.POINTER(C.c_uint32), C.c_void_p)

_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "libgt4oracle.so")
        src = os.path.join(ORACLE_DIR, "gt4_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(so)
        u8p = C.c_void_p
        _lib.gt4o_compare.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint, C.c_int, C.c_uint32,
                                      C.c_int, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(Stat)]
        _lib.gt4o_compare.restype = C.c_int
        for name in ("gt4o_union_multi", "gt4o_intersect_multi"):
            f = getattr(_lib, name)
            f.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint, C.c_uint32, C.c_int,
                          C.c_uint32, C.c_void_p, C.POINTER(Stat)]
            f.restype = C.c_int
        _lib.gt4o_write_union.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint, C.c_uint32,
                                          C.c_void_p, C.POINTER(Stat)]
        _lib.gt4o_write_union.restype = C.c_int
        for name in ("gt4o_union", "gt4o_is_union"):
            f = getattr(_lib, name)
            f.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint, CALLBACK, C.c_void_p]
            f.restype = C.c_uint
    return _lib


def index_decode(file_bytes):
    """GT4I index file image -> (word_length, num_locations, records) via the C restatement."""
    buf = np.frombuffer(bytes(file_bytes), dtype=np.uint8)
    f = lib().gt4o_index_decode
    f.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p]
    f.restype = C.c_int
    wl, nw, nl = C.c_uint32(), C.c_uint64(), C.c_uint64()
    rc = f(buf.ctypes.data, len(buf), C.byref(wl), C.byref(nw), C.byref(nl), None)
    if rc:
        raise ValueError("gt4o_index_decode: %d" % rc)
    rec = np.zeros(nw.value, dtype=RECORD_DTYPE)
    f(buf.ctypes.data, len(buf), C.byref(wl), C.byref(nw), C.byref(nl), rec.ctypes.data)
    return wl.value, nl.value, rec


def _ptr(a: np.ndarray):
    return C.c_void_p(a.ctypes.data)


def _rec(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=RECORD_DTYPE)


def compare(a, b, ops, rule=0, cutoff=1, subtract=0, count_override=1, count_only=False):
    """Returns {op_bit: (n_words, total_count, records or None)} for the requested ops."""
    a, b = _rec(a), _rec(b)
    outs = (C.c_void_p * 4)()
    bufs = [None] * 4
    if not count_only:
        for k in range(4):
            if ops >> k & 1:
                bufs[k] = np.zeros(len(a) + len(b), dtype=RECORD_DTYPE)
                outs[k] = bufs[k].ctypes.data
    st = (Stat * 4)()
    rc = lib().gt4o_compare(_ptr(a), len(a), _ptr(b), len(b), ops, rule, cutoff, subtract, count_override,
                            None if count_only else outs, st)
    assert rc == 0
    res = {}
    for k in range(4):
        if ops >> k & 1:
            recs = None if count_only else bufs[k][: st[k].n_words].copy()
            res[1 << k] = (st[k].n_words, st[k].total_count, recs)
    return res


def _multi_args(lists):
    lists = [_rec(x) for x in lists]
    ptrs = (C.c_void_p * len(lists))(*[x.ctypes.data for x in lists])
    ns = (C.c_uint64 * len(lists))(*[len(x) for x in lists])
    return lists, ptrs, ns


def _multi(fn, lists, cutoff, rule, count_override):
    lists, ptrs, ns = _multi_args(lists)
    out = np.zeros(sum(len(x) for x in lists), dtype=RECORD_DTYPE)
    st = Stat()
    rc = fn(ptrs, ns, len(lists), cutoff, rule, count_override, _ptr(out), C.byref(st))
    if rc:
        return rc, None, None, None
    return 0, st.n_words, st.total_count, out[: st.n_words].copy()


def union_multi(lists, cutoff=1, rule=0, count_override=1):
    return _multi(lib().gt4o_union_multi, lists, cutoff, rule, count_override)


def intersect_multi(lists, cutoff=1, rule=0, count_override=1):
    return _multi(lib().gt4o_intersect_multi, lists, cutoff, rule, count_override)


def write_union(lists, cutoff=1):
    lists, ptrs, ns = _multi_args(lists)
    out = np.zeros(sum(len(x) for x in lists), dtype=RECORD_DTYPE)
    st = Stat()
    rc = lib().gt4o_write_union(ptrs, ns, len(lists), cutoff, _ptr(out), C.byref(st))
    return rc, st.n_words, st.total_count, out[: st.n_words].copy()


def _walk(fn, lists, stop_after=0):
    lists, ptrs, ns = _multi_args(lists)
    rows = []
    n = len(lists)

    def cb(word, counts, _):
        rows.append((word,) + tuple(counts[j] for j in range(n)))
        return 7 if stop_after and len(rows) == stop_after else 0

    r = fn(ptrs, ns, n, CALLBACK(cb), None)
    return r, rows


def union_walk(lists, stop_after=0):
    return _walk(lib().gt4o_union, lists, stop_after)


def is_union_walk(lists, stop_after=0):
    return _walk(lib().gt4o_is_union, lists, stop_after)
