"""ctypes binding of the C ABI in include/gt4hip.h (libgt4hip.so) -- plumbing for tests and bench.

Nothing here computes: every call goes straight into the HIP library.  If the library is missing
this module raises at import of `lib()`; there is no Python or CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .listio import RECORD_DTYPE

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GT4HIP_LIB") or os.path.join(PKG_DIR, "libgt4hip.so")  # GT4HIP_LIB: diagnostic builds

OK = 0
EINVAL, ENODEVICE, ENOMEM, ERULE, EHIP, EWORDLEN, EINTERNAL = 1, 2, 3, 4, 5, 6, 7
OP_UNION, OP_INTRSEC, OP_DIFF1, OP_DIFF2 = 1, 2, 4, 8
RULE_DEFAULT, RULE_ADD, RULE_SUBTRACT, RULE_MIN, RULE_MAX, RULE_FIRST, RULE_SECOND, RULE_NUMBER = range(8)


class CompareParams(C.Structure):
    _fields_ = [("ops", C.c_uint32), ("rule", C.c_int32), ("cutoff", C.c_uint32), ("subtract", C.c_int32),
                ("count_override", C.c_uint32), ("count_only", C.c_int32)]


class CompareResult(C.Structure):
    _fields_ = [("n_words", C.c_uint64 * 4), ("total_count", C.c_uint64 * 4), ("out", C.c_void_p * 4),
                ("merge_kernel_ms", C.c_double), ("device_ms", C.c_double), ("merge_tiles", C.c_uint64)]


class MultiResult(C.Structure):
    _fields_ = [("n_words", C.c_uint64), ("total_count", C.c_uint64), ("out", C.c_void_p), ("device_ms", C.c_double),
                ("records_read", C.c_uint64), ("records_written", C.c_uint64)]


class CountTable(C.Structure):
    _fields_ = [("n_keys", C.c_uint64), ("n_lists", C.c_uint32), ("device_keys", C.c_void_p),
                ("device_counts", C.c_void_p), ("owner", C.c_void_p * 2), ("ragged", C.c_void_p)]


# every symbol include/gt4hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "gt4hip_create", "gt4hip_destroy", "gt4hip_last_error", "gt4hip_strerror", "gt4hip_device_count",
    "gt4hip_device_info", "gt4hip_list_upload", "gt4hip_list_upload_index", "gt4hip_list_wrap", "gt4hip_list_alloc", "gt4hip_list_slice",
    "gt4hip_list_download", "gt4hip_list_download_range", "gt4hip_list_free", "gt4hip_list_n_words",
    "gt4hip_list_word_length", "gt4hip_list_device_ptr", "gt4hip_list_set_n_words", "gt4hip_list_sum_counts",
    "gt4hip_list_is_sorted", "gt4hip_list_lower_bound", "gt4hip_list_get_word", "gt4hip_compare",
    "gt4hip_union_multi", "gt4hip_intersect_multi", "gt4hip_union_table", "gt4hip_probe_table", "gt4hip_probe_table_ex", "gt4hip_table_compact", "gt4hip_table_download",
    "gt4hip_table_free", "gt4hip_generate", "gt4hip_generate_ex", "gt4hip_synchronize", "gt4hip_set_option",
    "gt4hip_get_counter", "gt4hip_device_memory", "gt4hip_list_upload_fd", "gt4hip_list_load_fd", "gt4hip_list_load",
    "gt4hip_list_write_fd", "gt4hip_lists_write_fd", "gt4hip_shard_first_key", "gt4hip_shard_cuts", "gt4hip_comm_unique_id", "gt4hip_comm_create", "gt4hip_comm_destroy", "gt4hip_comm_allgather_totals", "gt4hip_comm_allgather_u64", "gt4hip_context_device", "gt4hip_trim",
    "gt4hip_comm_rank", "gt4hip_comm_size", "gt4hip_comm_last_error", "gt4hip_comm_gatherv", "gt4hip_sort_words", "gt4hip_words_to_list",
    "gt4hip_device_words_to_list",
]

_lib = None


class Gt4HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gt4hip error %d: %s" % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `make -C genometester4_amd/csrc` "
                              "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32
        sig = {
            "gt4hip_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
            "gt4hip_destroy": (None, [vp]),
            "gt4hip_last_error": (C.c_char_p, [vp]),
            "gt4hip_strerror": (C.c_char_p, [C.c_int]),
            "gt4hip_device_count": (C.c_int, []),
            "gt4hip_device_info": (C.c_char_p, [vp]),
            "gt4hip_list_upload": (C.c_int, [vp, vp, u64, u32, C.POINTER(vp)]),
            "gt4hip_list_upload_index": (C.c_int, [vp, vp, u64, u64, u32, C.POINTER(vp)]),
            "gt4hip_list_wrap": (C.c_int, [vp, vp, u64, u32, C.POINTER(vp)]),
            "gt4hip_list_alloc": (C.c_int, [vp, u64, u32, C.POINTER(vp)]),
            "gt4hip_list_slice": (C.c_int, [vp, vp, u64, u64, C.POINTER(vp)]),
            "gt4hip_list_download": (C.c_int, [vp, vp, vp]),
            "gt4hip_list_download_range": (C.c_int, [vp, vp, u64, u64, vp]),
            "gt4hip_list_free": (None, [vp]),
            "gt4hip_list_n_words": (u64, [vp]),
            "gt4hip_list_word_length": (u32, [vp]),
            "gt4hip_list_device_ptr": (vp, [vp]),
            "gt4hip_list_set_n_words": (C.c_int, [vp, u64]),
            "gt4hip_list_sum_counts": (C.c_int, [vp, vp, C.POINTER(u64)]),
            "gt4hip_list_is_sorted": (C.c_int, [vp, vp, C.POINTER(C.c_int)]),
            "gt4hip_list_lower_bound": (C.c_int, [vp, vp, u64, C.POINTER(u64)]),
            "gt4hip_list_get_word": (C.c_int, [vp, vp, u64, C.POINTER(u64), C.POINTER(u32)]),
            "gt4hip_compare": (C.c_int, [vp, vp, vp, C.POINTER(CompareParams), C.POINTER(CompareResult)]),
            "gt4hip_union_multi": (C.c_int, [vp, C.POINTER(vp), u32, u32, i32, u32, i32, C.POINTER(MultiResult)]),
            "gt4hip_intersect_multi": (C.c_int, [vp, C.POINTER(vp), u32, u32, i32, u32, i32, C.POINTER(MultiResult)]),
            "gt4hip_union_table": (C.c_int, [vp, C.POINTER(vp), u32, C.POINTER(CountTable)]),
            "gt4hip_probe_table": (C.c_int, [vp, C.POINTER(vp), u32, C.POINTER(CountTable)]),
            "gt4hip_probe_table_ex": (C.c_int, [vp, C.POINTER(vp), u32, C.c_int, C.POINTER(CountTable)]),
            "gt4hip_table_compact": (C.c_int, [vp, C.POINTER(CountTable)]),
            "gt4hip_table_download": (C.c_int, [vp, C.POINTER(CountTable), u64, u64, vp, vp]),
            "gt4hip_table_free": (None, [C.POINTER(CountTable)]),
            "gt4hip_generate": (C.c_int, [vp, vp, u64, u64, u32]),
            "gt4hip_generate_ex": (C.c_int, [vp, vp, u64, u64, u64, u32, u64, u64]),
            "gt4hip_synchronize": (C.c_int, [vp]),
            "gt4hip_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
            "gt4hip_get_counter": (C.c_int, [vp, C.c_char_p, C.POINTER(u64)]),
            "gt4hip_device_memory": (C.c_int, [vp, C.POINTER(u64), C.POINTER(u64)]),
            "gt4hip_list_upload_fd": (C.c_int, [vp, C.c_int, u64, u64, u32, C.POINTER(vp)]),
            "gt4hip_list_load_fd": (C.c_int, [vp, vp, C.c_int, u64, u64]),
            "gt4hip_list_load": (C.c_int, [vp, vp, vp, u64]),
            "gt4hip_list_write_fd": (C.c_int, [vp, vp, u64, u64, C.c_int, u64]),
            "gt4hip_lists_write_fd": (C.c_int, [vp, u32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_int), C.POINTER(u64)]),
            "gt4hip_shard_first_key": (u64, [u32, u32, u32]),
            "gt4hip_shard_cuts": (C.c_int, [vp, C.POINTER(vp), u32, u32, C.POINTER(u64)]),
            "gt4hip_comm_unique_id": (C.c_int, [vp]),
            "gt4hip_comm_create": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]),
            "gt4hip_comm_destroy": (None, [vp]),
            "gt4hip_comm_rank": (C.c_int, [vp]),
            "gt4hip_comm_size": (C.c_int, [vp]),
            "gt4hip_comm_last_error": (C.c_char_p, []),
            "gt4hip_comm_gatherv": (C.c_int, [vp, vp, C.POINTER(u64), C.c_int, vp]),
            "gt4hip_comm_allgather_totals": (C.c_int, [vp, u64, u64, C.POINTER(u64)]),
            "gt4hip_comm_allgather_u64": (C.c_int, [vp, C.POINTER(u64), u32, C.POINTER(u64)]),
            "gt4hip_sort_words": (C.c_int, [vp, vp, u64, u32]),
            "gt4hip_words_to_list": (C.c_int, [vp, vp, u64, u32, C.POINTER(vp)]),
            "gt4hip_device_words_to_list": (C.c_int, [vp, vp, u64, u32, C.POINTER(vp)]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    rc = lib().gt4hip_comm_unique_id(buf)
    if rc:
        raise Gt4HipError(rc, lib().gt4hip_comm_last_error().decode())
    return buf.raw


def comm_destroy(comm):
    lib().gt4hip_comm_destroy(comm)


class DeviceList:
    """Owning handle of a gt4hip_list."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle
        ctx._lists.add(self)

    @property
    def n_words(self):
        return lib().gt4hip_list_n_words(self.h)

    @property
    def word_length(self):
        return lib().gt4hip_list_word_length(self.h)

    @property
    def device_ptr(self):
        return lib().gt4hip_list_device_ptr(self.h)

    def download(self) -> np.ndarray:
        out = np.empty(self.n_words, dtype=RECORD_DTYPE)
        self.ctx._chk(lib().gt4hip_list_download(self.ctx.h, self.h, out.ctypes.data))
        return out

    def download_range(self, first, count) -> np.ndarray:
        out = np.empty(count, dtype=RECORD_DTYPE)
        self.ctx._chk(lib().gt4hip_list_download_range(self.ctx.h, self.h, first, count, out.ctypes.data))
        return out

    def sum_counts(self) -> int:
        v = C.c_uint64()
        self.ctx._chk(lib().gt4hip_list_sum_counts(self.ctx.h, self.h, C.byref(v)))
        return v.value

    def is_sorted(self) -> bool:
        v = C.c_int()
        self.ctx._chk(lib().gt4hip_list_is_sorted(self.ctx.h, self.h, C.byref(v)))
        return bool(v.value)

    def lower_bound(self, key) -> int:
        v = C.c_uint64()
        self.ctx._chk(lib().gt4hip_list_lower_bound(self.ctx.h, self.h, key, C.byref(v)))
        return v.value

    def get_word(self, idx):
        w, c = C.c_uint64(), C.c_uint32()
        self.ctx._chk(lib().gt4hip_list_get_word(self.ctx.h, self.h, idx, C.byref(w), C.byref(c)))
        return w.value, c.value

    def slice(self, first, count) -> "DeviceList":
        h = C.c_void_p()
        self.ctx._chk(lib().gt4hip_list_slice(self.ctx.h, self.h, first, count, C.byref(h)))
        v = DeviceList(self.ctx, h)
        v._parent = self  # keep storage alive
        return v

    def free(self):
        # lists return their storage to the context's pool, so they must not outlive it
        if self.h and self.ctx.h:
            lib().gt4hip_list_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    def __init__(self, device=0):
        import weakref
        self._lists = weakref.WeakSet()
        self.h = C.c_void_p()
        rc = lib().gt4hip_create(device, C.byref(self.h))
        if rc:
            raise Gt4HipError(rc, lib().gt4hip_last_error(None).decode())

    def _chk(self, rc):
        if rc:
            raise Gt4HipError(rc, lib().gt4hip_last_error(self.h).decode())

    def close(self):
        if self.h:
            for l in list(self._lists):
                l.free()
            lib().gt4hip_destroy(self.h)
            self.h = None

    def device_info(self):
        return lib().gt4hip_device_info(self.h).decode()

    def set_option(self, name, value):
        self._chk(lib().gt4hip_set_option(self.h, name.encode(), value))

    def get_counter(self, name) -> int:
        v = C.c_uint64()
        self._chk(lib().gt4hip_get_counter(self.h, name.encode(), C.byref(v)))
        return v.value

    def synchronize(self):
        self._chk(lib().gt4hip_synchronize(self.h))

    def shard_cuts(self, lists, n_shards):
        """First key of every key-range shard, from samples of the lists themselves (gt4hip_shard_cuts)."""
        arr = (C.c_void_p * len(lists))(*[l.h for l in lists])
        out = (C.c_uint64 * n_shards)()
        self._chk(lib().gt4hip_shard_cuts(self.h, arr, len(lists), n_shards, out))
        return [int(x) for x in out]

    def words_to_list(self, words, word_length) -> "DeviceList":
        """Packed k-mer words (any order, repeats) -> sorted (word, occurrences) list on the device."""
        w = np.ascontiguousarray(words, dtype=np.uint64)
        h = C.c_void_p()
        self._chk(lib().gt4hip_words_to_list(self.h, w.ctypes.data if len(w) else None, len(w), word_length, C.byref(h)))
        return DeviceList(self, h)

    def sort_words(self, device_ptr, n_words, word_length):
        """Sorts n_words packed words at `device_ptr` (device memory) ascending, in place."""
        self._chk(lib().gt4hip_sort_words(self.h, C.c_void_p(device_ptr), n_words, word_length))

    def device_words_to_list(self, device_ptr, n_words, word_length) -> "DeviceList":
        """The same for n_words packed words at `device_ptr` (device memory; sorted in place)."""
        h = C.c_void_p()
        self._chk(lib().gt4hip_device_words_to_list(self.h, C.c_void_p(device_ptr), n_words, word_length, C.byref(h)))
        return DeviceList(self, h)

    def union_table_device(self, lists):
        """gt4hip_union_table without the download: (n_keys, free function) -- for timing."""
        arr = (C.c_void_p * len(lists))(*[l.h for l in lists])
        t = CountTable()
        self._chk(lib().gt4hip_union_table(self.h, arr, len(lists), C.byref(t)))
        n = t.n_keys
        lib().gt4hip_table_free(C.byref(t))
        return n

    def device_memory(self):
        f, t = C.c_uint64(), C.c_uint64()
        self._chk(lib().gt4hip_device_memory(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def upload_fd(self, fd, file_offset, n_words, word_length) -> "DeviceList":
        h = C.c_void_p()
        self._chk(lib().gt4hip_list_upload_fd(self.h, fd, file_offset, n_words, word_length, C.byref(h)))
        return DeviceList(self, h)

    def write_fd(self, lst, first, count, fd, file_offset):
        self._chk(lib().gt4hip_list_write_fd(self.h, lst.h, first, count, fd, file_offset))

    def comm_create(self, comm_id: bytes, n_ranks: int, rank: int):
        """RCCL communicator of this context's GPU (id from `comm_unique_id()` of ONE rank)."""
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(comm_id), 128)
        self._chk(lib().gt4hip_comm_create(self.h, buf, n_ranks, rank, C.byref(h)))
        return h

    def comm_allgather_totals(self, comm, world, n_words, total_count):
        """[(n_words, total_count)] by rank: one ncclAllGather on the library's stream (gt4hip_comm_allgather_totals)."""
        out = (C.c_uint64 * (2 * world))()
        self._chk(lib().gt4hip_comm_allgather_totals(comm, n_words, total_count, out))
        return [(int(out[2 * r]), int(out[2 * r + 1])) for r in range(world)]

    def comm_allgather_u64(self, comm, world, words):
        """every rank's `words` (at most eight u64) on every rank: [[words of rank 0], ...] (gt4hip_comm_allgather_u64)"""
        n = len(words)
        mine = (C.c_uint64 * n)(*[int(w) & 0xFFFFFFFFFFFFFFFF for w in words])
        out = (C.c_uint64 * (n * world))()
        self._chk(lib().gt4hip_comm_allgather_u64(comm, mine, n, out))
        return [[int(out[n * r + i]) for i in range(n)] for r in range(world)]

    def comm_gatherv(self, comm, local, counts, root=0, gathered=None):
        arr = (C.c_uint64 * len(counts))(*counts)
        self._chk(lib().gt4hip_comm_gatherv(comm, local.h if local is not None else None, arr, root,
                                             gathered.h if gathered is not None else None))

    def upload_index(self, kmers, num_locations, word_length) -> DeviceList:
        """kmers: (n, 2) uint64 array of (word, first location) entries of a GT4I index."""
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        h = C.c_void_p()
        self._chk(lib().gt4hip_list_upload_index(self.h, kmers.ctypes.data, len(kmers), num_locations, word_length, C.byref(h)))
        return DeviceList(self, h)

    def upload(self, records, word_length) -> DeviceList:
        rec = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
        h = C.c_void_p()
        self._chk(lib().gt4hip_list_upload(self.h, rec.ctypes.data if len(rec) else None, len(rec), word_length, C.byref(h)))
        return DeviceList(self, h)

    def alloc(self, capacity, word_length) -> DeviceList:
        h = C.c_void_p()
        self._chk(lib().gt4hip_list_alloc(self.h, capacity, word_length, C.byref(h)))
        return DeviceList(self, h)

    def wrap(self, device_ptr, n_words, word_length) -> DeviceList:
        h = C.c_void_p()
        self._chk(lib().gt4hip_list_wrap(self.h, device_ptr, n_words, word_length, C.byref(h)))
        return DeviceList(self, h)

    def generate(self, lst: DeviceList, n, seed, max_count=8):
        self._chk(lib().gt4hip_generate(self.h, lst.h, n, seed, max_count))

    def generate_ex(self, lst: DeviceList, n, key_seed, count_seed, max_count=8, mult=1, add=0):
        self._chk(lib().gt4hip_generate_ex(self.h, lst.h, n, key_seed, count_seed, max_count, mult, add))

    def compare(self, a: DeviceList, b: DeviceList, ops, rule=0, cutoff=1, subtract=0, count_override=1,
                count_only=False, out=None):
        """Returns (stats, lists, timing): stats[bit] = (n_words, total_count); lists[bit] = DeviceList."""
        prm = CompareParams(ops, rule, cutoff, subtract, count_override, 1 if count_only else 0)
        res = CompareResult()
        if out:
            for k in range(4):
                if out.get(1 << k) is not None:
                    res.out[k] = out[1 << k].h.value
        self._chk(lib().gt4hip_compare(self.h, a.h, b.h, C.byref(prm), C.byref(res)))
        stats, lists = {}, {}
        for k in range(4):
            if ops >> k & 1:
                stats[1 << k] = (res.n_words[k], res.total_count[k])
                if not count_only:
                    if out and out.get(1 << k) is not None:
                        lists[1 << k] = out[1 << k]
                    else:
                        lists[1 << k] = DeviceList(self, C.c_void_p(res.out[k]))
        timing = dict(merge_kernel_ms=res.merge_kernel_ms, device_ms=res.device_ms, merge_tiles=res.merge_tiles)
        return stats, lists, timing

    def _multi(self, fn, lists, cutoff, rule, count_override, count_only, out=None):
        arr = (C.c_void_p * len(lists))(*[l.h for l in lists])
        res = MultiResult()
        if out is not None:
            res.out = out.h.value
        rc = fn(self.h, arr, len(lists), cutoff, rule, count_override, 1 if count_only else 0, C.byref(res))
        if rc == ERULE:
            return rc, None, None, None
        self._chk(rc)
        if count_only:
            result = None
        elif out is not None:
            result = out
        else:
            result = DeviceList(self, C.c_void_p(res.out))
        self.last_multi_device_ms = res.device_ms
        self.last_multi_records = (res.records_read, res.records_written)
        return 0, res.n_words, res.total_count, result

    def union_multi(self, lists, cutoff=1, rule=0, count_override=1, count_only=False, out=None):
        return self._multi(lib().gt4hip_union_multi, lists, cutoff, rule, count_override, count_only, out)

    def intersect_multi(self, lists, cutoff=1, rule=0, count_override=1, count_only=False, out=None):
        return self._multi(lib().gt4hip_intersect_multi, lists, cutoff, rule, count_override, count_only, out)

    def union_table(self, lists, probe=False, presence=False, compact=False):
        """(keys, counts) of the count table; compact: through gt4hip_table_compact first (a ragged table made
        contiguous on the device); self.last_table_was_ragged says what the library built."""
        arr = (C.c_void_p * len(lists))(*[l.h for l in lists])
        t = CountTable()
        if probe:
            self._chk(lib().gt4hip_probe_table_ex(self.h, arr, len(lists), 1 if presence else 0, C.byref(t)))
        else:
            self._chk(lib().gt4hip_union_table(self.h, arr, len(lists), C.byref(t)))
        self.last_table_was_ragged = bool(t.ragged)
        if compact:
            self._chk(lib().gt4hip_table_compact(self.h, C.byref(t)))
            assert not t.ragged
        keys = np.empty(t.n_keys, dtype=np.uint64)
        counts = np.empty((t.n_keys, len(lists)), dtype=np.uint32)
        if t.n_keys:
            self._chk(lib().gt4hip_table_download(self.h, C.byref(t), 0, t.n_keys, keys.ctypes.data, counts.ctypes.data))
        lib().gt4hip_table_free(C.byref(t))
        return keys, counts
