"""Key-range sharding of the set operations across the GPUs of one node.

Every set operation is key-local, so the key space is cut into `world` contiguous prefix ranges;
rank g merges only the slices of the input lists that fall into range g, and concatenating the
per-rank outputs in rank order is the globally sorted result.  The only exchanges are

  1. an all-gather of each rank's (n_words, total_count)  -> header totals and file offsets
  2. a gatherv of the record payloads to the writer rank  -> RCCL has no native gatherv, so it is
     a grouped send/recv (`batch_isend_irecv` = ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd);
     on the fully connected xGMI node every non-root rank sends over its own link to the root.

One process per GPU.  Two forms of the same scheme live here:

  * `DeviceShards` -- the production form: lists resident in HBM, shards cut on the device, the
    merge and the RCCL gatherv both through the C ABI (`gt4hip_union_multi`, `gt4hip_comm_gatherv`),
    no host copy of the payload anywhere;
  * `sharded_nway` / `gatherv_records` / `exchange_totals` -- the same steps over `torch.distributed`
    tensors (backend "nccl" = RCCL on GPUs, "gloo" on CPU), which is how the CPU test-suite covers
    the scheme with world sizes 2 and 3 (the CPU oracle standing in for the per-shard merge).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .listio import RECORD_DTYPE


def key_range_bounds(word_length: int, world: int) -> List[int]:
    """world+1 ascending key bounds; shard g owns keys in [bounds[g], bounds[g+1]).

    Equal-width prefix ranges of the 4^k key space (2^64 for k = 32); the last bound is 2^64 so
    that the all-ones key belongs to the last shard."""
    space = 1 << 64 if word_length >= 32 else 1 << (2 * word_length)
    b = [(space * g) // world for g in range(world)]
    b.append(1 << 64)
    return b


def slice_indices(keys: np.ndarray, bounds: Sequence[int]) -> List[Tuple[int, int]]:
    """[first, last) record index of every shard in an ascending key array (two binary searches
    per shard boundary; slices are contiguous in the file)."""
    keys = np.asarray(keys, dtype=np.uint64)
    cuts = [0]
    for b in bounds[1:-1]:
        cuts.append(int(np.searchsorted(keys, np.uint64(b), side="left")))
    cuts.append(len(keys))
    return [(cuts[g], cuts[g + 1]) for g in range(len(bounds) - 1)]


def exchange_totals(n_words: int, total_count: int, device=None, group=None) -> List[Tuple[int, int]]:
    """All-gather of the per-shard header totals.  Returns [(n_words, total_count)] by rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    # total_count is a u64 header field; carry it as two non-negative int64 halves
    mine = torch.tensor([n_words, total_count & 0xFFFFFFFF, total_count >> 32], dtype=torch.int64, device=device)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    rows = torch.stack(out).cpu().tolist()  # ONE device-to-host copy (element-wise int() would synchronise 3 x world times)
    return [(int(r[0]), int(r[1]) | (int(r[2]) << 32)) for r in rows]


def gatherv_records(local, counts: Sequence[int], root: int = 0, group=None, out=None):
    """Gathers per-rank record payloads of different lengths on `root`.

    `local`: 1-D int32 tensor holding this rank's packed records (3 words per record) on the
    device the backend communicates from.  `counts[r]`: records of rank r (from exchange_totals).
    Returns the concatenated int32 tensor on root (rank order = key order; `out` if the root passes
    one of at least 3 x sum(counts) words -- its own extent may already lie there), None elsewhere."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    assert local.dtype == torch.int32 and local.numel() == 3 * counts[rank]
    if rank != root:
        if counts[rank]:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, root, group=group)]):
                w.wait()
        return None
    total = sum(counts)
    if out is None:
        out = torch.empty(3 * total, dtype=torch.int32, device=local.device)
    else:
        assert out.dtype == torch.int32 and out.numel() >= 3 * total
        out = out[: 3 * total]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ops = []
    for r in range(world):
        if not counts[r]:
            continue
        dst = out[3 * offs[r]: 3 * offs[r + 1]]
        if r == root:
            if dst.data_ptr() != local.data_ptr():
                dst.copy_(local)
        else:
            ops.append(dist.P2POp(dist.irecv, dst, r, group=group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return out


def records_to_tensor(records: np.ndarray, device=None):
    """Packed records -> int32 tensor view (3 words per record)."""
    import torch
    a = np.ascontiguousarray(records, dtype=RECORD_DTYPE).view(np.int32)
    t = torch.from_numpy(a.copy())
    return t.to(device) if device is not None else t


def tensor_to_records(t) -> np.ndarray:
    return t.detach().cpu().numpy().view(RECORD_DTYPE)


def sharded_nway(lists_host: Sequence[np.ndarray], word_length: int,
                 local_op: Callable[[List[np.ndarray]], Tuple[int, int, np.ndarray]],
                 root: int = 0, device=None, group=None) -> Optional[Tuple[int, int, np.ndarray]]:
    """Generic sharded N-way operation over host-resident sorted lists.

    Every rank holds (or can map) all input lists, takes its key-range slice of each, runs
    `local_op(slices) -> (n_words, total_count, records)` on them, and the results are gathered
    on `root`.  Returns (n_words, total_count, records) on root, None elsewhere.  `local_op` is
    the GPU merge in production and the CPU oracle in the CPU tests."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    bounds = key_range_bounds(word_length, world)
    mine = []
    for rec in lists_host:
        lo, hi = slice_indices(rec["key"], bounds)[rank]
        mine.append(rec[lo:hi])
    n, total, out = local_op(mine)
    totals = exchange_totals(n, total, device=device, group=group)
    gathered = gatherv_records(records_to_tensor(out, device), [t[0] for t in totals], root=root, group=group)
    if rank != root:
        return None
    return sum(t[0] for t in totals), sum(t[1] for t in totals) & 0xFFFFFFFFFFFFFFFF, tensor_to_records(gathered)


class _DeviceWords:
    """int32 view of device memory for torch (the CUDA array interface: no copy)"""

    def __init__(self, ptr, n_words32):
        self.__cuda_array_interface__ = {"shape": (int(n_words32),), "typestr": "<i4", "data": (int(ptr), False), "version": 3}


def list_as_tensor(lst, n_records, host=False):
    """The first n_records packed records of a device list as a 1-D int32 tensor: a zero-copy view of the list's HBM
    (backend "nccl" = RCCL), or -- host=True, the CPU backend of the one-device test hook -- a host copy."""
    import torch
    if host:
        rec = lst.download_range(0, n_records) if n_records else np.empty(0, dtype=RECORD_DTYPE)
        return torch.from_numpy(np.ascontiguousarray(rec).view(np.int32).copy())
    if not n_records:
        return torch.empty(0, dtype=torch.int32, device="cuda")
    return torch.as_tensor(_DeviceWords(lst.device_ptr, 3 * n_records), device="cuda")


class DeviceShards:
    """Key-range sharded N-way operations on lists that are RESIDENT in this rank's HBM.

    No host bounce anywhere: shard g of a list is the view between two device lower bounds
    (`gt4hip_list_lower_bound` + `gt4hip_list_slice`), the per-shard merge writes into a device list,
    and the payload is gathered from that buffer over RCCL by the C ABI (`gt4hip_comm_gatherv`:
    grouped ncclSend / ncclRecv) -- the same entry point the C command-line tool uses.  The
    communicator id is made by rank 0 (`capi.comm_unique_id`) and handed over by the caller
    (`torch.distributed.broadcast_object_list`, a file, MPI ...); `comm_id=None` means no communicator
    of the library's own: the exchanges then go over `torch.distributed`.

    WHICH exchange path the ranks take is decided ONCE, collectively, here: `agree(ok) -> bool`
    is a logical AND over the ranks (one all_reduce); the communicator is kept only if every rank made it and
    one trial all-gather went through everywhere.  After that nothing is caught: a failure inside a step
    propagates (a rank that silently changed paths would leave the others inside a different collective).
    `gather_via` -- "rccl" (gt4hip_comm_gatherv) or "torch" (gatherv_records over torch.distributed) -- is only
    ever changed by the caller, on all ranks at once (`use_torch_gather`)."""

    def __init__(self, ctx, rank=0, world=1, comm_id=None, agree=None):
        self.ctx, self.rank, self.world = ctx, rank, world
        self.comm = None
        self.comm_error = None
        self.last_ms = {}
        if comm_id is not None:
            ok = True
            try:
                self.comm = ctx.comm_create(comm_id, world, rank)
            except Exception as e:
                ok, self.comm_error = False, "communicator: %s" % e
            if agree is not None:
                ok = agree(ok)
            if ok and world > 1:
                try:
                    ctx.comm_allgather_totals(self.comm, world, 1, 1)
                except Exception as e:
                    ok, self.comm_error = False, "trial all-gather: %s" % e
                if agree is not None:
                    ok = agree(ok)
            if not ok:
                if self.comm_error is None:
                    self.comm_error = "another rank could not use the communicator"
                self.close()
        self.gather_via = "rccl" if self.comm is not None else "torch"
        self.gather_note = None
        self.host_tensors = False  # torch path: tensors on the host (backend gloo: the one-device test hook)

    def close(self):
        if self.comm is not None:
            from . import capi
            capi.comm_destroy(self.comm)
            self.comm = None

    def use_torch_gather(self, why):
        """every rank, together: the payload goes through torch.distributed from here on"""
        self.gather_via = "torch"
        self.gather_note = why

    def plan(self, lists, sampled=True):
        """The shards' first keys for a job over `lists` (all ranks hold the same lists and get the same cuts):
        SAMPLED from the lists (gt4hip_shard_cuts: equal numbers of input records per shard whatever the keys'
        distribution) or, `sampled=False`, equal-width ranges of the key space (gt4hip_shard_first_key: balanced
        for uniformly spread keys only)."""
        from . import capi
        if self.world == 1:
            self.cuts = [0]
        elif sampled:
            self.cuts = self.ctx.shard_cuts(lists, self.world)
        else:
            wl = max(l.word_length for l in lists)
            self.cuts = [int(capi.lib().gt4hip_shard_first_key(wl, self.world, g)) for g in range(self.world)]
        return self.cuts

    def shard_of(self, lst, word_length, rank=None):
        """Rank `rank`'s (default: this rank's) key range of a device list, as a view.  The cuts come from plan();
        without one: equal-width ranges."""
        from . import capi
        g = self.rank if rank is None else rank
        cuts = getattr(self, "cuts", None)
        if cuts is None:
            cuts = [int(capi.lib().gt4hip_shard_first_key(word_length, self.world, r)) for r in range(self.world)]
        first = lst.lower_bound(cuts[g]) if g else 0
        last = lst.lower_bound(cuts[g + 1]) if g + 1 < self.world else lst.n_words
        return lst.slice(first, last - first)

    def gather(self, local, counts, root, gathered):
        """The payload to `root`: over RCCL through the C ABI, or over torch.distributed (gatherv_records on views of
        the lists' HBM).  Returns the gathered device list on root (None elsewhere; None on root too when the tensors
        had to go through the host: the one-device test hook)."""
        import os
        hook = os.environ.get("GT4_BENCH_BREAK_GATHER", "")
        if self.gather_via == "rccl":
            if hook in ("1", "2") and self.rank == self.world - 1:  # test hook: the C gather fails on ONE rank, before it enters the collective
                raise RuntimeError("gt4hip_comm_gatherv: injected failure (GT4_BENCH_BREAK_GATHER)")
            if self.comm is None:
                raise RuntimeError("no RCCL communicator")
            self.ctx.comm_gatherv(self.comm, local, counts, root, gathered if self.rank == root else None)
            return gathered if self.rank == root else None
        if hook == "2":
            raise RuntimeError("gatherv_records: injected failure (GT4_BENCH_BREAK_GATHER=2)")
        if hook == "3" and self.rank == self.world - 1:
            import time
            time.sleep(10 ** 6)  # test hook: a rank that never arrives
        mine = list_as_tensor(local, counts[self.rank], host=self.host_tensors)
        out_t = None
        if self.rank == root and not self.host_tensors:
            out_t = list_as_tensor(gathered, sum(counts))
        res = gatherv_records(mine, counts, root=root, out=out_t)
        if self.rank != root:
            return None
        if self.host_tensors:
            self.host_gathered_words = int(res.numel())
            return None
        from . import capi
        capi.lib().gt4hip_list_set_n_words(gathered.h, sum(counts))
        return gathered

    def run(self, shards, op, totals_exchange, root=0, out=None, gathered=None, gather=True):
        """`op(shards, out) -> (n_words, total_count, device list)` on this rank's shards, then the
        totals all-gather (`totals_exchange(n, total) -> [(n, total)] by rank`) and the gatherv.
        Returns (n_words, total_count, gathered device list or None, per-rank totals)."""
        import time
        t0 = time.perf_counter()
        n, total, local = op(shards, out)  # (returns with the totals read back: the library's stream is idle)
        t1 = time.perf_counter()
        if self.world == 1:
            totals = [(n, total)]
        elif self.comm is not None:
            # over RCCL through the C ABI: one all-gather on the library's stream, one synchronisation (round 5; the
            # torch.distributed form costs a Python collective and a device-to-host copy per step)
            totals = self.ctx.comm_allgather_totals(self.comm, self.world, n, total)
        else:
            totals = totals_exchange(n, total)
        res = local
        if gather and self.world > 1:
            counts = [t[0] for t in totals]
            if self.rank == root and gathered is None and not self.host_tensors:
                gathered = self.ctx.alloc(max(1, sum(counts)), local.word_length)
            res = self.gather(local, counts, root, gathered)
        t2 = time.perf_counter()
        self.last_ms = {"merge": (t1 - t0) * 1e3, "exchange_and_gather": (t2 - t1) * 1e3}
        return sum(t[0] for t in totals), sum(t[1] for t in totals) & 0xFFFFFFFFFFFFFFFF, res, totals


def gpu_union_multi_op(ctx, cutoff: int = 1, rule: int = 0, count_override: int = 1):
    """op for DeviceShards.run: N-way union of device-resident shards (C ABI), result stays in HBM."""
    def op(shards, out=None):
        rc, n, total, res = ctx.union_multi(shards, cutoff, rule, count_override, out=out)
        if rc:
            raise RuntimeError("union_multi rejected rule %d" % rule)
        return n, total, res
    return op


def gpu_intersect_multi_op(ctx, cutoff: int = 1, rule: int = 0, count_override: int = 1):
    def op(shards, out=None):
        rc, n, total, res = ctx.intersect_multi(shards, cutoff, rule, count_override, out=out)
        if rc:
            raise RuntimeError("intersect_multi rejected rule %d" % rule)
        return n, total, res
    return op
