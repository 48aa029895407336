"""Worker of tests/test_distributed.py: one rank of a gloo (CPU) run of the sharded N-way path.
The per-shard merge is the CPU oracle here (test infrastructure); the sharding, the totals
all-gather and the gatherv are the product code in genometester4_amd/distributed.py."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import torch.distributed as dist  # noqa: E402

import oracle_lib as O  # noqa: E402
from genometester4_amd import distributed as D  # noqa: E402
from genometester4_amd.listio import make_records  # noqa: E402


def make_lists(k, n_lists, seed):
    rng = np.random.default_rng(seed)
    limit = (1 << 63) if k >= 32 else (1 << (2 * k))
    keys = np.unique(rng.integers(0, limit, size=20000, dtype=np.uint64))
    if k >= 32:
        keys = np.unique(np.concatenate([keys * np.uint64(2) + np.uint64(1), np.array([0, 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)]))
    lists = []
    for j in range(n_lists):
        m = rng.random(len(keys)) < (0.15 + 0.1 * j)
        lists.append(make_records(keys[m], rng.integers(1, 7, size=int(m.sum()), dtype=np.uint32)))
    lists.append(lists[0][:0])  # an empty member
    return lists


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ok = True
    for k, seed in ((12, 1), (25, 2), (32, 3)):
        lists = make_lists(k, 5, seed)
        for name, oracle_fn, cutoff, rule in (("union", O.union_multi, 1, 0), ("union", O.union_multi, 3, 4),
                                              ("intersect", O.intersect_multi, 1, 0), ("intersect", O.intersect_multi, 2, 1)):
            use = lists if name == "union" else lists[:-1]

            def local_op(slices, fn=oracle_fn, c=cutoff, r=rule):
                rc, n, total, recs = fn(slices, c, r, 1)
                assert rc == 0
                return n, total, recs

            got = D.sharded_nway(use, k, local_op, root=0)
            if rank == 0:
                rc, n, total, recs = oracle_fn(use, cutoff, rule, 1)
                same = got[0] == n and got[1] == total and got[2].tobytes() == recs.tobytes()
                if not same:
                    print("MISMATCH", k, name, cutoff, rule, got[0], n, flush=True)
                ok &= same
            else:
                assert got is None
    # ragged gatherv incl. an empty contribution
    import torch
    counts = [3 * (r + 1) if r != 1 else 0 for r in range(world)]
    local = torch.arange(3 * counts[rank], dtype=torch.int32) + 1000 * rank
    out = D.gatherv_records(local, counts, root=0)
    if rank == 0:
        exp = torch.cat([torch.arange(3 * c, dtype=torch.int32) + 1000 * r for r, c in enumerate(counts)])
        ok &= bool(torch.equal(out, exp))
    # the same into a buffer of the root's in which its own extent already lies (rank 0 of a sharded job merges straight
    # into the gathered list: no copy of its own records)
    total = sum(counts)
    if rank == 0:
        buf = torch.full((3 * total + 5,), -1, dtype=torch.int32)
        buf[: 3 * counts[0]] = local
        out2 = D.gatherv_records(buf[: 3 * counts[0]], counts, root=0, out=buf)
        ok &= bool(out2.data_ptr() == buf.data_ptr() and torch.equal(out2, exp) and int(buf[3 * total]) == -1)
    else:
        D.gatherv_records(local, counts, root=0)
    # which exchange path DeviceShards takes is agreed on ONCE: a communicator that ONE rank cannot make
    # (or whose trial all-gather fails on one rank) is dropped on EVERY rank; nothing is caught after that

    def agree(flag):
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    class StubCtx:
        def __init__(self, fail_create, fail_trial):
            self.fail_create, self.fail_trial, self.made = fail_create, fail_trial, 0

        def comm_create(self, comm_id, n, r):
            if self.fail_create:
                raise RuntimeError("injected: no communicator on this rank")
            self.made += 1
            return object()

        def comm_allgather_totals(self, comm, w, n, t):
            if self.fail_trial:
                raise RuntimeError("injected: trial all-gather")
            return [(n, t)] * w

    import genometester4_amd.capi as capi_mod
    destroyed = []
    real_destroy = capi_mod.comm_destroy
    capi_mod.comm_destroy = lambda c: destroyed.append(c)
    try:
        for fail_create, fail_trial, want in ((False, False, "rccl"), (rank == world - 1, False, "torch"), (False, rank == 0, "torch")):
            sh = D.DeviceShards(StubCtx(fail_create, fail_trial), rank, world, b"x" * 128, agree=agree)
            ok &= sh.gather_via == want and (sh.comm is not None) == (want == "rccl")
            if want == "torch":
                ok &= sh.comm_error is not None
    finally:
        capi_mod.comm_destroy = real_destroy
    bounds = D.key_range_bounds(25, world)
    ok &= bounds[0] == 0 and bounds[-1] == 1 << 64 and all(a < b for a, b in zip(bounds, bounds[1:]))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        with open(os.environ["GT4_DIST_RESULT"], "w") as f:
            f.write("OK" if ok else "FAIL")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
