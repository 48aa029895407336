"""The multi-rank paths on real devices.

With TWO OR MORE GPUs visible: `bench.py` under `torch.distributed.run` with the nccl (= RCCL) backend --
the strong-scaling intersection (one job cut by gt4hip_shard_first_key, header totals all-gathered
every step) and the 8-way union with the RCCL gatherv of the C ABI (grouped ncclSend / ncclRecv between
two devices) -- must report the very totals the single-GPU run reports; `glistcompare --gpus 2` with
GT4HIP_GATHER=rccl is covered by tests/test_sharded_cli.py (it expects the reference's bytes there).
With ONE GPU these cases skip, and the bench's multi-rank control flow is driven with both ranks on
device 0 over gloo instead (GT4_BENCH_ONE_DEVICE: RCCL refuses two ranks on one device).

Reference for what is sharded: scripts/MakeUnion.pl:31-95 (the tree of pairwise unions this replaces),
src/glistcompare.c:843-905 (the pair loop)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_devices():
    from genometester4_amd import capi
    return capi.lib().gt4hip_device_count()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench_raw(n_ranks, args, env_extra=None):
    """-> (exit code, the ONE JSON line of stdout, stderr)"""
    env = dict(os.environ, **(env_extra or {}))
    if n_ranks == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks)] + args
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    if n_ranks == 1:
        assert p.stdout.strip() == lines[0], "stdout carries the line and nothing else: " + p.stdout[-500:]
    return p.returncode, json.loads(lines[0]), p.stderr


def _bench(n_ranks, args, env_extra=None):
    rc, line, err = _bench_raw(n_ranks, args, env_extra)
    assert rc == 0, err[-3000:]
    return line


SMALL_PAIR = ["--workload", "intersect", "--entries", "6000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-union8"]
SMALL_BOTH = ["--entries", "6000000", "--entries8", "1500000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
SMALL_UNION = ["--workload", "union8", "--entries8", "1500000", "--steps", "2", "--warmup", "1"]


def test_weak_scaling_line_equals_the_default_line_on_one_gpu():
    one = _bench(1, SMALL_PAIR)
    assert one["scaling"] == "strong" and one["self_check"] == "ok"  # the default: the job of BASELINE's metric
    weak = _bench(1, SMALL_PAIR + ["--scaling", "weak"])
    assert weak["config"]["output_records"] == one["config"]["output_records"]
    assert weak["config"]["output_total_count"] == one["config"]["output_total_count"]
    assert weak["metric"] == one["metric"] and weak["n_gpus"] == 1


def test_default_line_carries_the_union8_record_and_checks_itself():
    """what the driver runs: no workload flags -> the intersection line with the 8-way union embedded, both
    checked against the generator's closed forms (|A n B| = n / 2, 5 n distinct keys)"""
    r = _bench(1, SMALL_BOTH)
    assert r["self_check"] == "ok" and r["config"]["output_records"]["intrsec"] == 3000000
    u = r["union8"]
    assert u["self_check"] == "ok" and u["output_records"] == 5 * 1500000
    assert u["value_with_gather"] > 0 and u["merge_only"] > 0 and u["gathered_bytes"] == 0
    alone = _bench(1, SMALL_UNION)
    assert (alone["config"]["output_records"], alone["config"]["output_total_count"]) == (u["output_records"], u["output_total_count"])


def test_two_ranks_on_one_device_default_line():
    """the driver's N > 1 command with both ranks on device 0 (gloo; no RCCL gather there): strong scaling by
    default, union8 embedded, the totals those of one GPU"""
    one = _bench(1, SMALL_BOTH)
    two = _bench(2, SMALL_BOTH, {"GT4_BENCH_ONE_DEVICE": "1"})
    assert two["scaling"] == "strong" and two["n_gpus"] == 2 and two["self_check"] == "ok"
    assert two["config"]["output_records"] == one["config"]["output_records"]["intrsec"]
    assert two["union8"]["self_check"] == "ok"
    assert (two["union8"]["output_records"], two["union8"]["output_total_count"]) == (one["union8"]["output_records"], one["union8"]["output_total_count"])
    assert sum(r["shard_input_records"] for r in two["union8"]["per_rank"]) == 8 * 1500000


def test_eight_ranks_on_one_device_default_line():
    """the driver's N = 8 command with all eight ranks on device 0 (gloo): the totals of one GPU, every record in
    exactly one shard, empty shards included (equal-width ranges of a clustered list leave ranks without records)"""
    small = ["--entries", "1600000", "--entries8", "400000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    one = _bench(1, small)
    eight = _bench(8, small, {"GT4_BENCH_ONE_DEVICE": "1"})
    assert eight["scaling"] == "strong" and eight["n_gpus"] == 8 and eight["self_check"] == "ok"
    assert eight["config"]["output_records"] == one["config"]["output_records"]["intrsec"]
    u = eight["union8"]
    assert u["self_check"] == "ok"
    assert (u["output_records"], u["output_total_count"]) == (one["union8"]["output_records"], one["union8"]["output_total_count"])
    assert len(u["per_rank"]) == 8 and sum(r["shard_input_records"] for r in u["per_rank"]) == 8 * 400000


@pytest.mark.parametrize("dist,ranks", [("clustered", 8), ("genomic", 4)])
def test_sampled_splitters_balance_the_shards(dist, ranks):
    """gt4hip_shard_cuts: the shards of lists whose keys are NOT spread evenly hold the same number of
    input records within 5 %; equal-width ranges of the key space give the same union (eight ranks on the clustered lists,
    four on the genomic ones: every rank is a process with a torch of its own, and the suite has a wall clock)"""
    args = ["--workload", "union8", "--entries8", "400000", "--dist", dist, "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    r = _bench(ranks, args, {"GT4_BENCH_ONE_DEVICE": "1"})
    assert r["self_check"] == "ok" and r["config"]["splitters"] == "sampled"
    loads = [x["shard_input_records"] for x in r["config"]["per_rank"]]
    mean = sum(loads) / float(ranks)
    assert len(loads) == ranks and max(loads) <= 1.05 * mean and min(loads) >= 0.95 * mean, loads
    if dist == "genomic":
        e = _bench(ranks, args + ["--splitters", "equal"], {"GT4_BENCH_ONE_DEVICE": "1"})
        assert e["self_check"] == "ok"
        assert (e["config"]["output_records"], e["config"]["output_total_count"]) == (r["config"]["output_records"], r["config"]["output_total_count"])


def test_projection_from_one_gpu_adds_up():
    """bench.py --project-shards 8: the eight shards' outputs add up to the whole job's, the line says what it projects"""
    r = _bench(1, ["--workload", "union8", "--entries8", "400000", "--steps", "2", "--warmup", "1", "--project-shards", "8"])
    assert r["self_check"] == "ok" and r["unit"] == "x" and r["value"] > 0
    proj = r["config"]["projection"]
    assert [p["splitters"] for p in proj] == ["sampled", "equal"]
    for p in proj:
        assert p["outputs_add_up"] and len(p["per_shard"]) == 8
        assert sum(x["input_records"] for x in p["per_shard"]) == 8 * 400000


def test_union32_lines_say_which_path_they_took():
    """glistmaker's collation width: lists that share no key go through ONE pass of the 32-list tile kernel, the default
    lists (even lists the same) through levels of eight-way merges -- the library probes; both check their totals"""
    base = ["--workload", "union32", "--entries32", "300000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    d = _bench(1, base + ["--dist", "disjoint"])
    assert d["self_check"] == "ok" and d["config"]["output_records"] == 32 * 300000
    assert d["config"]["path"].startswith("one pass of the N-way tile kernel (up to 32 lists per launch")
    s_ = _bench(1, base)
    assert s_["self_check"] == "ok" and s_["config"]["output_records"] == 17 * 300000
    assert s_["config"]["path"].startswith("levels of eight-way passes")


def test_rccl_totals_exchange_with_a_communicator_of_one():
    """gt4hip_comm_allgather_u64 / _totals (the step's totals exchange over RCCL, C ABI): the call itself, with the only
    communicator a one-GPU box can make"""
    from genometester4_amd import capi
    ctx = capi.Context(0)
    try:
        comm = ctx.comm_create(capi.comm_unique_id(), 1, 0)
        assert ctx.comm_allgather_u64(comm, 1, [1, 2, (1 << 64) - 1, 4]) == [[1, 2, (1 << 64) - 1, 4]]
        assert ctx.comm_allgather_totals(comm, 1, 7, 1 << 40) == [(7, 1 << 40)]
        for n in range(1, 9):
            assert ctx.comm_allgather_u64(comm, 1, list(range(n))) == [list(range(n))]
        capi.comm_destroy(comm)
    finally:
        ctx.close()


def test_self_check_failure_exits_non_zero():
    """a line whose totals contradict the generator's closed form must not exit 0 (GT4_BENCH_BREAK_CHECK: test hook)"""
    env = dict(os.environ, GT4_BENCH_BREAK_CHECK="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL_PAIR, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 3, (p.returncode, p.stderr[-1000:])
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][0]
    assert json.loads(line)["self_check"].startswith("FAILED")


def test_two_ranks_on_one_device_strong_scaling_control_flow():
    """both ranks on device 0, reductions over gloo: the sharding arithmetic, the per-step totals
    exchange and the report, without RCCL"""
    one = _bench(1, SMALL_PAIR)
    two = _bench(2, SMALL_PAIR, {"GT4_BENCH_ONE_DEVICE": "1"})
    assert two["scaling"] == "strong" and two["n_gpus"] == 2
    assert two["config"]["output_records"] == one["config"]["output_records"]["intrsec"]
    assert two["config"]["output_total_count"] == one["config"]["output_total_count"]["intrsec"]
    shards = two["config"]["per_rank"]
    assert sum(r["shard_input_records"] for r in shards) == 2 * 6000000


@pytest.mark.skipif("_n_devices() < 2", reason="needs two GPUs")
def test_two_gpus_strong_scaling_intersection_over_rccl():
    one = _bench(1, SMALL_PAIR)
    two = _bench(2, SMALL_PAIR)
    assert two["scaling"] == "strong" and two["self_check"] == "ok"
    assert two["config"]["output_records"] == one["config"]["output_records"]["intrsec"]
    assert two["config"]["output_total_count"] == one["config"]["output_total_count"]["intrsec"]
    assert sum(r["shard_input_records"] for r in two["config"]["per_rank"]) == 2 * 6000000


@pytest.mark.skipif("_n_devices() < 2", reason="needs two GPUs")
def test_two_gpus_eight_way_union_with_rccl_gatherv():
    one = _bench(1, SMALL_UNION)
    two = _bench(2, SMALL_UNION)
    assert two["config"]["output_records"] == one["config"]["output_records"]
    assert two["config"]["output_total_count"] == one["config"]["output_total_count"]
    assert two["config"]["gathered_bytes_per_step"] > 0 and two["self_check"] == "ok"
    assert sum(r["shard_input_records"] for r in two["config"]["per_rank"]) == 8 * 1500000


@pytest.mark.skipif("_n_devices() < 2", reason="needs two GPUs")
def test_two_gpus_default_line_over_rccl():
    one = _bench(1, SMALL_BOTH)
    two = _bench(2, SMALL_BOTH)
    assert two["self_check"] == "ok" and two["union8"]["self_check"] == "ok" and two["union8"]["gathered_bytes"] > 0
    assert (two["union8"]["output_records"], two["union8"]["output_total_count"]) == (one["union8"]["output_records"], one["union8"]["output_total_count"])


def test_default_line_at_one_gpu_carries_config2_the_projection_and_the_file_to_file_run():
    """VERDICT round 5, next 2 b: what only the builder's own runs showed is in the driver's line -- BASELINE configs[2] on the
    same resident pair (verified against the reference binary), the eight shards of the union one after another, the
    C command-line tool file -> file against the reference binary (byte-identical outputs), file-writing CPU leg = median of 3"""
    r = _bench(1, ["--entries", "6000000", "--entries8", "1500000", "--steps", "2", "--warmup", "1", "--cpu-sample", "4000000", "--e2e-n", "1000000"])
    assert r["self_check"] == "ok" and r["verified"] is True
    assert "median of 3" in r["cpu_baseline"]["file_writing"]["sample"]
    c2 = r["c2"]
    assert "error" not in c2 and c2["verified"] is True and c2["roofline"]["frac"] > 0 and set(c2["output_records"]) == {"union", "diff1"}
    e = r["e2e"]
    assert "error" not in e and e["verified"] is True and [x["flags"] for x in e["runs"]] == ["-i", "-u -i -d"]
    assert all(x["byte_identical"] and x["gpu_s"] > 0 and x["reference_s"] > 0 for x in e["runs"]) and e["runs"][1]["output_files"] == 3
    pr = r["shard_projection"]
    assert pr["shards"] == 8 and pr["outputs_add_up"] and len(pr["projection"][0]["per_shard"]) == 8 and pr["projected_speedup"] > 0
    assert r["union8"]["self_check"] == "ok" and "error" not in r["union8"]


SMALL_GATHER = ["--entries", "3000000", "--entries8", "800000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]


@pytest.mark.parametrize("ranks", [2, 8])
def test_a_gather_that_fails_at_first_contact_falls_back_on_every_rank(ranks):
    """GT4_BENCH_BREAK_GATHER=1: the C gather raises on ONE rank before it enters the collective; one all_reduce makes the
    verdict common, all ranks switch to torch.distributed's send / recv together, the line says so"""
    one = _bench(1, SMALL_GATHER + ["--no-extras"])
    r = _bench(ranks, SMALL_GATHER, {"GT4_BENCH_ONE_DEVICE": "1", "GT4_BENCH_BREAK_GATHER": "1"})
    assert r["self_check"] == "ok" and r["config"]["output_records"] == one["config"]["output_records"]["intrsec"]
    u = r["union8"]
    assert u["self_check"] == "ok" and "error" not in u and u["merge_only"] > 0 and u["value_with_gather"] > 0
    assert "torch.distributed" in u["gather_path"] and "first contact" in u["gather_note"] and u["gathered_bytes"] > 0
    assert (u["output_records"], u["output_total_count"]) == (one["union8"]["output_records"], one["union8"]["output_total_count"])


@pytest.mark.parametrize("ranks", [2])
def test_a_gather_that_fails_on_both_paths_costs_only_its_own_number(ranks):
    """GT4_BENCH_BREAK_GATHER=2: the intersection line and merge_only are there, union8 carries the error, exit code 0"""
    r = _bench(ranks, SMALL_GATHER, {"GT4_BENCH_ONE_DEVICE": "1", "GT4_BENCH_BREAK_GATHER": "2"})
    assert r["self_check"] == "ok" and r["value"] > 0
    u = r["union8"]
    assert "both paths" in u["error"] and u["merge_only"] > 0 and u["self_check"] == "ok" and u["gathered_bytes"] == 0


def test_a_gather_that_hangs_is_cut_off_with_the_line_printed():
    """GT4_BENCH_BREAK_GATHER=3: one rank never arrives in the gather.  The leg's wall-clock guard prints the line as far
    as the run got -- the intersection and the union's merge_only -- and every rank leaves with exit code 4: no hang"""
    rc, r, err = _bench_raw(2, SMALL_GATHER + ["--leg-timeout", "25"], {"GT4_BENCH_ONE_DEVICE": "1", "GT4_BENCH_BREAK_GATHER": "3"})
    assert rc != 0 and "TIMEOUT" in err
    assert r["self_check"] == "ok" and r["value"] > 0
    assert "wall-clock bound" in r["union8"]["error"] and r["union8"]["merge_only"] > 0


def test_device_lists_as_torch_tensors_without_a_copy():
    """the fall-back of the payload gather (torch.distributed send / recv) works on VIEWS of the lists' HBM
    (distributed.list_as_tensor: the CUDA array interface): what torch sees is what the list holds, and a write through
    the view lands in the list"""
    import numpy as np
    import torch
    from genometester4_amd import capi, distributed as D
    from genometester4_amd.listio import RECORD_DTYPE, make_records
    ctx = capi.Context(0)
    try:
        rng = np.random.default_rng(5)
        keys = np.unique(rng.integers(0, 1 << 50, size=20000, dtype=np.uint64))
        rec = make_records(keys, rng.integers(1, 9, size=len(keys), dtype=np.uint32))
        lst = ctx.upload(rec, 25)
        t = D.list_as_tensor(lst, len(rec))
        assert t.is_cuda and t.dtype == torch.int32 and t.numel() == 3 * len(rec) and t.data_ptr() == lst.device_ptr
        assert t.cpu().numpy().view(RECORD_DTYPE).tobytes() == rec.tobytes()
        dst = ctx.alloc(len(rec), 25)
        D.list_as_tensor(dst, len(rec)).copy_(t)
        torch.cuda.synchronize()
        assert dst.download_range(0, len(rec)).tobytes() == rec.tobytes()
        assert D.list_as_tensor(lst, 0).numel() == 0
        h = D.list_as_tensor(lst, 100, host=True)
        assert not h.is_cuda and h.numpy().view(RECORD_DTYPE).tobytes() == rec[:100].tobytes()
    finally:
        ctx.close()
