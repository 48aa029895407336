#!/usr/bin/env python3
"""bench.py -- k-mers merged/sec of the glistcompare intersection hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one complete two-list intersection (merge-path partition + merge kernel + totals
read-back) over synthetic sorted lists that are already resident in HBM.  Workload at N=1 is
BASELINE.json configs[1]: two 2x10^9-entry k=25 lists (~24 GB each), `glistcompare -i`.  With N>1 the
default is the SAME job (`"scaling": "strong"`): every rank keeps its key range of the one pair
(cut at keys sampled from the lists, gt4hip_shard_cuts; the set operations are key-local), merges it, and the per-shard
(n_words, total_count) header totals are all-gathered over RCCL inside every step.  `--scaling weak`
gives every GPU an independent pair instead (no exchange; a reference line, not the metric's job).

The default line also carries a `"union8"` record -- BASELINE configs[3], the 8-way union of eight
5x10^8-entry lists as ONE job sharded by key range over the ranks: `merge_only` (what the shards sustain when every
rank keeps / writes its own extent, the C host's default; measured FIRST) and `value_with_gather` (the gatherv of the
payload to rank 0 inside the step: gt4hip_comm_gatherv over RCCL, or -- if that fails on any rank at first contact -- on
every rank together torch.distributed's send / recv; `gather_path` says which).  north_star asks for a final gatherv AND
for >= 6x at 8 GPUs; the two exclude each other, so both numbers are reported.  At N = 1 the line
further embeds `"c2"` (BASELINE configs[2] on the same resident pair, verified against the reference binary),
`"shard_projection"` (the union's eight key-range shards timed one after another: what 8 GPUs would make of them) and
`"e2e"` (the C command-line tool file -> file against the reference binary on a 2 x 2e8 sample in /dev/shm, outputs
byte-compared).  Nothing can lose the line: a leg that raises leaves an "error" in its record, a leg that exceeds
--leg-timeout makes rank 0 print the line as far as the run got and every rank exit 4; stdout carries the one JSON line
and nothing else.  Every line checks its job totals against the closed forms of the generator (|A n B| = n / 2, 5 n
distinct keys of the eight lists) and the totals committed for the default sizes, and exits 3 on a mismatch.

`--dist` chooses the key distribution of the synthetic lists (genometester4_amd/synth.py): stride
(default; one key per stride of the key space), iid, clustered, genomic.

Prints ONE JSON line on rank 0.  `roofline` is computed from the merge kernel's own HIP-event time
(recorded by the library on the stream it launches on); `cpu_baseline` times the REFERENCE binary
(oracle/_ref/glistcompare, `kind: reference`) or, when that is absent, the C oracle (`kind: port`)
on a bounded prefix sample of the same lists -- `--count_only` (the pure merge loop) and, on a
smaller prefix, the file-writing variant (BASELINE.md protocol).  `verified`: after the timed loop
the GPU runs the same operation on exactly the CPU sample and its (n_words, total_count) must equal
what the reference binary printed.

`--workload c2` is BASELINE configs[2]: union + first complement with cutoff 3 on the same pair
(the any-combination kernel, two simultaneous output streams).
"""
import argparse
import json
import os
import shutil
import statistics
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ---- nothing below may lose the line: what has been measured so far, a watchdog per leg, one place that prints

PROGRESS = {"headline": None, "union8": {}}   # what rank 0 would print if the run ended now
_EMIT_LOCK = threading.Lock()
_EMITTED = []
_LINE_FD = [None]


def own_stdout():
    """stdout carries the ONE JSON line and nothing else: RCCL (2.26, the build torch bundles) prints a version banner on
    the C stdout of every process that makes a communicator, whatever NCCL_DEBUG says, and flushes it at exit -- behind
    the line.  The line goes to a private duplicate of the original descriptor; descriptor 1 itself is pointed at stderr."""
    if _LINE_FD[0] is None:
        sys.stdout.flush()
        _LINE_FD[0] = os.dup(1)
        os.dup2(2, 1)


def emit(res):
    """the ONE JSON line of the run (rank 0); whoever comes second -- the main flow or the watchdog -- prints nothing"""
    with _EMIT_LOCK:
        if _EMITTED:
            return False
        _EMITTED.append(True)
        line = (json.dumps(res) + "\n").encode()
        if _LINE_FD[0] is None:
            sys.stdout.write(line.decode())
            sys.stdout.flush()
        else:
            while line:
                line = line[os.write(_LINE_FD[0], line):]
        return True


def partial_line(why):
    """the line as far as the run got, for the watchdog and for a leg that raised"""
    res = PROGRESS["headline"]
    if res is None:
        res = {"metric": "k-mers merged/sec, 2-list k=25 intersection (glistcompare -i), lists resident in HBM", "value": None, "unit": "k-mers/s",
               "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "higher_is_better": True, "error": why}
    else:
        res = dict(res)
    if PROGRESS["union8"] and "union8" not in res:
        res["union8"] = dict(PROGRESS["union8"], error=why)
    elif "union8" not in res and why:
        res["union8"] = {"error": why}
    return res


class Guard:
    """A wall-clock bound per leg.  A collective that never returns (a peer that died, a gather that hangs at first
    contact) cannot be interrupted from Python: when a leg overruns, rank 0 prints the line with what has been measured
    so far and an "error" that names the leg, and every rank leaves with exit code 4 -- no re-exec, no hang."""

    def __init__(self, rank):
        self.rank = rank
        self.leg, self.deadline = None, None
        self.lock = threading.Lock()
        t = threading.Thread(target=self._watch, daemon=True)
        t.start()

    def arm(self, leg, seconds):
        with self.lock:
            self.leg, self.deadline = leg, time.monotonic() + seconds

    def disarm(self):
        with self.lock:
            self.leg, self.deadline = None, None

    def _watch(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                leg, dl = self.leg, self.deadline
            if dl is not None and time.monotonic() > dl:
                why = "leg '%s' exceeded its wall-clock bound on rank %d" % (leg, self.rank)
                try:
                    log("TIMEOUT: " + why)
                    if self.rank == 0:
                        for _ in range(3):  # (the main thread may be writing into the records this very moment)
                            try:
                                emit(partial_line(why))
                                break
                            except RuntimeError:
                                time.sleep(0.05)
                    sys.stdout.flush()
                finally:
                    os._exit(4)


class _NoGuard:
    def arm(self, *a):
        pass

    def disarm(self):
        pass


GUARD = _NoGuard()


def _agree(ok):
    """logical AND over the ranks: every rank takes the same branch behind it"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_xdev())
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def build_lists(ctx, capi, n, k, seed_base, dist="stride"):
    """A and B of about n records, half of each shared (genometester4_amd/synth.py; `stride`: exactly n
    and |A n B| = n // 2 from three disjoint residue classes)."""
    from genometester4_amd import synth
    return synth.make_pair(ctx, n, k, dist, seed_base)


# job totals of the default sizes, committed with the round-3 profiles (profiles/round3/r3final_*_bench.json)
EXPECTED_TOTALS = {
    ("intersect", "stride", 2_000_000_000, 25): (1_000_000_000, 3_187_480_123),
    ("union8", "stride", 500_000_000, 25): (2_500_000_000, 17_999_789_507),
}


def self_check(kind, dist, n, k, n_words, total_count):
    """Closed forms of the stride generator + the committed totals of the default sizes -> None or a message."""
    if dist == "disjoint":
        want = MULTI[kind]["lists"] * n if kind in MULTI and MULTI[kind]["op"] == "union" else None
        return None if want is None or n_words == want else "union holds %d records, generator says %d" % (n_words, want)
    if dist != "stride":
        return None
    if os.environ.get("GT4_BENCH_BREAK_CHECK"):  # test hook: the failure path itself
        n_words += 1
    if kind == "intersect" and n_words != n // 2:
        return "intersection holds %d records, generator says %d" % (n_words, n // 2)
    if kind in ("union8", "union32") and n_words != (MULTI[kind]["lists"] // 2 + 1) * n:
        return "union holds %d records, generator says %d" % (n_words, (MULTI[kind]["lists"] // 2 + 1) * n)
    if kind == "intersect8" and n_words != n // 2:
        return "intersection holds %d records, generator says %d" % (n_words, n // 2)
    exp = EXPECTED_TOTALS.get((kind, dist, n, k))
    if exp and (n_words, total_count) != exp:
        return "%s totals (%d, %d) differ from the committed N=1 totals %s" % (kind, n_words, total_count, exp)
    return None


def _ref_flags(ops, cutoff):
    f = []
    if ops & 1:
        f.append("-u")
    if ops & 2:
        f.append("-i")
    if ops & 4:
        f.append("-d")
    if cutoff != 1:
        f += ["-c", str(cutoff)]
    return f


def _parse_count_only(stdout):
    """`NUnique\t<n>\nNTotal\t<t>\n` per requested op, in the order union, intrsec, diff1 (reference
    src/glistcompare.c:916-952) -> [(n, t), ...]"""
    vals = [int(line.split("\t")[1]) for line in stdout.strip().split("\n") if "\t" in line]
    return list(zip(vals[0::2], vals[1::2]))


def cpu_baseline(ctx, capi, a, b, k, sample_records, ops, cutoff):
    """Times the reference CPU path on a prefix sample covering the same key range of both lists,
    then runs the GPU on exactly that sample and compares the totals (`verified`)."""
    from genometester4_amd.listio import write_list
    m_a = min(sample_records, a.n_words)
    last_key, _ = a.get_word(m_a - 1)
    m_b = b.lower_bound(last_key + 1) if last_key < 0xFFFFFFFFFFFFFFFF else b.n_words
    ha, hb = a.download_range(0, m_a), b.download_range(0, m_b)
    flags = _ref_flags(ops, cutoff)
    # the GPU on the very same sample (device slices of the resident lists)
    st, _, _ = ctx.compare(a.slice(0, m_a), b.slice(0, m_b), ops, cutoff=cutoff, count_only=True)
    gpu_totals = [st[bit] for bit in (1, 2, 4) if ops & bit]
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "glistcompare")
    sample = "first %d + %d records of the same two lists (equal key range), %s --count_only, warm page cache" % (m_a, m_b, " ".join(flags))
    nproc = os.cpu_count()
    if os.path.exists(ref_bin) and os.access(ref_bin, os.X_OK):
        shm = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 40 * (m_a + m_b) else None
        d = tempfile.mkdtemp(prefix="gt4bench_", dir=shm)
        try:
            write_list(os.path.join(d, "a.list"), ha, k)
            write_list(os.path.join(d, "b.list"), hb, k)
            times = []
            for _ in range(3):
                t0 = time.perf_counter()
                r = subprocess.run([ref_bin, "a.list", "b.list"] + flags + ["--count_only"], cwd=d, capture_output=True)
                times.append(time.perf_counter() - t0)
                if r.returncode != 0:
                    raise RuntimeError("reference glistcompare failed: %s" % r.stderr.decode())
            t = statistics.median(times)
            ref_totals = _parse_count_only(r.stdout.decode())
            # the file-writing variant (output fwrite/write syscalls dominate the reference):
            # a 10x smaller prefix, files in the same directory
            w_a, w_b = max(1, m_a // 10), max(1, m_b // 10)
            write_list(os.path.join(d, "wa.list"), ha[:w_a], k)
            write_list(os.path.join(d, "wb.list"), hb[:w_b], k)
            tw = []
            for _ in range(3):
                t0 = time.perf_counter()
                rw = subprocess.run([ref_bin, "wa.list", "wb.list"] + flags + ["-o", "w"], cwd=d, capture_output=True)
                tw.append(time.perf_counter() - t0)
            t_w = statistics.median(tw)
            out_bytes = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.startswith("w_"))
            res = dict(value=(m_a + m_b) / t, unit="k-mers/s", cores=3, host_nproc=nproc, kind="reference",
                       sample=sample + "; reference glistcompare 4.2.16: 1 merge thread + 2 scout threads (all it can use of the "
                                       "host's %d logical CPUs); median of 3 runs" % nproc,
                       stdout=r.stdout.decode().strip().replace("\n", " ").replace("\t", "="),
                       file_writing=dict(value=(w_a + w_b) / t_w if rw.returncode == 0 else None, unit="k-mers/s",
                                         sample="first %d + %d records, %s writing %d output bytes to %s, median of 3 runs"
                                                % (w_a, w_b, " ".join(flags), out_bytes, "tmpfs" if shm else "the temp dir")))
            return res, ref_totals == gpu_totals, dict(reference=ref_totals, gpu=gpu_totals)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        exp = O.compare(ha, hb, ops, cutoff=cutoff, count_only=True)
        times.append(time.perf_counter() - t0)
    ora_totals = [exp[bit][:2] for bit in (1, 2, 4) if ops & bit]
    res = dict(value=(m_a + m_b) / statistics.median(times), unit="k-mers/s", cores=1, host_nproc=nproc, kind="port",
               sample=sample + "; oracle/gt4_oracle.c scalar restatement, median of 3 runs")
    return res, ora_totals == gpu_totals, dict(oracle=ora_totals, gpu=gpu_totals)


def e2e_leg(args, ctx, a, b):
    """File -> file (BASELINE.md's end-to-end protocol): the C command-line tool of the product
    (genometester4_amd/glistcompare: mmap + copy threads -> HBM -> merge -> pinned -> pwrite, tmp + rename) against the
    reference binary (oracle/_ref/glistcompare) on a prefix sample of the resident pair, files in /dev/shm (warm page
    cache), `-i` and `-u -i -d`, one run each; every output file byte-compared with the reference's."""
    from genometester4_amd.listio import write_list
    cli = os.path.join(ROOT, "genometester4_amd", "glistcompare")
    ref = os.path.join(ROOT, "oracle", "_ref", "glistcompare")
    for exe in (cli, ref):
        if not (os.path.exists(exe) and os.access(exe, os.X_OK)):
            raise RuntimeError("%s is not there" % os.path.relpath(exe, ROOT))
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (1 << 30) else None
    m_a = min(args.e2e_n, a.n_words)
    free = shutil.disk_usage(shm or tempfile.gettempdir()).free
    while m_a > 1_000_000 and 12 * 2 * m_a * 8 > free:  # inputs + both tools' three outputs, with room to spare
        m_a //= 2
    last_key, _ = a.get_word(m_a - 1)
    m_b = b.lower_bound(last_key + 1) if last_key < 0xFFFFFFFFFFFFFFFF else b.n_words
    d = tempfile.mkdtemp(prefix="gt4e2e_", dir=shm)
    runs = []
    try:
        write_list(os.path.join(d, "a.list"), a.download_range(0, m_a), args.k)
        write_list(os.path.join(d, "b.list"), b.download_range(0, m_b), args.k)
        for flags in (["-i"], ["-u", "-i", "-d"]):
            rec = {"flags": " ".join(flags)}
            for who, exe in (("gpu", cli), ("reference", ref)):
                t0 = time.perf_counter()
                p = subprocess.run([exe, "a.list", "b.list"] + flags + ["-o", who], cwd=d, capture_output=True, timeout=args.leg_timeout)
                rec[who + "_s"] = time.perf_counter() - t0
                if p.returncode != 0:
                    raise RuntimeError("%s %s failed (%d): %s" % (who, " ".join(flags), p.returncode, p.stderr.decode()[-300:]))
            ours = sorted(f for f in os.listdir(d) if f.startswith("gpu_"))
            theirs = sorted(f for f in os.listdir(d) if f.startswith("reference_"))
            same = len(ours) == len(theirs) == len(flags)
            out_bytes = 0
            for f, g in zip(ours, theirs):
                same = same and f[len("gpu_"):] == g[len("reference_"):]
                same = same and subprocess.run(["cmp", "-s", f, g], cwd=d).returncode == 0
                out_bytes += os.path.getsize(os.path.join(d, f))
            for f in ours + theirs:
                os.remove(os.path.join(d, f))
            rec.update(output_files=len(ours), output_bytes=out_bytes, byte_identical=bool(same), speedup=rec["reference_s"] / rec["gpu_s"],
                       gpu_k_mers_per_s=(m_a + m_b) / rec["gpu_s"], reference_k_mers_per_s=(m_a + m_b) / rec["reference_s"])
            runs.append(rec)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return {"workload": "glistcompare a.list b.list <flags> -o <name>: first %d + %d records of the pair (equal key range), %.1f GB of input files in %s" % (m_a, m_b, 12 * (m_a + m_b) / 1e9, "tmpfs (/dev/shm)" if shm else "the temp dir"),
            "runs": runs, "verified": all(r["byte_identical"] for r in runs),
            "note": "wall time of the whole process, one run each, warm page cache; the product's time includes ~0.3 s of HIP start-up and the PCIe transfers both ways; "
                    "the reference is single-threaded for the merge (+ 2 scout threads)"}


def csrc_sha16():
    """What the replayed PMC traffic is tied to: a hash over the device sources and their build flags (csrc/*.hip, *.h, Makefile) as they lie in
    the tree -- the GPU box has no .git.  tools/summarize_profiles.py records the same value with the passes."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "genometester4_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "Makefile"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_traffic(workload, n, kernel=None):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/
    (tools/collect_profiles.sh: FETCH_SIZE and WRITE_SIZE cannot be collected inside this run).
    REFUSED (traffic stays unmeasured, the reason is in traffic_source) when the kernel sources have changed since
    the passes ran, or the recorded dominant kernel is not the one this run names."""
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    if not os.path.exists(tpath) and workload == "intersect":
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        tj = json.load(open(tpath))
    except Exception:
        return None, None
    src = {"file": os.path.relpath(tpath, ROOT), "commit": tj.get("commit"), "kernel": tj.get("kernel"), "csrc_sha16": tj.get("csrc_sha16")}
    if tj.get("n_per_list") != n:
        return None, None
    now = csrc_sha16()
    if tj.get("csrc_sha16") != now:
        src["note"] = "REFUSED: the PMC passes ran on kernel sources %s, this tree has %s (re-run tools/collect_profiles.sh)" % (tj.get("csrc_sha16"), now)
        return None, src
    if kernel and tj.get("kernel") and tj["kernel"].split("<")[0].replace("k_nway32_", "k_nway_") != kernel.split("<")[0].split(" ")[0].replace("km32::", ""):
        src["note"] = "REFUSED: the PMC passes name %s as the dominant kernel, this run %s" % (tj.get("kernel"), kernel)
        return None, src
    src["note"] = "replayed from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE), not measured in this run"
    return tj.get("hbm_bytes_per_launch"), src


MULTI = {
    # kind: lists, default entries per list, operation
    "union8": dict(lists=8, op="union", what="8-way k=%d union (MakeUnion.pl replacement)", ref="scripts/MakeUnion.pl:31-95, src/glistcompare.c:545-591"),
    "union32": dict(lists=32, op="union", what="32-way k=%d union (glistmaker's collation width, gt4_write_union)", ref="src/glistmaker.c:787-835, src/set-operations.c:40-129"),
    "intersect8": dict(lists=8, op="intersect", what="8-way k=%d intersection (MakeIntersection.pl replacement)", ref="scripts/MakeIntersection.pl, src/glistcompare.c:605-717"),
}


def multi_roofline(ctx, kind, n_in_local, n_out_local, device_ms, kernel_ms, one_pass, workload_n, dist="stride", moved=None, levels=False):
    """Rank 0's shard: algorithmic bytes (every input record read once, every output record written once).
    `frac` is over the average launch of the dominant kernel where ONE kernel does the work (the one-pass N-way
    tile kernel, HIP events on the library's stream); `whole_call_frac` over the whole call on the device (key
    samples, their merges, tile partition, every launch) -- the only figure where the work is a chain or a tree
    of pair-kernel launches, and then `frac` equals it."""
    rd, wr = moved if moved else ctx.last_multi_records
    alg = 12 * (n_in_local + n_out_local)
    t_ms = kernel_ms if one_pass and kernel_ms > 0 else device_ms
    achieved = alg / (t_ms * 1e-3) / 1e9
    if one_pass:
        kernel = "k_nway_merge<1024, 4, 1, NWAY_UNION> (one pass over up to eight lists per launch)"
        if MULTI[kind]["lists"] > 8:
            kernel = "km32::k_nway_merge<1024, 4, 1, NWAY_UNION> (one pass over up to 32 lists per launch: runs end to end in the tile, samples every 64 records)"
    elif levels:
        kernel = "k_nway_merge<1024, 4, 1, NWAY_UNION> (levels of eight-way passes: every record goes through the tile kernel twice)"
    elif MULTI[kind]["op"] == "intersect":
        kernel = "k_pair_merge<1024, 6, MODE_LOOKBACK, intersection> (left-to-right chain, one launch per list after the first)"
    else:
        kernel = "k_pair_merge<1024, 4, 1, 1> (pairwise union tree)"
    traffic, traffic_source = load_traffic(kind if dist == "stride" else "%s_%s" % (kind, dist), workload_n, kernel)
    return {"bound": "hbm", "kernel": kernel,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "frac_definition": "algorithmic bytes / average launch of the tile kernel" if one_pass and kernel_ms > 0 else "algorithmic bytes / device time of the whole call",
            "traffic": traffic if traffic is not None else 12 * (rd + wr), "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": alg, "kernel_ms_avg": t_ms, "device_ms_avg": device_ms,
            "whole_call_frac": alg / (device_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "per shard (rank 0); device_ms_avg = the whole call on the device; without committed PMC passes traffic = 12 x the records the library read and wrote (intermediate levels included)"}


def _xdev():
    """device of the tensors that go through torch.distributed (the one-device test hook runs over gloo)"""
    import torch.distributed as dist
    return "cuda" if (not dist.is_initialized() or dist.get_backend() == "nccl") else "cpu"


def multi_cpu_leg(args, ctx, capi, kind, full):
    """The reference on a key window of all the lists (files in tmpfs): `glistcompare L1 .. LN -u / -i --count_only`
    (union_multi / intersect_multi, src/glistcompare.c:500-717) -- for union32 `ref_setops write_union` (the reference's
    set-operations.c:40-129 behind this repo's driver) -- or, where the binaries are absent, the oracle's loop.
    The GPU's totals on the same window must agree."""
    from genometester4_amd.listio import write_list
    spec = MULTI[kind]
    m = min(full[0].n_words, max(1000, args.cpu_sample // (4 * spec["lists"])))
    last_key, _ = full[0].get_word(m - 1)
    cuts = [l.lower_bound(last_key + 1) for l in full]
    host = [l.download_range(0, c) for l, c in zip(full, cuts)]
    n_rec = sum(len(h) for h in host)
    dev = [ctx.upload(h, args.k) for h in host]
    fn = ctx.union_multi if spec["op"] == "union" else ctx.intersect_multi
    rc_g, n_g, t_g, _ = fn(dev, count_only=True)
    for d in dev:
        d.free()
    sample = "first %d records of list 0 and the same key range of the other %d lists (%d records)" % (m, spec["lists"] - 1, n_rec)
    ref_cmp = os.path.join(ROOT, "oracle", "_ref", "glistcompare")
    ref_set = os.path.join(ROOT, "oracle", "_ref", "ref_setops")
    exe = ref_set if kind == "union32" else ref_cmp
    if os.path.exists(exe) and os.access(exe, os.X_OK):
        shm = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 40 * n_rec else None
        d = tempfile.mkdtemp(prefix="gt4bench_", dir=shm)
        try:
            names = []
            for j, h in enumerate(host):
                names.append("l%d.list" % j)
                write_list(os.path.join(d, names[-1]), h, args.k)
            if kind == "union32":
                cmd = [exe, "write_union", "1", "out.list"] + names
            else:
                cmd = [exe] + names + ["-u" if spec["op"] == "union" else "-i", "--count_only"]
            times = []
            for _ in range(3):
                t0 = time.perf_counter()
                r = subprocess.run(cmd, cwd=d, capture_output=True)
                times.append(time.perf_counter() - t0)
                if r.returncode != 0:
                    raise RuntimeError("reference run failed: %s" % r.stderr.decode()[-300:])
            ref_tot = _parse_count_only(r.stdout.decode())
            ok = bool(rc_g == 0 and ref_tot and ref_tot[0] == (n_g, t_g))
            return ({"value": n_rec / statistics.median(times), "unit": "k-mers/s", "cores": 1 if kind == "union32" else 1 + spec["lists"], "host_nproc": os.cpu_count(), "kind": "reference",
                     "sample": sample + "; " + " ".join(cmd[:2] if kind == "union32" else [os.path.basename(exe), "L1 .. L%d" % spec["lists"]] + cmd[-2:])
                               + ", files in %s, warm page cache, median of 3 runs; %s" % ("tmpfs" if shm else "the temp dir", "one thread (no scouts)" if kind == "union32" else "one merge thread + one scout thread per list"),
                     "stdout": r.stdout.decode().strip().replace("\n", " ").replace("\t", "=")}, ok)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    t0c = time.perf_counter()
    rc_o, n_o, t_o, _ = (O.union_multi if spec["op"] == "union" else O.intersect_multi)(host, 1, 0, 1)
    dtc = time.perf_counter() - t0c
    return ({"value": n_rec / dtc, "unit": "k-mers/s", "cores": 1, "host_nproc": os.cpu_count(), "kind": "port",
             "sample": sample + "; oracle/gt4_oracle.c %s_multi, one thread" % spec["op"]}, bool(rc_o == 0 and rc_g == 0 and (n_g, t_g) == (n_o, t_o)))


def bench_multi(args, ctx, capi, rank, local_rank, world, kind="union8", progress=None, project=0):
    """The multi-list workloads, each ONE job sharded by key range over the ranks (strong scaling): every rank keeps
    its key range of every list resident in HBM and runs the operation on its shards, the header totals are
    all-gathered and the payload is gathered on rank 0 over RCCL (gt4hip_comm_gatherv of the C ABI: grouped ncclSend
    / ncclRecv -- the entry point the C command-line tool uses); rank 0 merges straight into the gathered list (its
    range is the first extent).
      union8      BASELINE configs[3]: eight lists (MakeUnion.pl replacement, reference scripts/MakeUnion.pl:31-95), one
                  pass of the N-way tile kernel -- or the pairwise tree where the library finds the keys clustered (`--tree`
                  forces it)
      union32     glistmaker's collation width (src/glistmaker.c:787-835): four eight-way passes, then a four-way one
      intersect8  MakeIntersection.pl's job (src/glistcompare.c:605-717): the left-to-right chain of pair intersections
    `progress`: a dict that receives what is known as soon as it is known (merge_only before the gather is tried).
    `project` = N (one GPU): the job's N key-range shards are also timed one after another (record "shard_projection").
    Returns the result line (rank 0) or None."""
    import torch
    import torch.distributed as dist
    from genometester4_amd import distributed as D
    from genometester4_amd import synth
    progress = progress if progress is not None else {}
    spec = MULTI[kind]
    n_lists = spec["lists"]
    n8 = {"union8": args.n8, "union32": args.n32, "intersect8": args.n8}[kind]
    if args.tree:
        ctx.set_option("kway", 0)
    full = synth.make_lists8(ctx, n8, args.k, args.dist, n_lists) if spec["op"] == "union" else synth.make_lists_shared(ctx, n8, args.k, args.dist, n_lists)
    n_in = sum(l.n_words for l in full)
    comm_id = None
    if world > 1 and _xdev() == "cuda":
        box = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm_id = box[0]
    sh = D.DeviceShards(ctx, rank, world, comm_id, agree=_agree)
    sh.host_tensors = _xdev() == "cpu"  # (the one-device test hook: torch.distributed runs over gloo)
    if world > 1 and os.environ.get("GT4_BENCH_BREAK_GATHER") in ("1", "2"):
        sh.gather_via = "rccl"  # (test hook: the first attempt is the C gather, which the hook makes fail)
    cuts = sh.plan(full, sampled=args.splitters == "sampled")
    shards = [sh.shard_of(l, args.k) for l in full]
    n_local_in = sum(s.n_words for s in shards)
    # rank 0 of a sharded job writes its result into the list the payload is gathered in: its extent is the first
    root_direct = world > 1 and rank == 0 and not sh.host_tensors
    worst_local = n_local_in if spec["op"] == "union" else min(s.n_words for s in shards)
    worst_job = n_in if spec["op"] == "union" else min(l.n_words for l in full)
    out = ctx.alloc(max(1, worst_job if root_direct else worst_local), args.k)
    op = D.gpu_union_multi_op(ctx) if spec["op"] == "union" else D.gpu_intersect_multi_op(ctx)

    def totals_exchange(n, total):
        return D.exchange_totals(n, total, device=_xdev())

    def step(gather=True):
        n, total, res, totals = sh.run(shards, op, totals_exchange, root=0, out=out, gathered=out if root_direct else None, gather=gather)
        return n, total, totals, dict(sh.last_ms)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- FIRST what every rank sustains on its own extent (merge_only: no payload moves; what BASELINE's >= 6x is read
    # on) -- so that nothing the gather does at first contact can lose it
    for _ in range(args.warmup):
        step(gather=False)
    fence()
    dev_ms, ker_ms, ms0 = [], [], []
    t1 = time.perf_counter()
    for _ in range(args.steps):
        n_out, total_out, totals, m = step(gather=False)
        ms0.append(m)
        dev_ms.append(ctx.last_multi_device_ms)
        ker_ms.append(ctx.get_counter("nway_kernel_us") / 1000.0)
    fence()
    merge_only = time.perf_counter() - t1
    moved = ctx.last_multi_records  # records the library read and wrote in the last timed call (before the CPU leg's small calls)
    if world > 1:
        t = torch.tensor([merge_only], dtype=torch.float64, device=_xdev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        merge_only = float(t[0].item())
    progress.update({"workload": "%s, %d lists of %d entries, ONE job over %d GPU(s)" % (spec["what"] % args.k, n_lists, n8, world), "n_gpus": world,
                     "merge_only": n_in * args.steps / merge_only, "merge_only_ms_per_step": merge_only / args.steps * 1e3, "unit": "k-mers/s",
                     "output_records": n_out, "output_total_count": total_out})
    # ---- THEN the same steps with the payload gathered on rank 0.  First contact of the gather is its warm-up: every
    # rank reports whether its call came back, one all_reduce makes the verdict common, and on a failure of the C
    # gather (gt4hip_comm_gatherv) all ranks switch to torch.distributed's send / recv together and say so.
    gather_error = None
    elapsed = merge_only
    ms = ms0
    if world > 1:
        attempts = 0
        done = 0
        while done < max(1, args.warmup):
            err = None
            try:
                step(gather=True)
            except Exception as e:
                err = "%s: %s" % (type(e).__name__, e)
            if _agree(err is None):
                done += 1
                continue
            attempts += 1
            if sh.gather_via == "rccl" and attempts == 1:
                sh.use_torch_gather("gt4hip_comm_gatherv failed on a rank at first contact (%s): payload over torch.distributed send / recv instead" % (err or "on another rank"))
                log("rank %d: %s" % (rank, sh.gather_note))
                continue
            gather_error = "the payload gather failed on both paths (this rank: %s)" % (err or "ok, another rank failed")
            break
        if gather_error is None:
            fence()
            t0 = time.perf_counter()
            ms = []
            for _ in range(args.steps):
                n_out, total_out, totals, m = step(gather=True)
                ms.append(m)
            fence()
            elapsed = time.perf_counter() - t0
    one_pass = spec["op"] == "union" and bool(ctx.get_counter("nway_one_pass"))
    per_rank = [{"rank": rank, "shard_input_records": n_local_in, "merge_ms": statistics.mean(x["merge"] for x in ms),
                 "exchange_and_gather_ms": statistics.mean(x["exchange_and_gather"] for x in ms)}]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=_xdev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        box = [None] * world
        dist.all_gather_object(box, per_rank[0])
        per_rank = box
    projection = None
    if project and rank == 0 and world == 1:
        try:
            projection = shard_projection(args, ctx, capi, full, op, out, project, kind, merge_only / args.steps * 1e3, statistics.mean(ker_ms), (n_out, total_out))
        except Exception as e:
            projection = {"error": "%s: %s" % (type(e).__name__, e)}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.union8_no_cpu:
        try:
            cpu = multi_cpu_leg(args, ctx, capi, kind, full)
        except Exception as e:  # the GPU number stands on its own; say why the CPU leg is missing
            cpu = ({"value": None, "unit": "k-mers/s", "cores": 0, "kind": "reference", "sample": "failed: %s" % e}, False)
    res = None
    if rank == 0:
        width = ctx.get_counter("kway_width") if n_lists > 8 else 8
        if one_pass and n_lists <= width:
            path = "one pass of the N-way tile kernel" + (" (up to 32 lists per launch; a key lies in %.2f of the lists)" % (ctx.get_counter("kway_shared_x100") / 100.0) if n_lists > 8 else "")
        elif one_pass:
            path = "levels of eight-way passes of the N-way tile kernel (a key lies in %.2f of the lists: the one-pass form would rank every copy)" % (ctx.get_counter("kway_shared_x100") / 100.0)
        else:
            path = "left-to-right chain of pair intersections" if spec["op"] == "intersect" else "pairwise tree of the pair kernel"
        res = {
            "metric": "k-mers merged/sec, %s, lists resident in HBM, result gathered on rank 0" % (spec["what"] % args.k),
            "value": n_in * args.steps / elapsed, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64 keys + u32 counts", "data": "synthetic",
            "config": {"workload": "%s, %d lists of %d entries (%s keys), ONE job key-range sharded over %d GPU(s), RCCL gatherv to rank 0" % (spec["what"] % args.k, n_lists, n8, args.dist, world),
                       "reference": spec["ref"], "lists": n_lists, "entries_per_list": n8, "input_records": n_in, "dist": args.dist, "output_records": n_out, "output_total_count": total_out,
                       "device": ctx.device_info(), "per_rank": per_rank, "path": path, "splitters": args.splitters, "shard_first_keys": cuts,
                       "merge_only_k_mers_per_s": n_in * args.steps / merge_only,
                       "merge_only_ms_per_step": merge_only / args.steps * 1e3,
                       "gathered_bytes_per_step": 12 * (n_out - totals[0][0]) if world > 1 and gather_error is None else 0,
                       "gather_path": ("none (one GPU)" if world == 1 else ("gt4hip_comm_gatherv (RCCL: grouped ncclSend / ncclRecv)" if sh.gather_via == "rccl" else "torch.distributed send / recv (gatherv_records)%s" % (" through the host: test hook" if sh.host_tensors else ""))),
                       "gather_note": sh.gather_note or sh.comm_error, "gather_error": gather_error,
                       "note": "value includes the gatherv of the payload to rank 0 (when the gather failed: value = merge_only and gather_error says why); merge_only_* is the same job with every rank keeping (or writing) its own extent"},
            "roofline": multi_roofline(ctx, kind, n_local_in, totals[0][0], statistics.mean(dev_ms), statistics.mean(ker_ms), one_pass and n_lists <= width, n8, args.dist, moved, levels=one_pass and n_lists > width),
            **({"cpu_baseline": cpu[0], "verified": cpu[1]} if cpu else {}),
            **({"shard_projection": projection} if projection else {}),
        }
        bad = self_check(kind, args.dist, n8, args.k, n_out, total_out)
        res["self_check"] = "ok" if bad is None else "FAILED: " + bad
    sh.close()
    out.free()
    for l in shards + full:
        l.free()
    return res


def shard_projection(args, ctx, capi, full, op, out, N, kind, t1_ms, t1_ker, r1):
    """Single-GPU evidence for the N-GPU number (BASELINE: >= 6x at 8 GPUs on the 8-way union): the job's N key-range
    shards -- exactly the views rank g of an N-GPU run would merge (DeviceShards.plan / shard_of) -- run ONE AFTER
    ANOTHER on this GPU, each as its own timed call, next to the whole job as one call (t1_ms per call, t1_ker in the
    kernel, r1 = its (n_words, total_count)).  An N-GPU step takes the slowest shard plus the totals exchange (no
    payload moves in the merge_only form: every rank keeps or writes its extent), so
    projected_speedup = t(N = 1) / (max shard + exchange).  What this cannot see: the other ranks' skew at the barrier,
    xGMI, host jitter of eight processes -- the driver's SCALE run measures those."""
    from genometester4_amd import distributed as D
    n_in = sum(l.n_words for l in full)

    def timed(lists, steps, warmup):
        for _ in range(warmup):
            op(lists, out)
        ctx.synchronize()
        ms, ker = [], []
        r = None
        for _ in range(steps):
            t0 = time.perf_counter()
            r = op(lists, out)
            ctx.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
            ker.append(ctx.get_counter("nway_kernel_us") / 1000.0 if kind == "union8" else ctx.last_multi_device_ms if kind == "intersect8" else PAIR_KERNEL_MS[0])
        return statistics.mean(ms), statistics.mean(ker), r

    rows = []
    for splitters in (["sampled", "equal"] if (args.splitters == "sampled" and args.project_shards) else [args.splitters]):
        sh = D.DeviceShards(ctx, 0, N, None)
        cuts = sh.plan(full, sampled=splitters == "sampled")
        per = []
        n_sum = t_sum = 0
        for g in range(N):
            shards = [sh.shard_of(l, args.k, rank=g) for l in full]
            ms, ker, r = timed(shards, args.steps, max(1, min(args.warmup, 2)))
            n_sum += r[0]
            t_sum = (t_sum + r[1]) & 0xFFFFFFFFFFFFFFFF
            per.append({"shard": g, "first_key": cuts[g], "input_records": sum(x.n_words for x in shards), "output_records": r[0], "call_ms": ms, "kernel_ms": ker})
            for x in shards:
                x.free()
        rows.append({"splitters": splitters, "per_shard": per, "max_call_ms": max(p["call_ms"] for p in per), "sum_call_ms": sum(p["call_ms"] for p in per),
                     "input_imbalance": max(p["input_records"] for p in per) * N / max(1, n_in), "outputs_add_up": (n_sum, t_sum) == (r1[0], r1[1])})
    # the totals exchange: measured where a communicator can be made (one rank: the latency floor of the call itself)
    exch = args.exchange_ms
    exch_how = "given (--exchange-ms)"
    if exch is None:
        try:
            comm = ctx.comm_create(capi.comm_unique_id(), 1, 0)
            for _ in range(5):
                ctx.comm_allgather_totals(comm, 1, 1, 1)
            t0 = time.perf_counter()
            for _ in range(50):
                ctx.comm_allgather_totals(comm, 1, 1, 1)
            exch = (time.perf_counter() - t0) / 50 * 1e3
            capi.comm_destroy(comm)
            exch_how = "gt4hip_comm_allgather_totals on a communicator of ONE rank (H2D + ncclAllGather + D2H + synchronise): the call's floor, not eight ranks' skew"
        except Exception as e:  # noqa
            exch, exch_how = 0.1, "assumed (no RCCL communicator here: %s)" % e
    best = rows[0]
    return {"shards": N, "projected_speedup": t1_ms / (best["max_call_ms"] + exch), "shard_efficiency": t1_ms / best["sum_call_ms"],
            "whole_job_call_ms": t1_ms, "whole_job_kernel_ms": t1_ker, "whole_job_output_records": r1[0], "exchange_ms": exch, "exchange_measured": exch_how,
            "projection": [{"splitters": r["splitters"], "projected_speedup": t1_ms / (r["max_call_ms"] + exch), "shard_efficiency": t1_ms / r["sum_call_ms"],
                            "max_call_ms": r["max_call_ms"], "sum_call_ms": r["sum_call_ms"], "input_imbalance": r["input_imbalance"],
                            "outputs_add_up": r["outputs_add_up"], "per_shard": r["per_shard"]} for r in rows],
            "outputs_add_up": all(r["outputs_add_up"] for r in rows),
            "note": "projected_speedup = whole job / (slowest shard + totals exchange); shard_efficiency = whole job / sum of the shards (1.0: no per-call fixed cost); merge_only form (no payload gather); a projection from ONE GPU, not a scaling measurement"}


PAIR_KERNEL_MS = [0.0]


def project_shards(args, ctx, capi):
    """`--project-shards N` as a line of its own (see shard_projection)."""
    from genometester4_amd import distributed as D
    from genometester4_amd import synth
    N = args.project_shards
    kind = args.workload if args.workload in ("union8", "intersect8") else "intersect"
    if kind == "union8":
        full = synth.make_lists8(ctx, args.n8, args.k, args.dist, 8)
        op = D.gpu_union_multi_op(ctx)
        what = "8-way k=%d union, 8 lists of %d entries (%s keys)" % (args.k, args.n8, args.dist)
    elif kind == "intersect8":
        full = synth.make_lists_shared(ctx, args.n8, args.k, args.dist, 8)
        op = D.gpu_intersect_multi_op(ctx)
        what = "8-way k=%d intersection, 8 lists of %d entries (%s keys)" % (args.k, args.n8, args.dist)
    else:
        a, b = build_lists(ctx, capi, args.n, args.k, 0, args.dist)
        full = [a, b]
        what = "2-list k=%d intersection, 2 lists of %d entries (%s keys)" % (args.k, args.n, args.dist)

        def op(shards, out=None):
            st, outs, timing = ctx.compare(shards[0], shards[1], 2, out={2: out})
            PAIR_KERNEL_MS[0] = timing["merge_kernel_ms"]
            return st[2][0], st[2][1], outs[2]
    n_in = sum(l.n_words for l in full)
    union = kind == "union8"
    out = ctx.alloc(max(1, n_in if union else min(l.n_words for l in full)), args.k)
    for _ in range(args.warmup):
        op(full, out)
    ctx.synchronize()
    ms, ker = [], []
    r1 = None
    for _ in range(args.steps):
        t0 = time.perf_counter()
        r1 = op(full, out)
        ctx.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
        ker.append(ctx.get_counter("nway_kernel_us") / 1000.0 if kind == "union8" else ctx.last_multi_device_ms if kind == "intersect8" else PAIR_KERNEL_MS[0])
    t1_ms, t1_ker = statistics.mean(ms), statistics.mean(ker)
    proj = shard_projection(args, ctx, capi, full, op, out, N, kind, t1_ms, t1_ker, r1)
    res = {
        "metric": "projected %d-GPU speed-up of ONE job from its key-range shards timed one after another on one GPU" % N,
        "value": proj["projected_speedup"], "unit": "x", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": t1_ms, "higher_is_better": True, "scaling": "strong (projected)", "vs_baseline": None,
        "dtype": "u64 keys + u32 counts", "data": "synthetic",
        "config": dict({"workload": what + ", %d key-range shards" % N, "device": ctx.device_info(), "input_records": n_in}, **proj),
        "self_check": "ok" if proj["outputs_add_up"] else "FAILED: the shards' outputs do not add up to the whole job's",
    }
    out.free()
    for l in full:
        l.free()
    return res


def bench_sort(args, ctx, capi):
    """SURVEY 8f N2 (glistmaker's table step): --ns random k-mer words in HBM -> sorted (word, occurrences)
    list (wordtable_sort + wordtable_find_frequencies, reference src/word-table.c:217-260 on top of
    src/utils.c:127-198), by gt4hip_device_words_to_list: LSD radix sort, 8- and 9-bit digits (one histogram kernel for all passes,
    one chained-scan scatter kernel per pass), then the fold."""
    import numpy as np
    import torch
    n, k = args.ns, args.k
    bits = 64 if k >= 32 else 2 * k
    passes = (bits + 8) // 9  # 8- and 9-bit digits: as gt4hip_sort.hip plans them (k = 25: 9 + 9 + 8 + 8 + 8 + 8)
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    # repeats as a genome has them: a quarter of the words drawn from a small pool, the rest uniform
    pristine = torch.randint(0, (1 << min(bits, 62)), (n,), dtype=torch.int64, device="cuda", generator=g)
    pool = torch.randint(0, (1 << min(bits, 62)), (max(1, n // 64),), dtype=torch.int64, device="cuda", generator=g)
    idx = torch.randint(0, pool.numel(), (n // 4,), dtype=torch.int64, device="cuda", generator=g)
    pristine[: n // 4] = pool[idx]
    del idx
    work = torch.empty_like(pristine)
    sort_ms, fold_ms, wall = [], [], []
    n_out = total = 0
    for it in range(args.warmup + args.steps):
        work.copy_(pristine)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lst = ctx.device_words_to_list(work.data_ptr(), n, k)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        if it >= args.warmup:
            wall.append(dt)
            sort_ms.append(ctx.get_counter("sort_us") / 1000.0)
            fold_ms.append(ctx.get_counter("fold_us") / 1000.0)
        n_out, total = lst.n_words, lst.sum_counts()
        assert total == n
        if it == 0:
            assert lst.is_sorted()
        lst.free()
    s_ms = statistics.mean(sort_ms)
    sort_traffic, sort_traffic_source = load_traffic("sort", n)
    alg = 16 * passes * n  # SURVEY / VERDICT: 16 bytes moved per word and pass (read + write)
    res = {"metric": "k-mer words sorted and folded/sec (glistmaker table step), k=%d, words resident in HBM" % k,
           "value": n / statistics.mean(wall), "unit": "words/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": statistics.mean(wall) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u64 words", "data": "synthetic",
           "config": {"workload": "sort + fold of %d random k=%d words (a quarter drawn from a pool of %d) -> %d-record list" % (n, k, max(1, n // 64), n_out),
                      "words": n, "word_length": k, "radix_passes": passes, "output_records": n_out, "device": ctx.device_info(),
                      "wall_ms_per_step": [round(w * 1e3, 2) for w in wall]},
           "roofline": {"bound": "hbm", "kernel": "k_radix_hist + k_radix_scatter x %d passes" % passes, "achieved": alg / (s_ms * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": sort_traffic, "traffic_source": sort_traffic_source, "traffic_note": "per launch of k_radix_scatter = one pass (16 B x words algorithmic)",
                        "algorithmic_bytes_per_launch": alg, "kernel_ms_avg": s_ms, "fold_ms_avg": statistics.mean(fold_ms),
                        "note": "achieved = 16 B x passes x words / time of the sort (HIP events: histogram kernel + passes); the histogram kernel reads the words once more (8 B per word, once)"}}
    if not args.no_cpu_baseline:
        m = min(n, args.cpu_sample // 4)
        host = pristine[:m].cpu().numpy().astype(np.uint64)
        t0 = time.perf_counter()
        u, c = np.unique(host, return_counts=True)
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": m / dt, "unit": "words/s", "cores": 1, "host_nproc": os.cpu_count(), "kind": "port",
                               "sample": "numpy.unique(return_counts=True) of the first %d words (sort + run lengths, one thread): what wordtable_sort + wordtable_find_frequencies compute" % m}
        dev = ctx.words_to_list(host, k).download()
        res["verified"] = bool(len(dev) == len(u) and (dev["key"] == u).all() and (dev["count"] == c.astype(np.uint32)).all())
    return res


def bench_table(args, ctx, capi):
    """SURVEY 8f N3 (glistquery's multi-list dump): per-key count table of --nt-lists lists of --nt entries
    (gt4_union's callback rows, reference src/set-operations.c:131-183) by gt4hip_union_table: ONE launch of the
    N-way tile kernel (every tile writes its rows where its records start: a ragged table, gathered at download);
    9 .. 32 lists: the 32-list instance of the kernel, one launch as well (round 5; --kway-max 8 restores round 4's path:
    the N-way union for the keys and one streaming merge per column, which more than 32 lists still take)."""
    nl, n, k = args.nt_lists, args.nt, args.k
    if args.kway_max:
        ctx.set_option("kway_max", args.kway_max)
    one_launch = nl <= (8 if args.kway_max == 8 else 32)
    lists = []
    for j in range(nl):
        lst = ctx.alloc(n, k)
        shared = j % 2 == 0
        ctx.generate_ex(lst, n, 7 if shared else 100 + j, 50 + j, 8, 16 if nl < 16 else 64, 0 if shared else 1 + j)  # (residue 1 + j < modulus)
        lists.append(lst)
    wall, tab = [], []
    n_keys = 0
    for it in range(args.warmup + args.steps):
        ctx.synchronize()
        t0 = time.perf_counter()
        n_keys = ctx.union_table_device(lists)
        ctx.synchronize()
        if it >= args.warmup:
            wall.append(time.perf_counter() - t0)
            tab.append(ctx.get_counter("table_us") / 1000.0)
    t_ms = statistics.mean(tab)
    table_traffic, table_traffic_source = load_traffic("table" if nl == 6 else "table%d" % nl, n)
    alg = 12 * nl * n + (8 + 4 * nl) * n_keys
    res = {"metric": "k-mers tabulated/sec (glistquery multi-list dump: per-key counts of %d lists), lists resident in HBM" % nl,
           "value": nl * n / statistics.mean(wall), "unit": "k-mers/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": statistics.mean(wall) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u64 keys + u32 counts", "data": "synthetic",
           "config": {"workload": "count table of %d lists x %d k=%d entries -> %d keys x %d counts" % (nl, n, k, n_keys, nl),
                      "lists": nl, "entries_per_list": n, "keys": n_keys, "device": ctx.device_info(),
                      "wall_ms_per_step": [round(w * 1e3, 2) for w in wall]},
           "roofline": {"bound": "hbm", "kernel": ("%s::k_nway_merge<1024, 4, 1, NWAY_TABLE> (one launch)" % ("km8" if nl <= 8 else "km32")) if one_launch else "k_nway_merge (the keys) + %d x (k_pair_merge union + k_extract_column)" % nl,
                        "achieved": alg / (t_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": table_traffic, "traffic_source": table_traffic_source, "traffic_note": "the table launch of the tile kernel (the call's only pass over the records)",
                        "algorithmic_bytes_per_launch": alg, "kernel_ms_avg": t_ms,
                        "note": "algorithmic = every input record read once + the table's rows written once; achieved is over the whole call (sampling, partition, the tile kernel's launch, the ragged table's index)"}}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        m = min(n, args.cpu_sample // (16 * nl))  # (the reference formats every row as text: ~1e7 rows keep the leg at ~10 s)
        last_key, _ = lists[0].get_word(m - 1)
        host = [l.download_range(0, l.lower_bound(last_key + 1)) for l in lists]
        ref = os.path.join(ROOT, "oracle", "_ref", "glistquery")
        done = False
        if os.path.exists(ref):
            # the REFERENCE's own multi-list dump (glistquery L1 .. LN -> dump_lists -> gt4_union, src/glistquery.c:82-106,
            # src/set-operations.c:131-183) on the sample, its text thrown away; one row per distinct key
            import shutil
            import subprocess
            import tempfile
            from genometester4_amd.listio import write_list
            d = tempfile.mkdtemp(prefix="gt4tab_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
            try:
                names = []
                for j, h in enumerate(host):
                    write_list(os.path.join(d, "L%d.list" % j), h, k)
                    names.append("L%d.list" % j)
                t0 = time.perf_counter()
                p = subprocess.run("%s %s | wc -l" % (ref, " ".join(names)), shell=True, cwd=d, capture_output=True, text=True)
                dt = time.perf_counter() - t0
                rows = int(p.stdout.strip() or 0)
                rc_g, n_g, _, _ = ctx.union_multi([ctx.upload(h, k) for h in host], 0, 4, 1, True)
                # the product's own dump (examples/setops_driver.c dump = gt4_union + the same print loop) on the same files:
                # the SAME text, the reference's all-zero visits behind an exhausted list's last key included
                drv = os.path.join(ROOT, "genometester4_amd", "setops_driver")
                same_text = None
                if os.path.exists(drv):
                    q = subprocess.run("%s dump %s | md5sum; %s %s | md5sum" % (drv, " ".join(names), ref, " ".join(names)), shell=True, cwd=d, capture_output=True, text=True)
                    sums = [l.split()[0] for l in q.stdout.strip().splitlines() if l.strip()]
                    same_text = len(sums) == 2 and sums[0] == sums[1]
                res["cpu_baseline"] = {"value": sum(len(h) for h in host) / dt, "unit": "k-mers/s", "cores": 1, "host_nproc": os.cpu_count(), "kind": "reference",
                                       "sample": "oracle/_ref/glistquery L1 .. L%d | wc -l (dump_lists -> gt4_union; the reference formats every row as text) over the first %d records of every list, files in %s, one thread" % (nl, m, "tmpfs" if d.startswith("/dev/shm") else "the temp dir")}
                res["verified"] = bool(p.returncode == 0 and rc_g == 0 and 0 <= rows - n_g < nl and same_text is not False)
                res["verified_rows"] = {"reference_lines": rows, "gpu_distinct_keys": n_g, "gpu_dump_text_equals_reference": same_text,
                                        "note": "the reference visits an exhausted list's last key once more with all-zero counts (src/set-operations.c:166-170): up to lists - 1 lines more than distinct keys"}
                done = p.returncode == 0
            finally:
                shutil.rmtree(d, ignore_errors=True)
        if not done:
            t0 = time.perf_counter()
            rc, n_u, _, _ = O.union_multi(host, 0, 4, 1)
            dt = time.perf_counter() - t0
            res["cpu_baseline"] = {"value": sum(len(h) for h in host) / dt, "unit": "k-mers/s", "cores": 1, "host_nproc": os.cpu_count(), "kind": "port",
                                   "sample": "oracle/gt4_oracle.c union_multi walk (the loop gt4_union shares, set-operations.c:153-181) over the first %d records of every list, one thread" % m}
    return res


def bench_pair(args, ctx, capi, rank, world, torch, dist, extras=False):
    """The two-list workloads (intersect: BASELINE configs[1]; c2: configs[2]).  Returns the result line on rank 0.
    `extras` (the default line at N = 1): the same resident pair also runs configs[2] (record "c2") and feeds the
    file -> file run of the C command-line tool against the reference binary (record "e2e")."""
    n = args.n
    strong = args.scaling == "strong"
    sharded = strong and world > 1
    all_bits = (1, 2, 4)
    while True:
        a = b = full_a = full_b = sh = None
        ok = True
        try:
            # strong scaling: every rank builds the SAME pair and keeps its key range of it
            a, b = build_lists(ctx, capi, n, args.k, 0 if strong else 1000 * rank, args.dist)
            if sharded:
                from genometester4_amd import distributed as D
                # (a communicator where the ranks are on devices of their own: the step's totals exchange is then ONE
                # ncclAllGather on the library's stream, gt4hip_comm_allgather_u64; whether every rank has it is agreed
                # on once, inside DeviceShards)
                comm_id = None
                if _xdev() == "cuda":
                    box = [capi.comm_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    comm_id = box[0]
                sh = D.DeviceShards(ctx, rank, world, comm_id, agree=_agree)
                full_a, full_b = a, b
                sh.plan([full_a, full_b], sampled=args.splitters == "sampled")
                a, b = sh.shard_of(full_a, args.k), sh.shard_of(full_b, args.k)
            # the worst-case outputs of the workload must fit too (probe: allocate and free)
            probe_bits = [2] if args.workload == "intersect" else [1, 4]
            probe = [ctx.alloc(max(1, {1: a.n_words + b.n_words, 2: min(a.n_words, b.n_words), 4: a.n_words}[bit]), args.k) for bit in probe_bits]
            for l in probe:
                l.free()
        except capi.Gt4HipError as e:
            if e.code != capi.ENOMEM or n < 1_000_000:
                raise
            log("rank %d: %d entries per list do not fit (%s)" % (rank, n, e))
            ok = False
        # every rank must run the same shape (strong: the same JOB): agree on the smallest n that fits anywhere
        n_next = n if ok else n // 2
        if world > 1:
            t = torch.tensor([n_next], dtype=torch.int64, device=_xdev())
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            n_next = int(t.item())
        if ok and n_next == n:
            break
        if sharded and sh is not None:
            sh.close()
        for l in (a, b, full_a, full_b):
            if l is not None:
                l.free()
        n = n_next
    n_a, n_b = a.n_words, b.n_words
    n_job = (full_a.n_words + full_b.n_words) if sharded else (n_a + n_b)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def leg(workload, cpu_sample):
        """one workload on the resident pair: warm-up, the timed steps, the record (rank 0), the CPU leg (N = 1)"""
        ops = capi.OP_INTRSEC if workload == "intersect" else (capi.OP_UNION | capi.OP_DIFF1)
        cutoff = 1 if workload == "intersect" else 3
        op_bits = [bit for bit in all_bits if ops & bit]
        outs = {bit: ctx.alloc(max(1, {1: a.n_words + b.n_words, 2: min(a.n_words, b.n_words), 4: a.n_words}[bit]), args.k) for bit in op_bits}
        exchange_ms = []
        job_stat = {}  # strong scaling: the job-wide header totals of the last step

        def step():
            st, _, timing = ctx.compare(a, b, ops, cutoff=cutoff, out=outs)
            if sharded:
                # the one exchange a sharded pair operation needs: per-shard header totals,
                # all-gathered inside the step -- over RCCL on the library's stream where the ranks agreed on it
                from genometester4_amd import distributed as D
                t0 = time.perf_counter()
                if sh.comm is not None:
                    words = [w for bit in op_bits for w in st[bit]]
                    rows = ctx.comm_allgather_u64(sh.comm, world, words)
                    job = {bit: (sum(r[2 * i] for r in rows), sum(r[2 * i + 1] for r in rows)) for i, bit in enumerate(op_bits)}
                else:
                    job = {bit: tuple(sum(x[i] for x in D.exchange_totals(st[bit][0], st[bit][1], device=_xdev())) for i in (0, 1)) for bit in op_bits}
                exchange_ms.append((time.perf_counter() - t0) * 1e3)
                job_stat.update(job)
            return st, timing

        try:
            for _ in range(args.warmup):
                step()
            fence()
            t0 = time.perf_counter()
            kernel_ms, device_ms = [], []
            stat = None
            for _ in range(args.steps):
                stat, timing = step()
                kernel_ms.append(timing["merge_kernel_ms"])
                device_ms.append(timing["device_ms"])
            fence()
            elapsed = time.perf_counter() - t0
            totals = [sum(stat[bit][0] for bit in op_bits), sum(stat[bit][1] for bit in op_bits) & 0x7FFFFFFFFFFFFFFF]
            per_rank = None
            if world > 1:
                t = torch.tensor([elapsed], dtype=torch.float64, device=_xdev())
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
                box = [None] * world
                dist.all_gather_object(box, {"rank": rank, "shard_input_records": n_a + n_b, "merge_kernel_ms": statistics.mean(kernel_ms),
                                             "totals_exchange_ms": statistics.mean(exchange_ms) if exchange_ms else 0.0})
                per_rank = box
            if sharded:
                totals = [sum(job_stat[bit][0] for bit in op_bits), sum(job_stat[bit][1] for bit in op_bits) & 0x7FFFFFFFFFFFFFFF]  # exchanged inside the step
            elif world > 1:
                # independent shards: the header totals are summed once, for the report
                g = [torch.zeros(2, dtype=torch.int64, device=_xdev()) for _ in range(world)]
                dist.all_gather(g, torch.tensor(totals, dtype=torch.int64, device=_xdev()))
                totals = [int(sum(x[0].item() for x in g)), int(sum(x[1].item() for x in g))]
            res = None
            if rank == 0:
                n_out = sum(stat[bit][0] for bit in op_bits)
                k_ms = statistics.mean(kernel_ms)
                alg_bytes = 12 * (n_a + n_b) + 12 * n_out
                achieved = alg_bytes / (k_ms * 1e-3) / 1e9
                traffic, traffic_source = load_traffic(workload if args.dist == "stride" else "%s_%s" % (workload, args.dist), n)
                names = {1: "union", 2: "intrsec", 4: "diff1"}
                shape = "two %d-entry k=%d lists (%.1f GB each, %s keys)" % (n, args.k, 12 * n / 1e9, args.dist)
                if workload == "intersect":
                    metric = "k-mers merged/sec, 2-list k=%d intersection (glistcompare -i), lists resident in HBM" % args.k
                    if sharded:
                        wl = "ONE intersection of %s, key-range sharded over %d GPUs (gt4hip_shard_cuts), header totals all-gathered in every step, |A n B| ~ n/2" % (shape, world)
                    else:
                        wl = ("single-GPU" if world == 1 else "%d independent pairs, one per GPU:" % world) + " intersection, %s per GPU, |A n B| ~ n/2" % shape
                    kernel = "k_pair_merge<1024, 6, MODE_LOOKBACK, intersection, folded MIN>"
                else:
                    metric = "k-mers merged/sec, 2-list k=%d union + first complement, cutoff %d (glistcompare -u -d -c %d), lists resident in HBM" % (args.k, cutoff, cutoff)
                    wl = ("single-GPU" if world == 1 else ("ONE job key-range sharded over %d GPUs:" % world if strong else "%d independent pairs, one per GPU:" % world)) + " union + difference_first with --cutoff %d, %s per GPU, |A n B| ~ n/2" % (cutoff, shape)
                    kernel = "k_pair_merge<1024, 4, MODE_LOOKBACK, any combination of outputs, default rules>"
                res = {
                    "metric": metric,
                    "value": (n_job if (strong or world == 1) else world * (n_a + n_b)) * args.steps / elapsed,
                    "unit": "k-mers/s",
                    "n_gpus": world,
                    "steps": args.steps,
                    "warmup": args.warmup,
                    "ms_per_step": elapsed / args.steps * 1e3,
                    "higher_is_better": True,
                    "scaling": "strong" if strong else "weak",
                    "vs_baseline": None,
                    "dtype": "u64 keys + u32 counts",
                    "data": "synthetic",
                    "config": {
                        "workload": wl,
                        "dist": args.dist,
                        "per_rank": per_rank,
                        "entries_per_list_per_gpu": n if not sharded else None,
                        "entries_per_list": n if strong else None,
                        "input_records": n_job if (strong or world == 1) else world * (n_a + n_b),
                        "word_length": args.k,
                        "output_records": {names[bit]: stat[bit][0] for bit in op_bits} if world == 1 else totals[0],
                        "output_total_count": {names[bit]: stat[bit][1] for bit in op_bits} if world == 1 else totals[1],
                        "sharding": ("one job, key ranges holding equal numbers of input records (sampled splitters), totals all-gather per step" if strong else "independent pairs, no data-path collective") if world > 1 else "none",
                        "totals_exchange": (("RCCL all-gather on the library's stream (gt4hip_comm_allgather_u64)" if sh.comm is not None else "torch.distributed all_gather%s" % (" (%s)" % sh.comm_error if sh.comm_error else "")) if sharded else "none"),
                        "path": "two_pass" if args.two_pass else "single_pass_lookback",
                        "device": ctx.device_info(),
                    },
                    "roofline": {
                        "bound": "hbm",
                        "kernel": kernel,
                        "achieved": achieved,
                        "peak": HBM_PEAK_GBS,
                        "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS,
                        "read_only_frac": 12 * (n_a + n_b) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": alg_bytes,
                        "kernel_ms_avg": k_ms,
                        "device_ms_avg": statistics.mean(device_ms),
                        "traffic": traffic,
                        "traffic_source": traffic_source,
                        "note": "rank 0's launch" + (": its key range of the job" if sharded else ""),
                    },
                }
                if workload == "intersect" and (strong or world == 1):
                    bad = self_check("intersect", args.dist, n, args.k, totals[0], totals[1])
                    res["self_check"] = "ok" if bad is None else "FAILED: " + bad
                if world == 1 and not args.no_cpu_baseline:
                    try:
                        res["cpu_baseline"], res["verified"], res["verified_totals"] = cpu_baseline(ctx, capi, a, b, args.k, cpu_sample, ops, cutoff)
                    except Exception as e:  # the GPU number stands on its own; say why the CPU leg is missing
                        res["cpu_baseline"] = {"value": None, "unit": "k-mers/s", "cores": 0, "kind": "reference", "sample": "failed: %s" % e}
                        res["verified"] = False
                    if res["verified"] is False and res["cpu_baseline"].get("value") is not None:
                        log("VERIFICATION FAILED: %s" % res.get("verified_totals"))
            return res
        finally:
            for l in outs.values():
                l.free()

    try:
        res = leg(args.workload, args.cpu_sample)
        PROGRESS["headline"] = res
        if extras and rank == 0 and world == 1 and res is not None:
            GUARD.arm("c2", args.leg_timeout)
            try:
                c2 = leg("c2", max(1_000_000, args.cpu_sample // 4))
                res["c2"] = {"workload": c2["config"]["workload"], "metric": c2["metric"], "value": c2["value"], "unit": c2["unit"], "steps": c2["steps"],
                             "ms_per_step": c2["ms_per_step"], "output_records": c2["config"]["output_records"],
                             "roofline": {k: c2["roofline"][k] for k in ("kernel", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch", "kernel_ms_avg", "traffic", "traffic_source")},
                             "verified": c2.get("verified"), "verified_totals": c2.get("verified_totals"), "cpu_baseline": c2.get("cpu_baseline"),
                             "note": "BASELINE configs[2] on the SAME resident pair: glistcompare -u -d -c 3 (two outputs, count cutoff: output compaction); verified = the GPU's totals on the CPU sample equal the reference binary's"}
            except Exception as e:
                res["c2"] = {"error": "%s: %s" % (type(e).__name__, e)}
            PROGRESS["headline"] = res
            if not args.no_cpu_baseline:
                GUARD.arm("e2e", args.leg_timeout)
                try:
                    res["e2e"] = e2e_leg(args, ctx, a, b)
                except Exception as e:
                    res["e2e"] = {"error": "%s: %s" % (type(e).__name__, e)}
                PROGRESS["headline"] = res
    finally:
        if sharded and sh is not None:
            sh.close()
        for l in [a, b, full_a, full_b]:
            if l is not None:
                l.free()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--entries", dest="n", type=int, default=2_000_000_000,
                    help="entries per list (--entries: the spelling to use under torch.distributed.run, whose own parser takes --n for an abbreviation)")
    ap.add_argument("--k", type=int, default=25)
    ap.add_argument("--cpu-sample", type=int, default=200_000_000, help="records per list timed on the CPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--two-pass", action="store_true", help="count+scan+write instead of the single-pass kernel")
    ap.add_argument("--ns", type=int, default=1_000_000_000, help="sort: words")
    ap.add_argument("--nt", type=int, default=100_000_000, help="table: entries per list")
    ap.add_argument("--nt-lists", type=int, default=6, help="table: lists")
    ap.add_argument("--kway-max", type=int, default=0, choices=[0, 8, 32, 33], help="table: library option kway_max (8: lists beyond eight by merges, as in round 4)")
    ap.add_argument("--workload", choices=["intersect", "c2", "union8", "union32", "intersect8", "sort", "table"], default="intersect",
                    help="intersect: BASELINE configs[1] (default, the headline metric; the line also embeds a union8 record); c2: configs[2], "
                         "union + first complement with cutoff 3 on the same pair; union8: configs[3] alone, 8-way union sharded by key range "
                         "over the ranks with an RCCL gatherv to rank 0 (strong scaling); union32: 32 lists (glistmaker's collation width); "
                         "intersect8: eight lists, MakeIntersection.pl's job")
    ap.add_argument("--n32", "--entries32", dest="n32", type=int, default=125_000_000, help="union32: entries per list (whole job)")
    ap.add_argument("--n8", "--entries8", dest="n8", type=int, default=500_000_000, help="union8: entries per list (whole job)")
    ap.add_argument("--tree", action="store_true", help="union8: the pairwise tree instead of what the library chooses")
    ap.add_argument("--dist", choices=["stride", "iid", "clustered", "genomic", "disjoint"], default="stride", help="key distribution of the synthetic lists (genometester4_amd/synth.py)")
    ap.add_argument("--splitters", choices=["sampled", "equal"], default="sampled",
                    help="how a sharded job cuts the key space: sampled from the lists (gt4hip_shard_cuts: equal input records per shard) or equal-width ranges (gt4hip_shard_first_key)")
    ap.add_argument("--project-shards", type=int, default=0, metavar="N",
                    help="ONE GPU: run each of the N key-range shards of the job (union8 or intersect) as its own timed call and project the N-GPU speed-up")
    ap.add_argument("--exchange-ms", type=float, default=None, help="--project-shards: totals-exchange latency per step to add (default: measured with a one-rank RCCL all-gather)")
    ap.add_argument("--no-union8", action="store_true", help="intersect: leave the union8 record out of the line")
    ap.add_argument("--no-extras", action="store_true", help="intersect at N = 1: leave the c2, e2e and shard_projection records out of the line")
    ap.add_argument("--e2e-n", type=int, default=200_000_000, help="e2e record: records of list A written to the sample files (list B: the same key range)")
    ap.add_argument("--leg-timeout", type=float, default=420.0, help="wall-clock bound per leg (pair, c2, e2e, union8): beyond it the line is printed as far as it got and the run exits 4")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="intersect / c2 with --gpus N > 1: strong (default) = ONE pair of --n entries, every rank merges its key range "
                         "(gt4hip_shard_first_key) and the header totals are all-gathered inside every step; weak = one independent pair of "
                         "--n entries per GPU, no exchange")
    args = ap.parse_args()
    args.union8_no_cpu = False

    own_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if os.environ.get("GT4_BENCH_ONE_DEVICE"):
        # test hook: run the N-rank control flow with every rank on device 0 (RCCL refuses two ranks
        # on one device, so the barrier / reductions go over gloo); never set by the driver
        local_rank = 0
        torch.cuda.set_device(0)
        if world > 1:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local_rank)
        if world > 1:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    global GUARD
    GUARD = Guard(rank)
    from genometester4_amd import capi
    ctx = capi.Context(local_rank)
    if args.two_pass:
        ctx.set_option("two_pass", 1)
    res = None
    default_line = args.workload == "intersect" and not args.project_shards
    extras = default_line and world == 1 and not args.no_extras
    GUARD.arm(args.workload, args.leg_timeout)
    if args.project_shards:
        res = project_shards(args, ctx, capi)
    elif args.workload in MULTI:
        res = bench_multi(args, ctx, capi, rank, local_rank, world, args.workload)
    elif args.workload == "sort":
        res = bench_sort(args, ctx, capi)
    elif args.workload == "table":
        res = bench_table(args, ctx, capi)
    else:
        res = bench_pair(args, ctx, capi, rank, world, torch, dist, extras=extras)
        PROGRESS["headline"] = res
        if default_line and not args.no_union8:
            # the other north-star number in the same line: the 8-way union as ONE job over the same ranks.  Whatever
            # happens in here -- an exception on every rank, a gather that fails at first contact -- the intersection
            # line above is printed, with what the union had measured by then.
            args.union8_no_cpu = True  # (its CPU leg belongs to --workload union8)
            GUARD.arm("union8", args.leg_timeout)
            u, err = None, None
            try:
                u = bench_multi(args, ctx, capi, rank, local_rank, world, "union8", progress=PROGRESS["union8"], project=8 if extras else 0)
            except Exception as e:
                err = "%s: %s" % (type(e).__name__, e)
                log("rank %d: union8 leg failed: %s" % (rank, err))
            if rank == 0:
                if u is not None:
                    res["union8"] = {
                        "workload": u["config"]["workload"], "n_gpus": world, "scaling": "strong",
                        "value_with_gather": u["value"], "ms_per_step_with_gather": u["ms_per_step"],
                        "merge_only": u["config"]["merge_only_k_mers_per_s"], "merge_only_ms_per_step": u["config"]["merge_only_ms_per_step"],
                        "unit": "k-mers/s", "per_rank": u["config"]["per_rank"], "gathered_bytes": u["config"]["gathered_bytes_per_step"],
                        "gather_path": u["config"]["gather_path"], "gather_note": u["config"]["gather_note"],
                        "output_records": u["config"]["output_records"], "output_total_count": u["config"]["output_total_count"],
                        "path": u["config"]["path"], "roofline_frac": u["roofline"]["frac"], "whole_call_frac": u["roofline"]["whole_call_frac"],
                        "kernel_ms_avg": u["roofline"]["kernel_ms_avg"], "device_ms_avg": u["roofline"]["device_ms_avg"],
                        "self_check": u["self_check"],
                        "note": "north_star asks for >= 6x at 8 GPUs on the 8-way union AND for a final gatherv: merge_only (every rank keeps / writes its own "
                                "extent) is what can scale; value_with_gather moves 7/8 of the result into rank 0 over xGMI inside the step and is bound by "
                                "the root's inbound links (DESIGN.md section 6 states the conflict); both are reported"}
                    if u["config"]["gather_error"]:
                        res["union8"]["error"] = u["config"]["gather_error"]
                    if "shard_projection" in u:
                        res["shard_projection"] = u["shard_projection"]
                else:
                    res["union8"] = dict(PROGRESS["union8"], error=err)
    GUARD.disarm()
    failed = False
    if rank == 0 and res is not None:
        emit(res)
        checks = [res.get("self_check"), (res.get("union8") or {}).get("self_check")]
        failed = any(c and c != "ok" for c in checks)
        if failed:
            log("SELF-CHECK FAILED: %s" % checks)
    if world > 1:
        GUARD.arm("shutdown", 60.0)
        dist.barrier()
        dist.destroy_process_group()
        GUARD.disarm()
    ctx.close()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
