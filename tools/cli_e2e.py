"""End-to-end wall time of the drop-in CLI against the reference binary on files in /dev/shm
(GPU box; not part of the product).

    python tools/cli_e2e.py [records_per_list] [--no-ref] [--modes=plain,chunks,gpus2,plain]

Modes of the drop-in: plain (no environment variables: inputs of 4 GiB and more take the chunk pipeline with a
budget the tool chooses, smaller ones stay in one piece), chunks (GT4HIP_HBM_LIMIT: key-range chunks through the
loader / merger / writer pipeline with a quarter of the inputs in flight), gpus2 (two worker processes on the
visible device(s)), and the reference binary.  Prints one line per run and whether the outputs are identical."""
import os, subprocess, sys, time, shutil, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genometester4_amd import capi
from genometester4_amd.listio import write_list
from bench import build_lists

n = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 100_000_000
with_ref = "--no-ref" not in sys.argv
d = tempfile.mkdtemp(prefix="gt4cli_", dir="/dev/shm")
results = []
try:
    ctx = capi.Context(0)
    a, b = build_lists(ctx, capi, n, 25, 0)
    for name, lst in (("a", a), ("b", b)):
        fd = os.open(os.path.join(d, name + ".list"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        from genometester4_amd.listio import header_bytes
        os.write(fd, header_bytes(25, lst.n_words, lst.sum_counts()))
        ctx.write_fd(lst, 0, lst.n_words, fd, 48)
        os.close(fd)
    ctx.close()
    print("inputs: 2 x %d records (%.1f GB each) in %s, free %.1f GB" % (n, 12 * n / 1e9, d, shutil.disk_usage("/dev/shm").free / 1e9), flush=True)
    ours = os.path.join(ROOT, "genometester4_amd", "glistcompare")
    ref = os.path.join(ROOT, "oracle", "_ref", "glistcompare")
    limit = str(max(64 << 20, 12 * n // 2))  # about a quarter of the inputs in flight
    all_modes = {"plain": (ours, {}), "chunks": (ours, {"GT4HIP_HBM_LIMIT": limit}), "gpus2": (ours, {"GT4HIP_GPUS": "2", "GT4HIP_HBM_LIMIT": limit}),
                 "oneshot": (ours, {"GT4HIP_PIPELINE": "0"}), "ref": (ref, {})}
    all_modes["plain2"] = all_modes["plain"]  # the same again (first-run effects)
    verbose = {"GT4HIP_VERBOSE": "1"} if "--verbose" in sys.argv else {}
    order = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--modes=")]
    order = order[0].split(",") if order else ["plain", "chunks", "gpus2"] + (["ref"] if with_ref else [])
    modes = [(m, all_modes[m][0], all_modes[m][1]) for m in order]
    for tag, exe, env in modes:
        for args in (["-i", "--count_only"], ["-i", "-o", tag], ["-u", "-i", "-d", "-o", tag + "3"]):
            t0 = time.perf_counter()
            r = subprocess.run([exe, "a.list", "b.list"] + args, cwd=d, capture_output=True, env=dict(os.environ, **env, **verbose))
            dt = time.perf_counter() - t0
            print("%-7s %-24s rc %d  %.3f s  (%.1f M k-mers/s)" % (tag, " ".join(args[:-1] if "-o" in args else args), r.returncode, dt, 2 * n / dt / 1e6), flush=True)
            results.append(dict(mode=tag, args=args, rc=r.returncode, seconds=dt, k_mers_per_s=2 * n / dt))
            if r.returncode or verbose:
                print(r.stderr.decode()[-1500:])
        if tag != order[0] and os.path.exists(os.path.join(d, "%s_25_intrsec.list" % order[0])):
            same = all(subprocess.run(["cmp", "-s", os.path.join(d, x % tag), os.path.join(d, x % order[0])]).returncode == 0
                       for x in ("%s_25_intrsec.list", "%s3_25_union.list", "%s3_25_0_diff1.list", "%s3_25_intrsec.list"))
            print("%-7s outputs identical to %s: %s" % (tag, order[0], same), flush=True)
            results.append(dict(mode=tag, identical_to_plain=same))
            for x in ("%s_25_intrsec.list", "%s3_25_union.list", "%s3_25_0_diff1.list", "%s3_25_intrsec.list"):
                os.remove(os.path.join(d, x % tag))
    print(json.dumps(dict(records_per_list=n, results=results)))
finally:
    shutil.rmtree(d, ignore_errors=True)
