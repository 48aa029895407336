"""End-to-end wall time of the drop-in CLI against the reference binary on files in /dev/shm
(GPU box; not part of the product).  python tools/cli_e2e.py [records_per_list]"""
import os, subprocess, sys, time, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genometester4_amd import capi
from genometester4_amd.listio import write_list
from bench import build_lists

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
d = tempfile.mkdtemp(prefix="gt4cli_", dir="/dev/shm")
try:
    ctx = capi.Context(0)
    a, b = build_lists(ctx, capi, n, 25, 0)
    write_list(os.path.join(d, "a.list"), a.download(), 25)
    write_list(os.path.join(d, "b.list"), b.download(), 25)
    ctx.close()
    ours = os.path.join(ROOT, "genometester4_amd", "glistcompare")
    ref = os.path.join(ROOT, "oracle", "_ref", "glistcompare")
    for tag, exe in (("hip", ours), ("ref", ref)):
        for args in (["-i", "--count_only"], ["-i", "-o", tag], ["-u", "-i", "-d", "-o", tag + "3"]):
            t0 = time.perf_counter()
            r = subprocess.run([exe, "a.list", "b.list"] + args, cwd=d, capture_output=True)
            dt = time.perf_counter() - t0
            print("%s %-28s rc %d  %.3f s  (%.1f M k-mers/s)" % (tag, " ".join(args), r.returncode, dt, 2 * n / dt / 1e6), flush=True)
    same = all(open(os.path.join(d, x % "hip"), "rb").read() == open(os.path.join(d, x % "ref"), "rb").read()
               for x in ("%s_25_intrsec.list", "%s3_25_union.list", "%s3_25_0_diff1.list"))
    print("outputs identical:", same)
finally:
    shutil.rmtree(d, ignore_errors=True)
