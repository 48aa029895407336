"""Experiment driver (not product): wall time of `glistcompare -u -i -d` (three output files) on two lists in
/dev/shm for different settings of the result writers.  usage: exp_e2e_out.py [records per list]"""
import os, subprocess, sys, time, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genometester4_amd import capi
from genometester4_amd.listio import header_bytes
from bench import build_lists
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000_000
d = tempfile.mkdtemp(prefix="gt4e2e_", dir="/dev/shm")
try:
    ctx = capi.Context(0)
    a, b = build_lists(ctx, capi, n, 25, 0)
    for name, lst in (("a", a), ("b", b)):
        fd = os.open(os.path.join(d, name + ".list"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        os.write(fd, header_bytes(25, lst.n_words, lst.sum_counts()))
        ctx.write_fd(lst, 0, lst.n_words, fd, 48)
        os.close(fd)
    ctx.close()
    exe = os.path.join(ROOT, "genometester4_amd", "glistcompare")
    for env in ({}, {"GT4HIP_IO_THREADS": "8"}, {"GT4HIP_IO_THREADS": "16"}, {"GT4HIP_IO_THREADS": "24"}, {"GT4HIP_IO_THREADS": "12", "GT4HIP_IO_MMAP": "1"},
                {"GT4HIP_IO_THREADS": "12", "GT4HIP_IO_PIECE_MB": "32"}):
        for args in (["-u", "-i", "-d", "-o", "x"], ["-i", "-o", "y"]):
            t0 = time.perf_counter()
            r = subprocess.run([exe, "a.list", "b.list"] + args, cwd=d, capture_output=True, env=dict(os.environ, **env))
            dt = time.perf_counter() - t0
            out_bytes = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f[0] in "xy")
            print("%-50s %-12s rc %d %.2f s, %.1f GB out = %.1f GB/s" % (env, " ".join(args[:-2]), r.returncode, dt, out_bytes / 1e9, out_bytes / dt / 1e9), flush=True)
            for f in os.listdir(d):
                if f[0] in "xy":
                    os.remove(os.path.join(d, f))
finally:
    shutil.rmtree(d, ignore_errors=True)
