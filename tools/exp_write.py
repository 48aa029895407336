"""Experiment driver (not product): HBM -> file rate of gt4hip_list_write_fd into /dev/shm for different
numbers of copy threads, piece sizes and with / without the shared-mapping write."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometester4_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000_000
path = "/dev/shm/gt4_exp_write.bin"
for threads, piece, mm in ((8, 8, 0), (8, 8, 1), (16, 8, 1), (32, 8, 1), (64, 8, 1), (32, 32, 1), (32, 8, 0), (16, 8, 0)):
    os.environ["GT4HIP_IO_THREADS"] = str(threads)
    os.environ["GT4HIP_IO_PIECE_MB"] = str(piece)
    os.environ["GT4HIP_IO_MMAP"] = str(mm)
    ctx = capi.Context(0)
    lst = ctx.alloc(n, 25)
    ctx.generate(lst, n, 1, 8)
    ctx.synchronize()
    for rep in range(2):
        fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
        t0 = time.perf_counter()
        ctx.write_fd(lst, 0, n, fd, 48)
        dt = time.perf_counter() - t0
        os.close(fd)
        print("threads %2d piece %2d MiB mmap %d rep %d: %.1f GB in %.2f s = %.1f GB/s" % (threads, piece, mm, rep, 12 * n / 1e9, dt, 12 * n / dt / 1e9), flush=True)
    # read it back through the upload path
    fd = os.open(path, os.O_RDONLY)
    t0 = time.perf_counter()
    back = ctx.upload_fd(fd, 48, n, 25) if hasattr(ctx, "upload_fd") else None
    dt = time.perf_counter() - t0
    os.close(fd)
    if back is not None:
        print("   read back: %.1f GB/s" % (12 * n / dt / 1e9), flush=True)
        back.free()
    os.remove(path)
    lst.free()
    ctx.close()
