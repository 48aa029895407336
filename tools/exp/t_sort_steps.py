import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from genometester4_amd import capi
n, k = 1_000_000_000, 25
ctx = capi.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
pristine = torch.randint(0, 1 << 50, (n,), dtype=torch.int64, device="cuda", generator=g)
work = torch.empty_like(pristine)
for rep in range(8):
    work.copy_(pristine); torch.cuda.synchronize()
    t0 = time.perf_counter()
    lst = ctx.device_words_to_list(work.data_ptr(), n, k)
    ctx.synchronize()
    t1 = time.perf_counter()
    s = lst.sum_counts()
    t2 = time.perf_counter()
    lst.free()
    t3 = time.perf_counter()
    print("rep", rep, "call %.1f ms (sort %.1f fold %.1f) sum_counts %.1f ms free %.1f ms" % ((t1 - t0) * 1e3, ctx.get_counter("sort_us") / 1e3, ctx.get_counter("fold_us") / 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
