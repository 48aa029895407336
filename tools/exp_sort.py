"""Experiment driver (not product): gt4hip_device_words_to_list on random words; prints sort and fold times.
usage: exp_sort.py [n words] [k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genometester4_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 25
ctx = capi.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
pristine = torch.randint(0, 1 << min(2 * k, 62), (n,), dtype=torch.int64, device="cuda", generator=g)
work = torch.empty_like(pristine)
for rep in range(int(os.environ.get("REPS", "3"))):
    work.copy_(pristine); torch.cuda.synchronize()
    lst = ctx.device_words_to_list(work.data_ptr(), n, k)
    print("n", n, "k", k, "sort ms %.2f fold ms %.2f" % (ctx.get_counter("sort_us") / 1000.0, ctx.get_counter("fold_us") / 1000.0), "records", lst.n_words, "sorted", lst.is_sorted(), "sum", lst.sum_counts() == n, flush=True)
    lst.free()
