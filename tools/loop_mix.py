"""Experiment helper (not product): static instruction mix of the merge kernels' main loops.
    python tools/loop_mix.py [extra hipcc flags...]
Compiles csrc/gt4hip_kernels.hip to gfx950 assembly and counts, per selected kernel, the scalar-ALU,
vector-ALU, LDS and memory instructions of the tile loop (s_waitcnt / s_nop / branches listed apart)."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "genometester4_amd", "csrc", "gt4hip_kernels.hip")
out = "/tmp/loop_mix.s"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
                       "--cuda-device-only", "-S", src, "-o", out] + sys.argv[1:], stderr=subprocess.DEVNULL)
L = open(out).read().split("\n")
starts = [i for i, l in enumerate(L) if re.match(r"^_ZN\S*k_pair_merge\S*:", l)]
names = subprocess.run(["c++filt"], input="\n".join(L[i].split(":")[0] for i in starts), capture_output=True, text=True).stdout.split("\n")
WANT = ["<1024, 6, 1, 2, 1, 0>", "<1024, 4, 1, 1, 1, 0>", "<1024, 4, 1, 0, 1, 5>", "<1024, 4, 1, 4, 1, 0>", "<512, 4, 0, 2, 1, 0>"]
for n, (i, nm) in enumerate(zip(starts, names)):
    w = [x for x in WANT if x in nm]
    if not w:
        continue
    end = starts[n + 1] if n + 1 < len(starts) else len(L)
    B = L[i:end]
    hds = [j for j, l in enumerate(B) if "Loop Header: Depth=1" in l]
    if not hds:
        continue
    hd = hds[0]
    nxt = [j for j, l in enumerate(B) if "Loop Header: Depth=1" in l and j > hd]
    body = B[hd:nxt[0]] if nxt else B[hd:]
    c = collections.Counter()
    for l in body:
        t = l.strip().split()
        if not t or t[0].startswith(";") or t[0].startswith("."):
            continue
        op = t[0]
        if op in ("s_waitcnt", "s_nop"):
            c["wait/nop"] += 1
        elif op.startswith("s_cbranch") or op == "s_branch":
            c["branch"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        else:
            c["mem"] += 1
    print("%-24s salu %4d  branch %3d  wait/nop %3d  valu %4d  lds %3d  mem %2d" % (w[0], c["salu"], c["branch"], c["wait/nop"], c["valu"], c["lds"], c["mem"]))
