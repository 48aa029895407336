#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of a kernel source of csrc as hipcc sees it
(-Rpass-analysis=kernel-resource-usage; the build's own flags).
Usage: tools/kernel_resources.py [substring] [source file in csrc]
tests/test_kernel_resources.py asserts on the same table: no kernel may spill a vector register or use scratch."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "genometester4_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]  # = csrc/Makefile's HIPFLAGS


PER_SOURCE = {"gt4hip_kernels.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],      # = csrc/Makefile's KERNELS_SCHED
              "gt4hip_nway.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]}      # ... NWAY_SCHED


def table(source="gt4hip_kernels.hip", extra=()):
    """[{name, vgpr, sgpr, vspill, sspill, scratch, lds, occ}] for every kernel of csrc/<source>"""
    extra = list(PER_SOURCE.get(source, [])) + list(extra)
    r = subprocess.run(["hipcc"] + FLAGS + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", os.path.join(CSRC, source), "-o", "/dev/null"],
                       capture_output=True, text=True, cwd=CSRC)
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    if not blocks:
        raise RuntimeError("hipcc printed no resource remarks for %s:\n%s" % (source, r.stderr[-2000:]))
    names = [b.split("\n")[0].split()[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
    rows = []
    for b, n in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        n = re.sub(r"\(.*", "", n.replace("void ", "").replace("gt4::(anonymous namespace)::", "").replace("gt4::", ""))
        rows.append(dict(name=n, vgpr=g(" VGPRs"), sgpr=g("TotalSGPRs"), vspill=g("VGPRs Spill"), sspill=g("SGPRs Spill"),
                         scratch=g(r"ScratchSize \[bytes/lane\]"), lds=g(r"LDS Size \[bytes/block\]"), occ=g(r"Occupancy \[waves/SIMD\]")))
    return rows


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    src = sys.argv[2] if len(sys.argv) > 2 else "gt4hip_kernels.hip"
    print("%-46s %5s %5s %6s %6s %7s %7s %4s" % ("kernel", "VGPR", "SGPR", "vspill", "sspill", "scratch", "LDS", "occ"))
    for r in table(src):
        if pat in r["name"]:
            print("%-46s %5d %5d %6d %6d %7d %7d %4d" % (r["name"], r["vgpr"], r["sgpr"], r["vspill"], r["sspill"], r["scratch"], r["lds"], r["occ"]))


if __name__ == "__main__":
    main()
