#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of gt4hip_kernels.hip as hipcc sees it
(-Rpass-analysis=kernel-resource-usage).  Usage: tools/kernel_resources.py [substring] [source file in csrc]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "genometester4_amd", "csrc", "gt4hip_kernels.hip")


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    src = os.path.join(os.path.dirname(SRC), sys.argv[2]) if len(sys.argv) > 2 else SRC
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm",
                        "-amdgpu-atomic-optimizer-strategy=None", "-Rpass-analysis=kernel-resource-usage",
                        "-c", src, "-o", "/dev/null"], capture_output=True, text=True, cwd=os.path.dirname(SRC))
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    names = [b.split("\n")[0].split()[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
    print("%-46s %5s %5s %6s %6s %7s %7s %4s" % ("kernel", "VGPR", "SGPR", "vspill", "sspill", "scratch", "LDS", "occ"))
    for b, n in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        n = re.sub(r"\(.*", "", n.replace("void gt4::(anonymous namespace)::", ""))
        if pat in n:
            print("%-46s %5d %5d %6d %6d %7d %7d %4d" % (n, g(" VGPRs"), g("TotalSGPRs"), g("VGPRs Spill"), g("SGPRs Spill"),
                                                        g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]"),
                                                        g(r"Occupancy \[waves/SIMD\]")))


if __name__ == "__main__":
    main()
