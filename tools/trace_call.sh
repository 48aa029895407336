#!/bin/bash
# bash tools/trace_call.sh <tag> <bench.py args...>: rocprofv3 --kernel-trace of a bench line; per-call timeline of the
# LAST step (kernel, start offset, duration, gap to the previous kernel) -> gpurun_out/<tag>/timeline.txt
TAG=$1; shift
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 $R/bench.py "$@" > $O/bench.json 2> $O/bench.err
cd $R
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
p = glob.glob(O + "/t/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(p[0])), key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    return r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("gt4::", "").split("(")[0][:60]
# the last call = from the last k_nway_probe (or the last k_nway_sample after a gap) to the end
idx = [i for i, r in enumerate(rows) if "k_nway_probe" in r["Kernel_Name"] or "k_share_probe" in r["Kernel_Name"]]
start = idx[-1] if idx else max(0, len(rows) - 40)
t0 = int(rows[start]["Start_Timestamp"])
prev_end = t0
out = []
busy = 0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, nm(r)))
    busy += e - s
    prev_end = e
out.append("span %.1f us, kernels busy %.1f us, gaps %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3, (prev_end - t0 - busy) / 1e3))
open(O + "/timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
