#!/bin/bash
# bash tools/pmc_counters.sh <tag> "<counters>" <python script and args...>: one --pmc pass, per-kernel sums
TAG=$1; CNT=$2; shift; shift
export TMPDIR=/tmp; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp
timeout 300 rocprofv3 --pmc $CNT --output-format csv -d $O/c -o c -- python3 "$@" > $O/c.log 2>&1
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys, collections
p = glob.glob(sys.argv[1] + "/c/**/*_counter_collection.csv", recursive=True)
if not p:
    print("no counter file"); print(open(sys.argv[1] + "/c.log").read()[-2000:]); sys.exit()
acc = collections.OrderedDict()
for row in csv.DictReader(open(p[0])):
    name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("gt4::", "").replace("km8::", "").replace("km32::", "").split("(")[0]
    if not (name.startswith("k_pair_merge") or name.startswith("k_nway_merge")): continue
    key = (row["Dispatch_Id"], name)
    acc.setdefault(key, {}).setdefault(row["Counter_Name"], 0.0)
    acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
for (d, n), c in acc.items():
    print(d, n, " ".join("%s=%.4g" % kv for kv in sorted(c.items())))
PY
