#!/bin/bash
# bash tools/exp/run_merge_tree_probe.sh <tag>: the in-LDS merge tree probe (exp_lds_merge_tree.hip) on the GPU box --
# exactness + time per tile for four key distributions, then two PMC passes (stride, genomic).  Output: gpurun_out/<tag>/.
TAG=${1:-r6_merge_tree}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
B=$R/tools/exp/exp_lds_merge_tree
[ -x $B ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/exp/exp_lds_merge_tree.hip -o $B
for d in stride iid genomic clustered; do
  timeout 120 $B $d 2000 > $O/time_$d.log 2>&1; echo "== $d"; cat $O/time_$d.log
done
timeout 120 $B stride 2000 time 3072 > $O/time_stride_3072.log 2>&1; echo "== stride, 3072 records per tile"; cat $O/time_stride_3072.log
export TMPDIR=/tmp; cd /tmp
C1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
C2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS"
for d in stride genomic; do
  timeout 300 rocprofv3 --pmc $C1 --output-format csv -d $O/pmc1_$d -o c -- $B $d 500 pmc > $O/pmc1_$d.log 2>&1
  timeout 300 rocprofv3 --pmc $C2 --output-format csv -d $O/pmc2_$d -o c -- $B $d 500 pmc > $O/pmc2_$d.log 2>&1
done
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for d in ("stride", "genomic"):
    acc = collections.OrderedDict()
    for p in sorted(glob.glob(O + "/pmc?_%s/**/*_counter_collection.csv" % d, recursive=True)):
        for row in csv.DictReader(open(p)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc.setdefault(name, collections.OrderedDict()).setdefault(row["Counter_Name"], 0.0)
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
    for name, c in acc.items():
        wt = 256 * 16 * 500.0  # wavefronts x tiles per launch (one workgroup per CU)
        print(d, name, " ".join("%s=%.4g" % kv for kv in c.items()))
        print("   per wavefront and tile: " + " ".join("%s=%.1f" % (k.replace("SQ_", ""), v / wt) for k, v in c.items() if k.startswith("SQ_INSTS") or k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS")))
PY
