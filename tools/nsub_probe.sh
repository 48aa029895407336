#!/bin/bash
# bash tools/nsub_probe.sh <tag> [dist]: N-way tests, union8 timing (new and old tile kernel), phase stamps of three
# wavefronts (diagnostic builds libgt4hip_prof{0,896,960}.so), two PMC passes.  Output under gpurun_out/<tag>/.
TAG=$1; DIST=${2:-stride}
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_kway.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/tests.log
B="python bench.py --workload union8 --dist $DIST --steps 5 --warmup 2 --no-cpu-baseline"
GT4HIP_KWAY_SUB=1 $B > $O/new.log 2>&1
GT4HIP_KWAY_SUB=0 $B > $O/old.log 2>&1
export GT4HIP_KWAY_SUB=1
for t in 0 896 960; do
  if [ -f genometester4_amd/libgt4hip_prof$t.so ]; then GT4HIP_LIB=$PWD/genometester4_amd/libgt4hip_prof$t.so $B 2>&1 | grep "nway phases" | tail -1 > $O/stamps_$t.log; fi
done
A="$PWD/bench.py --workload union8 --dist $DIST --steps 2 --warmup 1 --no-cpu-baseline"
C1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
C2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS"
bash tools/pmc_counters.sh $TAG/pmc1 "$C1" $A | tail -1 > $O/pmc.log
bash tools/pmc_counters.sh $TAG/pmc2 "$C2" $A | tail -1 >> $O/pmc.log
GT4HIP_KWAY_SUB=1 python tools/nsub_count.py > $O/count.log 2>&1; GT4HIP_KWAY_SUB=0 python tools/nsub_count.py > $O/count_old.log 2>&1
echo "count new: $(grep kernel $O/count.log | tr "\n" " ")"; echo "count old: $(grep kernel $O/count_old.log | tr "\n" " ")"
echo "tests: $(tail -1 $O/tests.log)"
echo "new: $(grep -o 'kernel_ms_avg[^,]*' $O/new.log) $(grep -o 'device_ms_avg[^,]*' $O/new.log) $(grep -o '"self_check[^}]*' $O/new.log)"
echo "old: $(grep -o 'kernel_ms_avg[^,]*' $O/old.log) $(grep -o 'device_ms_avg[^,]*' $O/old.log)"
cat $O/pmc.log
for t in 0 896 960; do [ -f $O/stamps_$t.log ] && python3 - $O/stamps_$t.log <<'PY'
import sys,re
s=open(sys.argv[1]).read()
items=re.findall(r' ([ws]:[^%]*?) ([0-9.]+)%', s)
print(sys.argv[1].split('_')[-1], ' '.join('%s=%s'%(a.strip(),b) for a,b in items if float(b)>=0.5), re.findall(r'avg cycles/tile \d+', s))
PY
done
