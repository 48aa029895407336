C1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
C2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS"
for w in intersect c2 union8; do
  X=""; [ $w = intersect ] && X="--no-union8 --no-extras"
  A="$PWD/bench.py --workload $w $X --steps 2 --warmup 1 --no-cpu-baseline"
  echo "== $w"
  bash tools/pmc_counters.sh r6_pmc_inst/${w}_1 "$C1" $A | tail -1
  bash tools/pmc_counters.sh r6_pmc_inst/${w}_2 "$C2" $A | tail -1
done
