#!/bin/bash
# usage: run_sort_variants.sh <outdir> <n> variant...
O=gpurun_out/$1; N=$2; shift; shift; mkdir -p $O
for v in "$@"; do
  echo "== $v" | tee -a $O/variants.log
  GT4HIP_LIB=$PWD/genometester4_amd/libgt4hip_$v.so REPS=${REPS:-2} timeout 300 python tools/exp_sort.py $N 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee -a $O/variants.log
done
