#!/bin/bash
# usage: run_variants.sh <outdir> "<exp.py arguments>" variant...   ("" = the product library)
O=gpurun_out/$1; ARGS=$2; shift; shift; mkdir -p $O
for v in "$@"; do
  echo "== ${v:-product}" | tee -a $O/variants.log
  if [ -n "$v" ]; then export GT4HIP_LIB=$PWD/genometester4_amd/libgt4hip_$v.so; else unset GT4HIP_LIB; fi
  timeout 300 python tools/exp.py $ARGS 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee -a $O/variants.log
done
