#!/bin/bash
# Experiment helper (not product): build ONE source file of csrc with extra -D switches into
# genometester4_amd/libgt4hip_<name>.so (select it with GT4HIP_LIB=<path>); the other objects come
# from the normal build.   bash tools/build_variant.sh <name> [-f gt4hip_nway.hip] -DGT4_FETCH_TOP=3 ...
set -e
NAME=$1; shift
SRC=gt4hip_kernels.hip
if [ "$1" = "-f" ]; then SRC=$2; shift; shift; fi
cd "$(dirname "$0")/../genometester4_amd/csrc"
SCHED=""; if [ "$SRC" = "gt4hip_kernels.hip" ]; then SCHED="-mllvm -amdgpu-sched-strategy=iterative-ilp"; fi
if [ "$SRC" = "gt4hip_nway.hip" ]; then SCHED="-mllvm -amdgpu-sched-strategy=iterative-maxocc"; fi  # (= the Makefile's KERNELS_SCHED / NWAY_SCHED; GT4_NO_SCHED=1: neither)
if [ -n "$GT4_NO_SCHED" ]; then SCHED=""; fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-value -mllvm -amdgpu-atomic-optimizer-strategy=None $SCHED "$@" -c $SRC -o /tmp/${SRC%.hip}.$NAME.o
OBJS=""
for f in gt4hip_kernels gt4hip_nway gt4hip_sort gt4hip_api gt4hip_io gt4hip_comm; do
  if [ "$f.hip" = "$SRC" ]; then OBJS="$OBJS /tmp/$f.$NAME.o"; else OBJS="$OBJS $f.hip.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS gt4_listfile.o gt4_setops.o gt4_shard.o -o ../libgt4hip_$NAME.so -lpthread -ldl
echo built libgt4hip_$NAME.so
