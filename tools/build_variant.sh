#!/bin/bash
# Experiment helper (not product): build the kernels file with extra -D switches into
# genometester4_amd/libgt4hip_<name>.so (select it with GT4HIP_LIB=<path>); the other objects come
# from the normal build.   bash tools/build_variant.sh <name> -DGT4_FETCH_TOP=3 ...
set -e
NAME=$1; shift
cd "$(dirname "$0")/../genometester4_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-value -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" -c gt4hip_kernels.hip -o /tmp/gt4hip_kernels.$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/gt4hip_kernels.$NAME.o gt4hip_kway.hip.o gt4hip_sort.hip.o gt4hip_api.hip.o gt4hip_io.hip.o gt4hip_comm.hip.o gt4_listfile.o gt4_setops.o -o ../libgt4hip_$NAME.so -lpthread -ldl
echo built libgt4hip_$NAME.so
