#!/bin/bash
# tools/build_gemm_variant.sh <name> [-DFLAG ...]: gemm.hip compiled with the flags, linked with the tree's other objects
# into gpurun_exp/libhopwise_amd_<name>.so (load it with HOPWISE_AMD_LIB=... HW_CTYPES_ONLY=1).  Kernel experiments only.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p gpurun_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c hopwise_amd/csrc/gemm.hip -o gpurun_exp/gemm_$name.o
objs=$(ls hopwise_amd/csrc/*.o | grep -v '/gemm.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_exp/libhopwise_amd_$name.so $objs gpurun_exp/gemm_$name.o
echo gpurun_exp/libhopwise_amd_$name.so
