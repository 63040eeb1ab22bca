#!/bin/bash
# Timing-only variants of the fp32 GEMM's three-way-split loop (csrc/gemm.hip, GEMM_ABL_* switches): each built into its own
# shared library next to this script (GEMM_ABL_ONLY: one instantiation per operand layout, seconds of hipcc instead of a
# minute); tools/gemm_abl/run.py times them round-robin on one shape and checks the result of each.  Not part of the product.
set -e
cd "$(dirname "$0")/../.."
for v in base:"" nostore:"-DGEMM_ABL_NO_STORE" nomma:"-DGEMM_ABL_NO_MMA" noput:"-DGEMM_ABL_NO_PUT" nofetch:"-DGEMM_ABL_NO_FETCH" \
         nosync:"-DGEMM_ABL_NO_SYNC" noread:"-DGEMM_ABL_NO_READ" nolds:"-DGEMM_NO_LDS_EPI" \
         mmaonly:"-DGEMM_ABL_NO_STORE -DGEMM_ABL_NO_PUT -DGEMM_ABL_NO_FETCH -DGEMM_ABL_NO_SYNC -DGEMM_ABL_NO_READ"; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGEMM_ABL_ONLY $flags -Iinclude hopwise_amd/csrc/gemm.hip -o tools/gemm_abl/libgemm_$name.so 2>/dev/null &
done
wait
ls -la tools/gemm_abl/*.so
