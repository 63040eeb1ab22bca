#!/bin/bash
# Run ON THE GPU BOX: memory-path counter passes over one GEMM shape (tools/gemm_one.py).
#   tools/pmc_gemm2.sh <tag> M N K a_kc b_kc
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$1
shift
mkdir -p $OUT
cd /tmp
T="timeout -k 10 200"
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    $T rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/h$i -- python3 $ROOT/tools/gemm_one.py "$@" 3 > $OUT/h$i.log 2>&1 || echo "group $i failed: $grp" >> $OUT/summary2.txt
    echo "group $i done" >> $OUT/progress.txt
done
python3 $ROOT/tools/pmc_kernel.py $OUT gemm_f32 >> $OUT/summary2.txt 2>&1
cat $OUT/summary2.txt
