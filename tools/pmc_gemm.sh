#!/bin/bash
# Run ON THE GPU BOX: counter passes over one GEMM shape (tools/gemm_one.py), one rocprofv3 --pmc run per counter group.
#   tools/pmc_gemm.sh <tag> M N K a_kc b_kc     ->   gpurun_out/pmc_<tag>/<group>/ + gpurun_out/pmc_<tag>/summary.txt
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$1
shift
mkdir -p $OUT
cd /tmp
T="timeout -k 10 200"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
    i=$((i+1))
    $T rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $ROOT/tools/gemm_one.py "$@" 3 > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp" >> $OUT/summary.txt
    echo "group $i done" >> $OUT/progress.txt
done
python3 $ROOT/tools/pmc_kernel.py $OUT gemm_f32 >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
