#!/bin/bash
# Run ON THE GPU BOX: issue / co-execution counters over one GEMM shape.   tools/pmc_gemm3.sh <tag> M N K a_kc b_kc
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$1
shift
mkdir -p $OUT
cd /tmp
T="timeout -k 10 100"
i=0
for grp in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    $T rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/j$i -- python3 $ROOT/tools/gemm_one.py "$@" 3 > $OUT/j$i.log 2>&1 || echo "group $i failed: $grp" >> $OUT/summary3.txt
done
python3 $ROOT/tools/pmc_kernel.py $OUT/ gemm_f32 >> $OUT/summary3.txt 2>&1
cat $OUT/summary3.txt
