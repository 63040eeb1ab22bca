#!/bin/bash
# Run ON THE GPU BOX: counter passes over the kernels of a PGPR beam search (tools/pgpr_one.py), one rocprofv3 --pmc run per
# counter group (counters in their own runs, with --kernel-trace only).   tools/pmc_pgpr.sh <tag>  ->  gpurun_out/pmc_pgpr_<tag>/
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_pgpr_$1
mkdir -p $OUT
cd /tmp
T="timeout -k 10 150"
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    $T rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $ROOT/tools/pgpr_one.py 2 > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp" >> $OUT/summary.txt
    echo "group $i done" >> $OUT/progress.txt
done
for k in pgpr_actions_small pgpr_actions_trim_wave pgpr_actions_trim_block pgpr_topk_expand_kernel pgpr_policy_fused; do
    echo "== $k (launches with grid >= 100000 threads: the last hop)" >> $OUT/summary.txt
    python3 $ROOT/tools/pmc_kernel.py $OUT $k 100000 >> $OUT/summary.txt 2>&1
done
cat $OUT/summary.txt
