#!/bin/bash
# Run ON THE GPU BOX: per-step kernel table of the PGPR training step (1 024 users, FusedAdam, eager launches traced).
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_pgpr_train
rm -rf $OUT; mkdir -p $OUT
cd /tmp
HW_TRAIN_B=${1:-1024} timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/tools/pgpr_train_one.py > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
python3 $ROOT/tools/sum_train_trace.py $OUT/t 45 reinforce > $OUT/summary.txt 2>&1 || true
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/summary.txt
