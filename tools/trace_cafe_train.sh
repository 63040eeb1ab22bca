#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the CAFE training step replayed from its capture; per-step kernel table.
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_cafe_train
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for B in 64 512; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/g$B -- python3 $ROOT/tools/cafe_train_one.py $B graph > $OUT/g$B.log 2>&1
  tail -1 $OUT/g$B.log
  python3 $ROOT/tools/sum_train_trace.py $OUT/g$B 40 level_loss_grad > $OUT/summary_$B.txt 2>&1 || true
  find $OUT/g$B -name "*kernel_trace.csv" -delete
done
cat $OUT/summary_512.txt
