#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
# Counters are collected in their own runs with --kernel-trace only (MI355X_MICROARCH.md 'rocprofv3 PMC slots').
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp
ARGS="--steps 50 --warmup 5 --no-cpu-baseline --pgpr-steps 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py $ARGS --no-pgpr > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/bench.py $ARGS --no-pgpr > $OUT/bench_write.log 2>&1
ls -R $OUT | head -30
