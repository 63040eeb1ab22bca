#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py, and
# the kernel trace + one PMC pass (matrix-pipe busy cycles) of the TuckER step.  Counters are collected in their own runs
# with --kernel-trace only (MI355X_MICROARCH.md 'rocprofv3 PMC slots'); the profiled program sits right after `--`.
#   tools/collect_profile.sh <tag>        -> gpurun_out/prof_<tag>/{trace,fetch,write,tucker_trace,tucker_mfma}
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp
# the same command the driver runs (default flags), minus the CPU baseline
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
echo "trace done" > $OUT/progress.txt
# PMC passes: a short timed region (counters serialise the launches), scorer + roofline variants only
ARGS="--steps 5 --warmup 1 --batches-per-step 8 --no-cpu-baseline --no-pgpr --no-sharded --no-tucker"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
echo "fetch done" >> $OUT/progress.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1
echo "write done" >> $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tucker_trace -- python3 $ROOT/tools/bench_tucker.py --reps 10 > $OUT/tucker_trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/tucker_mfma -- python3 $ROOT/tools/bench_tucker.py --reps 2 > $OUT/tucker_mfma.log 2>&1
echo "tucker done" >> $OUT/progress.txt
ls -R $OUT | head -40
