#!/bin/bash
# Run ON THE GPU BOX (via gpurun): everything the committed summaries under profiles/<round>/ are made from.
#   tools/collect_profile.sh <tag>   ->   gpurun_out/prof_<tag>/{trace,fetch,write,pgpr_trace,pgpr_fetch,pgpr_write,
#                                                                 tucker_trace,tucker_mfma,cafe_trace}
# * bench.py under rocprofv3 --kernel-trace --stats (the driver's command minus the CPU baseline) + two separate PMC passes
#   (FETCH_SIZE, WRITE_SIZE) of its scorer / roofline variants.  Every pass writes bench.py's launch log
#   (HW_BENCH_LAUNCH_LOG): the variants of the grouped scorer run under one kernel name, the log says which launches are which,
#   and tools/summarize_profile.py splits the trace by it -- one row per variant.
# * bench_pgpr.py: kernel trace + FETCH_SIZE / WRITE_SIZE passes (bytes of one whole search).
# * tools/bench_tucker.py: kernel trace + matrix-pipe busy cycles.   * bench_cafe.py: kernel trace.
# Counters are collected in their own runs with --kernel-trace only (MI355X_MICROARCH.md 'rocprofv3 PMC slots'); the profiled
# program sits right after `--`.  A pass that fails stops the script (nothing else touches the card after a hang).
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp
T="timeout -k 10 400"
export HW_BENCH_LAUNCH_LOG=$OUT/launch_log_trace.json
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
echo "trace done" > $OUT/progress.txt
# PMC passes: a short timed region (counters serialise the launches), scorer + roofline variants only
ARGS="--steps 5 --warmup 1 --batches-per-step 8 --no-cpu-baseline --no-pgpr --no-sharded --no-tucker --no-cafe"
export HW_BENCH_LAUNCH_LOG=$OUT/launch_log_fetch.json
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
echo "fetch done" >> $OUT/progress.txt
export HW_BENCH_LAUNCH_LOG=$OUT/launch_log_write.json
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1
echo "write done" >> $OUT/progress.txt
unset HW_BENCH_LAUNCH_LOG
# the search's kernels one after another (HW_PGPR_OVERLAP=0): a lone search otherwise runs its action kernels beside the policy
# launch, and a kernel's duration in the trace then includes what ran beside it (bench_pgpr's roofline.kernel_ms is the lone launch)
export HW_PGPR_OVERLAP=0
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pgpr_trace -- python3 $ROOT/bench_pgpr.py --users 256 --steps 10 > $OUT/pgpr_trace.log 2>&1
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pgpr_fetch -- python3 $ROOT/bench_pgpr.py --users 256 --steps 3 > $OUT/pgpr_fetch.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pgpr_write -- python3 $ROOT/bench_pgpr.py --users 256 --steps 3 > $OUT/pgpr_write.log 2>&1
unset HW_PGPR_OVERLAP
echo "pgpr done" >> $OUT/progress.txt
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tucker_trace -- python3 $ROOT/tools/bench_tucker.py --reps 10 > $OUT/tucker_trace.log 2>&1
$T rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/tucker_mfma -- python3 $ROOT/tools/bench_tucker.py --reps 2 > $OUT/tucker_mfma.log 2>&1
echo "tucker done" >> $OUT/progress.txt
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cafe_trace -- python3 $ROOT/bench_cafe.py --users 256 > $OUT/cafe_trace.log 2>&1
echo "cafe done" >> $OUT/progress.txt
ls $OUT
