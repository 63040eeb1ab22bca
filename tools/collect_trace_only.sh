#!/bin/bash
# Run ON THE GPU BOX: the kernel-trace pass of tools/collect_profile.sh alone, with the per-launch trace reduced to the rows
# tools/summarize_profile.py splits by bench.py's launch log (the grouped scorer's launches) -- the full trace of a bench run is
# tens of MB, the merge back from the box is capped.   tools/collect_trace_only.sh <tag>
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp
export HW_BENCH_LAUNCH_LOG=$OUT/launch_log_trace.json
rm -rf $OUT/trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
for f in $OUT/trace/*/*kernel_trace.csv; do
  python3 - "$f" <<'PY'
import csv, sys
p = sys.argv[1]
rows = list(csv.DictReader(open(p)))
keep = [r for r in rows if "kge_score_grouped" in r["Kernel_Name"]]
with open(p, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(keep)
print(p, len(rows), "->", len(keep))
PY
done
tail -1 $OUT/bench_trace.log | cut -c1-300
