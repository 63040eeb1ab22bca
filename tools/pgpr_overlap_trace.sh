#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the PGPR search bench with 1 and with 3 streams; per kernel the launch count, the mean
# duration alone and the mean duration while other searches' kernels run beside it -- what overlap costs each kernel.
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_overlap
mkdir -p $OUT
cd /tmp
for s in 1 3; do
  rm -rf $OUT/s$s
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/s$s -- python3 $ROOT/bench_pgpr.py --users 256 --streams $s --steps 12 > $OUT/bench_s$s.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = {}
for s in (1, 3):
    f = glob.glob(f"{out}/s{s}/*/*kernel_trace.csv")[0]
    rows = list(csv.DictReader(open(f)))
    rows = rows[len(rows) // 3:]  # skip warm-up searches
    d = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0][-60:]
        d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
    res[s] = (d, (t1 - t0) / 1e3)
    import os
    os.remove(f)
names = sorted(res[1][0], key=lambda n: -sum(res[1][0][n]))
with open(f"{out}/overlap_summary.txt", "w") as fo:
    for s in (1, 3):
        fo.write(f"streams={s}: wall {res[s][1]:.0f} us, sum of kernel durations {sum(sum(v) for v in res[s][0].values()):.0f} us\n")
    fo.write(f"{'kernel':62s} {'n(1)':>6s} {'mean us (1 stream)':>20s} {'n(3)':>6s} {'mean us (3 streams)':>20s}\n")
    for n in names[:25]:
        a = res[1][0][n]; b = res[3][0].get(n, [0.0])
        fo.write(f"{n:62s} {len(a):6d} {sum(a)/len(a):20.1f} {len(b):6d} {sum(b)/len(b):20.1f}\n")
print(open(f"{out}/overlap_summary.txt").read())
PY
