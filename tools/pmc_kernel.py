#!/usr/bin/env python
"""Sum PMC counters of the launches of one kernel from rocprofv3 --pmc output (counter_collection.csv).
    python tools/pmc_kernel.py <dir> <kernel substring> [min grid size]"""
import collections, csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
min_grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"] and int(r.get("Grid_Size", 0) or 0) >= min_grid:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:32s} launches {n[k]:4d}  sum {tot[k]:.4g}  per launch {tot[k]/n[k]:.4g}")
