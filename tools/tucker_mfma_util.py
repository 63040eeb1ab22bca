#!/usr/bin/env python
"""gpurun_out/prof_<tag>/tucker_mfma (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace on
tools/bench_tucker.py) -> profiles/<round>/tucker_mfma_util_<tag>.json + profiles/tucker_mfma_latest.json (read by
bench.py's tucker leg).

GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs:
mfma_util = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024), per kernel and over the whole run; effective clock = GUI_ACTIVE / 8 / the
dispatch's wall time in the same pass (the chip holds its clock down under MFMA load, so a cycle-based utilisation and the
fraction of the 2.4 GHz peak differ by clock / 2.4)."""
import collections
import csv
import glob
import json
import os
import sys

tag, rnd = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "tucker_mfma", "*", "*counter_collection.csv"))
if not f:
    sys.exit("no counter_collection.csv")
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]) if "Grid_Size" in r else 0)
    per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("End_Timestamp"):
        per[key]["_wall_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
rows, busy_all, act_all = [], 0.0, 0.0
for (k, grid), c in per.items():
    busy, act = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), sum(c.get("GRBM_GUI_ACTIVE", [0]))
    busy_all += busy
    act_all += act
    n = max(len(c.get("GRBM_GUI_ACTIVE", [])), 1)
    if busy > 0:
        wall = sum(c.get("_wall_ns", [0]))
        rows.append(dict(kernel=k, grid_threads=grid, launches=n, gpu_cycles_per_launch=act / 8 / n,
                         mfma_busy_cycles_per_launch=busy / n, mfma_util=busy / (act / 8 * 1024),
                         wall_us_per_launch=wall / n / 1e3, effective_clock_ghz=(act / 8 / wall) if wall else None))
rows.sort(key=lambda r: -r["mfma_busy_cycles_per_launch"] * r["launches"])
out = {
    "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 tools/bench_tucker.py --reps 2 "
              "(TuckER d=200, iFashion shape, B=2048, forward and forward+backward)",
    "method": "GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs; "
              "mfma_util = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).  whole_run counts every kernel of the run, non-GEMM ones included",
    "whole_run_mfma_util": busy_all / (act_all / 8 * 1024) if act_all else None,
    "gemm_launches_mfma_util": (sum(r["mfma_busy_cycles_per_launch"] * r["launches"] for r in rows)
                                / sum(r["gpu_cycles_per_launch"] * r["launches"] * 1024 for r in rows)) if rows else None,
    # the chip lowers its clock under MFMA load (MI355X_MICROARCH.md, DVFS give-back): cycles / wall of the same pass
    "gemm_launches_effective_clock_ghz": (sum(r["gpu_cycles_per_launch"] * r["launches"] for r in rows if r["wall_us_per_launch"])
                                          / sum(r["wall_us_per_launch"] * 1e3 * r["launches"] for r in rows if r["wall_us_per_launch"]))
    if any(r["wall_us_per_launch"] for r in rows) else None,
    "kernels": rows,
}
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
json.dump(out, open(os.path.join(dst, f"tucker_mfma_util_{tag}.json"), "w"), indent=1)
json.dump({k: out[k] for k in ("whole_run_mfma_util", "gemm_launches_mfma_util", "gemm_launches_effective_clock_ghz", "method")} | {"source": f"profiles/{rnd}/tucker_mfma_util_{tag}.json"},
          open(os.path.join(ROOT, "profiles", "tucker_mfma_latest.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("whole_run_mfma_util", "gemm_launches_mfma_util", "gemm_launches_effective_clock_ghz")}))
