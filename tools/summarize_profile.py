#!/usr/bin/env python
"""Turn gpurun_out/prof_<tag>/ (made by tools/collect_profile.sh) into committed summaries under profiles/<round>/:

  bench_kernel_stats_<tag>.csv     rocprofv3's per-kernel statistics of the whole bench.py run
  bench_variants_<tag>.csv         ONE ROW PER VARIANT of the grouped RotatE scorer: the kernel trace split by bench.py's
                                   launch log (hbm_fused = the roofline of record, hbm_plain, entity_tails_*, and the timed
                                   region's launches) with launches / mean / median / min / max duration
  pmc_traffic_<tag>.json           FETCH_SIZE / WRITE_SIZE per kernel and per variant (bytes per launch)
  pgpr_kernel_stats_<tag>.csv      per-kernel statistics of bench_pgpr.py (256 users), + pgpr traffic per search
  tucker_kernel_stats_<tag>.csv, cafe_kernel_stats_<tag>.csv
  profiles/traffic_latest.json     what bench.py reads for roofline.traffic

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide (16 B/lane)
read, i.e. half the bytes -- it is doubled here as MI355X_MICROARCH.md §HBM prescribes.

    python tools/summarize_profile.py <tag> <round>"""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

tag, rnd = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = glob.glob(os.path.join(src, pattern))
    return f[0] if f else None


for sub, name in (("trace", "bench"), ("pgpr_trace", "pgpr"), ("tucker_trace", "tucker"), ("cafe_trace", "cafe")):
    st = one(f"{sub}/*/*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(dst, f"{name}_kernel_stats_{tag}.csv"))


def split_by_log(values_in_launch_order, log, kernel_substr):
    """values of ONE kernel name in launch order -> {tag: [values]} following the launch log's entries for that name; launches
    before / between logged runs (the timed region, warm-ups of other legs) are dropped into 'unlogged'."""
    out, pos = collections.OrderedDict(), 0
    ent = [e for e in log if e["kernel"] and e["kernel"] in kernel_substr]
    # bench.py logs every launch of these kernels from its first one on (timed region included); what follows the last
    # logged run (later legs that go through the model classes) is 'unlogged'
    for e in ent:
        out.setdefault(e["tag"], []).extend(values_in_launch_order[pos : pos + e["launches"]])
        pos += e["launches"]
    out["unlogged"] = values_in_launch_order[pos:]
    return out


def read_log(name):
    p = os.path.join(src, name)
    return json.load(open(p)) if os.path.exists(p) else []


# ---- one row per variant: durations from the kernel trace --------------------------------------------------------------
trace = one("trace/*/*kernel_trace.csv")
log = read_log("launch_log_trace.json")
variant_rows = []
if trace and log:
    by_name = collections.defaultdict(list)
    for r in sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"])):
        by_name[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name, vals in by_name.items():
        if "kge_score_grouped" not in name:
            continue
        parts = split_by_log(vals, log, name)
        for vt, v in parts.items():
            if not v:
                continue
            # the first 3 launches of a _time_kernel run are its warm-up
            body = v[3:] if len(v) > 8 and vt not in ("timed_region_and_single_stream_pass", "unlogged") else v
            variant_rows.append(dict(variant=vt, kernel=name.split("(")[0], launches=len(body),
                                     mean_us=round(statistics.fmean(body), 3), median_us=round(statistics.median(body), 3),
                                     min_us=round(min(body), 3), max_us=round(max(body), 3)))
    with open(os.path.join(dst, f"bench_variants_{tag}.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(variant_rows[0].keys()))
        w.writeheader()
        w.writerows(variant_rows)


def per_kernel(path, counter):
    """{kernel name: [counter value per launch, in launch order]}"""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    key = "Start_Timestamp" if rows and "Start_Timestamp" in rows[0] else "Dispatch_Id"
    agg = collections.defaultdict(list)
    for r in sorted(rows, key=lambda r: int(r[key])):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


out = {}
bytes_of = lambda fk, wk: (2.0 * fk + wk) * 1024.0
med = lambda xs: sorted(xs)[len(xs) // 2] if xs else 0.0
fetch, write = one("fetch/*/*counter_collection.csv"), one("write/*/*counter_collection.csv")
if fetch and write:
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    flog, wlog = read_log("launch_log_fetch.json"), read_log("launch_log_write.json")
    rows = []
    for k in sorted(set(f) | set(w)):
        if not k.startswith(("hw::", "void hw::")):
            continue
        fv, wv = f.get(k, [0.0]), w.get(k, [0.0])
        row = dict(kernel=k.split("(")[0], launches=len(fv), fetch_kib_raw_median=med(fv), fetch_kib_raw_max=max(fv),
                   write_kib_median=med(wv), write_kib_max=max(wv))
        if "kge_score_grouped" in k and flog:
            fp, wp = split_by_log(fv, flog, k), split_by_log(wv, wlog, k)
            row["variants"] = {vt: dict(launches=len(fp[vt]), bytes_per_launch=bytes_of(med(fp[vt]), med(wp.get(vt, [0.0]))))
                               for vt in fp if fp[vt]}
            names = {"hbm_fused": "kge_score_grouped_hbm_fused_bytes_per_launch",
                     "hbm_plain": "kge_score_grouped_hbm_variant_bytes_per_launch",
                     "entity_tails_plain": "kge_score_grouped_entity_tails_plain_bytes_per_launch",
                     "entity_tails_xcd_partitioned": "kge_score_grouped_entity_tails_xcd_bytes_per_launch"}
            for vt, v in row["variants"].items():
                if vt.startswith("scorer_hbm_"):  # the other scorers' HBM launches (bench.py: leg_scorer_rooflines)
                    out[f"{vt}_bytes_per_launch"] = v["bytes_per_launch"]
                elif vt in names:
                    out[names[vt]] = v["bytes_per_launch"]
                elif vt == "timed_region_and_single_stream_pass":
                    out["kge_score_grouped_bytes_per_launch"] = v["bytes_per_launch"]
        rows.append(row)
    json.dump(rows, open(os.path.join(dst, f"pmc_traffic_{tag}.json"), "w"), indent=1)

# ---- PGPR: fabric-side bytes of one whole search ----------------------------------------------------------------------
pf, pw = one("pgpr_fetch/*/*counter_collection.csv"), one("pgpr_write/*/*counter_collection.csv")
if pf and pw:
    f, w = per_kernel(pf, "FETCH_SIZE"), per_kernel(pw, "WRITE_SIZE")
    # one launch of the first action pass per hop, three hops per search (whatever the layout of the hop's tail)
    searches_f = max(1, sum(len(v) for k, v in f.items() if "pgpr_actions_small_kernel" in k) // 3)
    searches_w = max(1, sum(len(v) for k, v in w.items() if "pgpr_actions_small_kernel" in k) // 3)
    # bench_pgpr.py also runs the collector and a full rebuild of the leaves once (pgpr.leaf_layout): not part of a search
    not_search = ("collect_", "pgpr_materialize_leaves")
    per = []
    for k in sorted(set(f) | set(w)):
        if not k.startswith(("hw::", "void hw::")) or any(x in k for x in not_search):
            continue
        b = 2.0 * 1024.0 * sum(f.get(k, [])) / searches_f + 1024.0 * sum(w.get(k, [])) / searches_w
        per.append(dict(kernel=k.split("(")[0], bytes_per_search=b, launches_per_search=len(f.get(k, [])) / searches_f))
    per.sort(key=lambda r: -r["bytes_per_search"])
    out["pgpr_search_bytes_per_search"] = sum(r["bytes_per_search"] for r in per)
    json.dump(dict(searches_counted=dict(fetch=searches_f, write=searches_w), kernels=per,
                   total_bytes_per_search=out["pgpr_search_bytes_per_search"]),
              open(os.path.join(dst, f"pgpr_traffic_{tag}.json"), "w"), indent=1)

if out:
    out["note"] = ("fabric-side bytes per launch (FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE): requests that reach HBM "
                   "or hit the Infinity Cache.  Variants of the grouped scorer are told apart by bench.py's launch log "
                   "(profiles/<round>/bench_variants_<tag>.csv has their durations); pgpr_search_bytes_per_search sums every "
                   "kernel of one 256-user search")
    out["source"] = f"profiles/{rnd}/pmc_traffic_{tag}.json"
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
for r in variant_rows:
    print(r)
print(json.dumps(out, indent=1))
