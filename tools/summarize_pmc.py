#!/usr/bin/env python3
"""Aggregates rocprofv3 CSV output (kernel stats + counter_collection) per kernel name.
usage: tools/summarize_pmc.py gpurun_out/prof_<tag> [out.json]"""
import csv, glob, json, os, re, sys
from collections import defaultdict

root = sys.argv[1]
res = {"kernels": defaultdict(dict)}


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = res["kernels"][short(r["Name"])]
        k["calls"] = int(r["Calls"]); k["total_ms"] = float(r["TotalDurationNs"]) / 1e6; k["avg_us"] = float(r["AverageNs"]) / 1e3
        k["pct"] = float(r["Percentage"])
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"]); n[short(r["Kernel_Name"])][r["Counter_Name"]] += 1
    for kn, cs in acc.items():
        for cn, v in cs.items():
            res["kernels"][kn][cn] = v; res["kernels"][kn]["dispatches_pmc"] = n[kn][cn]
for kn, k in res["kernels"].items():
    if "SQ_WAVE_CYCLES" in k and k["SQ_WAVE_CYCLES"] > 0:
        wc = k["SQ_WAVE_CYCLES"]
        k["frac_wait_any"] = k.get("SQ_WAIT_ANY", 0) / wc; k["frac_wait_inst"] = k.get("SQ_WAIT_INST_ANY", 0) / wc
        k["frac_active"] = k.get("SQ_ACTIVE_INST_ANY", 0) / wc; k["frac_active_valu"] = k.get("SQ_ACTIVE_INST_VALU", 0) / wc
    if "TCC_HIT_sum" in k:
        k["l2_hit_rate"] = k["TCC_HIT_sum"] / max(1.0, k["TCC_HIT_sum"] + k["TCC_MISS_sum"])
    if "FETCH_SIZE" in k:
        k["fetch_bytes_per_dispatch_raw"] = k["FETCH_SIZE"] * 1024 / max(1, k["dispatches_pmc"])  # FETCH_SIZE is in KB
res["kernels"] = dict(res["kernels"])
out = json.dumps(res, indent=1, sort_keys=True)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
print(out)
