#!/usr/bin/env python3
"""Joins the known byte counts of tools/calib_fetch.hip (its stdout, one JSON line per kernel) with the raw FETCH_SIZE / WRITE_SIZE
counters of the same binary under `rocprofv3 --pmc` and writes the calibration bench.py's `roofline.traffic` quotes.
usage: tools/calib_fetch_report.py <stdout.txt> <pmc_fetch_dir> <pmc_write_dir> > profiles/r03_fetch_calibration.json"""
import csv, glob, json, os, sys
from collections import defaultdict

known = {}
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        k = json.loads(line)
        known[k["kernel"]] = k


def counters(root, name):
    acc, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"].split("(")[0].strip()
            k = k.replace("void ", "")
            acc[k] += float(r["Counter_Value"])
            n[k] += 1
    return {k: acc[k] * 1024.0 / n[k] for k in acc}  # KB -> bytes per dispatch


fetch, write = counters(sys.argv[2], "FETCH_SIZE"), counters(sys.argv[3], "WRITE_SIZE")
out = {"tool": "tools/calib_fetch.hip (6 GiB buffer, every line touched once per launch; beyond L2 and the Infinity Cache)",
       "meaning": "ratio = counter bytes per dispatch / bytes of the 64-B lines the kernel touches; a true-traffic estimate divides the raw counter by it",
       "kernels": {}}
for name, k in known.items():
    e = dict(k)
    for cname, table in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        raw = next((v for kk, v in table.items() if kk.startswith(name)), None)
        if raw is not None:
            e[cname + "_bytes_per_dispatch"] = round(raw)
            e[cname + "_over_lines64"] = round(raw / k["lines64_bytes"], 4)
            e[cname + "_over_asked"] = round(raw / k["asked_bytes"], 4)
    out["kernels"][name] = e
K = out["kernels"]


def ratio(kernel, counter):
    return K.get(kernel, {}).get(counter + "_over_lines64")


out["factors"] = {"fetch_stream16": ratio("calib_stream_read16", "FETCH_SIZE"), "fetch_gather16": ratio("calib_gather<1>", "FETCH_SIZE"),
                  "fetch_gather48": ratio("calib_gather<3>", "FETCH_SIZE"), "fetch_gather80": ratio("calib_gather<5>", "FETCH_SIZE"),
                  "write_stream16": ratio("calib_stream_write16", "WRITE_SIZE"), "write_scatter16": ratio("calib_scatter_write16", "WRITE_SIZE")}
print(json.dumps(out, indent=1))
