#!/bin/bash
# One rocprofv3 counter pass over a bench.py configuration.
# usage: tools/pmc.sh <tag> "<COUNTER ...>" <bench args...>   -> gpurun_out/pmc_<tag>/
set -u
tag=$1; counters=$2; shift 2
out=$PWD/gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
repo=$PWD
( cd /tmp && timeout 90 rocprofv3 --pmc $counters --output-format csv -d "$out" -- python "$repo/bench.py" "$@" --no-cpu-baseline --also none > "$out/run.log" 2>&1 )
python3 - "$out" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    if not k.startswith("k_trace") and not k.startswith("k_shade") and not k.startswith("k_gen"):
        continue
    print(k, "dispatches", n[k], " ".join(f"{c}={v:.4g}" for c, v in sorted(acc[k].items())))
PY
