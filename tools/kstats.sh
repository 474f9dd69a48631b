#!/bin/bash
# Per-kernel time of one bench.py configuration (rocprofv3 --kernel-trace --stats), printed as a table and kept under gpurun_out/kstats_<tag>/.
# usage: tools/kstats.sh <tag> <bench args...>
set -u
tag=$1; shift
repo=$PWD; out=$repo/gpurun_out/kstats_$tag
mkdir -p "$out"; export TMPDIR=/tmp
( cd /tmp && timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python "$repo/bench.py" "$@" --no-cpu-baseline --also none > "$out/run.log" 2>&1 )
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
cp "$f" "$out/kernel_stats.csv" 2>/dev/null
python3 - "$out/kernel_stats.csv" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    name = re.sub(r"pt::\(anonymous namespace\)::", "", r["Name"]); name = re.sub(r"\(.*", "", name)[:60]
    print(f'{name:60s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:10.1f} us total {float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
tail -1 "$out/run.log" | cut -c1-400
