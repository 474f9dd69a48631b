#!/bin/bash
# Round-3 GPU call W: the technique-keyed sort of the later bounces again, now that k_shade no longer waits on flat loads.
cd "$(dirname "$0")/.."; ulimit -c 0
O=$PWD/gpurun_out; mkdir -p $O
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    k = j["kernels"]
    print(f"RESULT {tag:18s} value {j['value']:9.2f} shade_first {k['shade_first']['ms_per_frame']:.4f} shade {k['shade']['ms_per_frame']:.4f} closest {k['trace_closest']['ms_per_frame']:.4f} shadow {k['trace_shadow']['ms_per_frame']:.4f}")
except Exception as e:
    print("RESULT", tag, "FAILED", e)
PY
}
run() { tag=$1; shift; timeout 300 "$@" > $O/r03w_$tag.json 2>$O/r03w_$tag.err; summ $tag $O/r03w_$tag.json; }
N="--no-cpu-baseline --also none"
for m in 0 3 1; do
  export MI_PT_SORT_SIMPLE=$m
  run atrium_sort$m python bench.py --workload atrium --steps 3 --warmup 1 $N
  run street_sort$m python bench.py --workload street --steps 2 --warmup 1 $N
  run helmet_sort$m python bench.py --workload helmet --steps 6 --warmup 1 $N
done
