#!/bin/bash
# Round-3 GPU call Y: core texture slots in one load; the GPU suite and the four workloads on the resulting build.
cd "$(dirname "$0")/.."; ulimit -c 0
O=$PWD/gpurun_out; mkdir -p $O
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    k = j["kernels"]
    print(f"RESULT {tag:18s} value {j['value']:9.2f} shade_first {k['shade_first']['ms_per_frame']:.4f} shade {k['shade']['ms_per_frame']:.4f} finish {k['finish_sample']['ms_per_frame']:.4f} shadow {k['trace_shadow']['ms_per_frame']:.4f}")
except Exception as e:
    print("RESULT", tag, "FAILED", e)
PY
}
run() { tag=$1; shift; timeout 300 "$@" > $O/r03y_$tag.json 2>$O/r03y_$tag.err; summ $tag $O/r03y_$tag.json; }
N="--no-cpu-baseline --also none"
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
run helmet python bench.py --workload helmet --steps 6 --warmup 1 $N
run atrium python bench.py --workload atrium --steps 3 --warmup 1 $N
run street python bench.py --workload street --steps 2 --warmup 1 $N
run glass python bench.py --workload glass --steps 1 --warmup 1 $N
run helmet4k python bench.py --workload helmet --width 3840 --height 2160 --in-flight 64 --steps 3 --warmup 1 $N
