#!/bin/bash
# Round-3 GPU call H: micro-tile-major path slots: full GPU suite + the four workloads.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    k = j["kernels"]
    print(f"RESULT {tag:18s} value {j['value']:9.2f} closest {k['trace_closest']['ms_per_frame']:.4f} shade {k['shade']['ms_per_frame']:.4f} shadow {k['trace_shadow']['ms_per_frame']:.4f} primary {k['trace_primary']['ms_per_frame']:.4f} first {k['shade_first']['ms_per_frame']:.4f}")
except Exception as e:
    print("RESULT", tag, "FAILED", e)
PY
}
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03h_gputest.txt 2>&1; echo "GPUTEST rc=$? $(tail -1 $O/r03h_gputest.txt)"
run() { tag=$1; shift; timeout 200 "$@" > $O/r03h_$tag.json 2>$O/r03h_$tag.err; summ $tag $O/r03h_$tag.json; }
run helmet python bench.py --workload helmet --steps 6 --warmup 1 --no-cpu-baseline --also none
run atrium python bench.py --workload atrium --steps 3 --warmup 1 --no-cpu-baseline --also none
run street python bench.py --workload street --steps 2 --warmup 1 --no-cpu-baseline --also none
run glass python bench.py --workload glass --steps 2 --warmup 1 --no-cpu-baseline --also none
run helmet4k python bench.py --workload helmet --width 3840 --height 2160 --steps 3 --warmup 1 --no-cpu-baseline --also none
run helmet_f64 python bench.py --workload helmet --steps 6 --warmup 1 --no-cpu-baseline --also none --in-flight 64
run helmet_f1 python bench.py --workload helmet --steps 2 --warmup 1 --no-cpu-baseline --also none --in-flight 1 --frames-per-step 64
