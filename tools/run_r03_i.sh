#!/bin/bash
# Round-3 GPU call I: pixel-major path slots: full GPU suite + the workloads, A/B against micro-tile major.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    k = j["kernels"]
    print(f"RESULT {tag:18s} value {j['value']:9.2f} closest {k['trace_closest']['ms_per_frame']:.4f} shade {k['shade']['ms_per_frame']:.4f} shadow {k['trace_shadow']['ms_per_frame']:.4f} primary {k['trace_primary']['ms_per_frame']:.4f} first {k['shade_first']['ms_per_frame']:.4f} total_dev {j['frame_ms_device']:.4f}")
except Exception as e:
    print("RESULT", tag, "FAILED", e)
PY
}
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03i_gputest.txt 2>&1; echo "GPUTEST rc=$? $(tail -1 $O/r03i_gputest.txt)"
run() { tag=$1; shift; timeout 200 "$@" > $O/r03i_$tag.json 2>$O/r03i_$tag.err; summ $tag $O/r03i_$tag.json; }
H="--workload helmet --steps 6 --warmup 1 --no-cpu-baseline --also none"
A="--workload atrium --steps 3 --warmup 1 --no-cpu-baseline --also none"
G="--workload glass --steps 2 --warmup 1 --no-cpu-baseline --also none"
run helmet_pixel python bench.py $H
MI_PT_MICROTILE_SLOTS=1 run helmet_mtile python bench.py $H
run atrium_pixel python bench.py $A
MI_PT_MICROTILE_SLOTS=1 run atrium_mtile python bench.py $A
run glass_pixel python bench.py $G
MI_PT_MICROTILE_SLOTS=1 run glass_mtile python bench.py $G
run street64 python bench.py --workload street --steps 1 --warmup 1 --no-cpu-baseline --also none --in-flight 64 --frames-per-step 64 --width 1920 --height 1080
MI_PT_MICROTILE_SLOTS=1 run street64_mtile python bench.py --workload street --steps 1 --warmup 1 --no-cpu-baseline --also none --in-flight 64 --frames-per-step 64 --width 1920 --height 1080
