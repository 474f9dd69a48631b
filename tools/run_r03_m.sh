#!/bin/bash
# Round-3 GPU call M: escaped camera paths evaluated in the packet kernel (pixel-major layout) A/B + full GPU suite.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    k = j["kernels"]; listed = sum(v["ms_per_frame"] for v in k.values())
    print(f"RESULT {tag:18s} value {j['value']:9.2f} closest {k['trace_closest']['ms_per_frame']:.4f} shade {k['shade']['ms_per_frame']:.4f} shadow {k['trace_shadow']['ms_per_frame']:.4f} primary {k['trace_primary']['ms_per_frame']:.4f} first {k['shade_first']['ms_per_frame']:.4f} other {j['frame_ms_device']-listed:.4f}")
except Exception as e:
    print("RESULT", tag, "FAILED", e)
PY
}
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03m_gputest.txt 2>&1; echo "GPUTEST rc=$? $(tail -1 $O/r03m_gputest.txt)"
run() { tag=$1; shift; timeout 300 "$@" > $O/r03m_$tag.json 2>$O/r03m_$tag.err; summ $tag $O/r03m_$tag.json; }
N="--no-cpu-baseline --also none"
V=$PWD/vk_gltf_renderer_amd/lib/var_missfinish/libmi_pt.so
run helmet python bench.py --workload helmet --steps 6 --warmup 1 $N
MI_PT_LIB=$V run helmet_old python bench.py --workload helmet --steps 6 --warmup 1 $N
run helmet4k python bench.py --workload helmet --width 3840 --height 2160 --steps 3 --warmup 1 $N
MI_PT_LIB=$V run helmet4k_old python bench.py --workload helmet --width 3840 --height 2160 --steps 3 --warmup 1 $N
run glass python bench.py --workload glass --steps 1 --warmup 1 $N
MI_PT_LIB=$V run glass_old python bench.py --workload glass --steps 1 --warmup 1 $N
run atrium python bench.py --workload atrium --steps 3 --warmup 1 $N
run street python bench.py --workload street --steps 2 --warmup 1 $N
MI_PT_LIB=$V run street_old python bench.py --workload street --steps 2 --warmup 1 $N
