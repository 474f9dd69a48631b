#!/bin/bash
# Round-3 GPU call E: SAH-optimal collapse A/B, per-wave queue append A/B.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    pf = j["per_frame"]; sec = max(1.0, pf["segments"] - pf["cameraPaths"])
    k = j["kernels"]
    print(f"RESULT {tag:18s} value {j['value']:9.2f} nodes/ray {pf['nodesClosest']/sec:6.2f} tris/ray {pf['trisClosest']/sec:6.2f} "
          f"closest {k['trace_closest']['ms_per_frame']:.4f} shade {k['shade']['ms_per_frame']:.4f} shadow {k['trace_shadow']['ms_per_frame']:.4f} primary {k['trace_primary']['ms_per_frame']:.4f} first {k['shade_first']['ms_per_frame']:.4f}")
except Exception as e:
    print("RESULT", tag, "FAILED", e)
PY
}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "acceleration_structure or host_collapse" 2>&1 | grep "8-wide nodes\|passed\|failed\|Error" | cut -c1-300
run() { tag=$1; shift; timeout 200 "$@" > $O/r03e_$tag.json 2>$O/r03e_$tag.err; summ $tag $O/r03e_$tag.json; }
A="--workload atrium --steps 4 --warmup 1 --no-cpu-baseline --also none"
S="--workload street --steps 3 --warmup 1 --no-cpu-baseline --also none"
H="--workload helmet --steps 8 --warmup 1 --no-cpu-baseline --also none"
run atrium_greedy python bench.py $A
MI_PT_COLLAPSE=sah run atrium_sah python bench.py $A
MI_PT_COLLAPSE=sah MI_PT_LEAF_TRIS=3 run atrium_sah_leaf3 python bench.py $A
run street_greedy python bench.py $S
MI_PT_COLLAPSE=sah run street_sah python bench.py $S
MI_PT_COLLAPSE=sah MI_PT_LEAF_TRIS=3 run street_sah_leaf3 python bench.py $S
run helmet_greedy python bench.py $H
MI_PT_COLLAPSE=sah run helmet_sah python bench.py $H
MI_PT_COLLAPSE=sah MI_PT_LEAF_TRIS=3 run helmet_sah_leaf3 python bench.py $H
W=$PWD/vk_gltf_renderer_amd/lib/var_wavepush/libmi_pt.so
MI_PT_LIB=$W run helmet_wavepush python bench.py $H
MI_PT_LIB=$W run atrium_wavepush python bench.py $A
MI_PT_LIB=$W run street_wavepush python bench.py $S
run glass_base python bench.py --workload glass --steps 2 --warmup 1 --no-cpu-baseline --also none
MI_PT_LIB=$W run glass_wavepush python bench.py --workload glass --steps 2 --warmup 1 --no-cpu-baseline --also none
MI_PT_COLLAPSE=sah run glass_sah python bench.py --workload glass --steps 2 --warmup 1 --no-cpu-baseline --also none
