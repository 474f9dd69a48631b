#!/bin/bash
# Round-3 GPU call B: NEE-technique sort of the SIMPLE shade kernel, opaque classification of the alpha cut, converged atrium test.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    pf = j["per_frame"]; sec = max(1.0, pf["segments"] - pf["cameraPaths"])
    k = j["kernels"]
    print(f"RESULT {tag:14s} value {j['value']:9.2f} tris {j['config']['scene_triangles']:9d} nodes/ray {pf['nodesClosest']/sec:6.2f} tris/ray {pf['trisClosest']/sec:6.2f} "
          f"closest {k['trace_closest']['ms_per_frame']:.4f} shade {k['shade']['ms_per_frame']:.4f} shadow {k['trace_shadow']['ms_per_frame']:.4f} primary {k['trace_primary']['ms_per_frame']:.4f} first {k['shade_first']['ms_per_frame']:.4f}")
except Exception as e:
    print("RESULT", tag, "FAILED", e)
PY
}
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03b_gputest.txt 2>&1; echo "GPUTEST rc=$? $(tail -1 $O/r03b_gputest.txt)"; grep "converged parity" $O/r03b_gputest.txt
timeout 300 python -m pytest tests/test_gpu_lobes.py -m gpu -x -q -s -k converged 2>&1 | grep "converged parity\|passed\|failed" 
A="--workload atrium --steps 5 --warmup 1 --no-cpu-baseline --also none"
S="--workload street --steps 3 --warmup 1 --no-cpu-baseline --also none"
H="--workload helmet --steps 8 --warmup 1 --no-cpu-baseline --also none"
run() { tag=$1; shift; timeout 150 "$@" > $O/r03b_$tag.json 2>$O/r03b_$tag.err; summ $tag $O/r03b_$tag.json; }
run atrium_base python bench.py $A
MI_PT_SORT_SIMPLE=0 run atrium_nosort python bench.py $A
MI_PT_SORT_SIMPLE=1 run atrium_sort1 python bench.py $A
run atrium_cut8 python bench.py $A --alpha-cut 8
MI_PT_DIAG_NO_OPAQUE_TRIS=1 run atrium_cut8_noopq python bench.py $A --alpha-cut 8
run atrium_cut16 python bench.py $A --alpha-cut 16
MI_PT_DIAG_NO_OPAQUE_TRIS=1 run atrium_cut16_noopq python bench.py $A --alpha-cut 16
run street_base python bench.py $S
MI_PT_SORT_SIMPLE=0 run street_nosort python bench.py $S
run street_cut8 python bench.py $S --alpha-cut 8
run street_cut16 python bench.py $S --alpha-cut 16
run helmet_base python bench.py $H
MI_PT_SORT_SIMPLE=1 run helmet_sort1 python bench.py $H
for c in 4 8; do
MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_prof/libmi_pt.so timeout 120 python bench.py $A --alpha-cut $c > $O/r03b_atrium_prof$c.json 2> $O/r03b_atrium_prof$c.err; echo "PROFILE cut $c"; grep "trace profile" $O/r03b_atrium_prof$c.err | tail -4
done
tools/pmc.sh r03b_atrium_sort "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" --workload atrium --steps 2 --warmup 1 | grep k_shade
MI_PT_SORT_SIMPLE=0 tools/pmc.sh r03b_atrium_nosort "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" --workload atrium --steps 2 --warmup 1 | grep k_shade
