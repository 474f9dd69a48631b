#!/bin/bash
# Round-3 GPU call A: test suite, the default bench line, TRACE_PROFILE section split, threshold variants of the per-lane walk.
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
    print(f"RESULT {tag:10s} value {j['value']:9.2f} nodes/ray {pf['nodesClosest']/sec:6.2f} tris/ray {pf['trisClosest']/sec:6.2f} "
          f"shadow n/ray {pf['nodesShadow']/max(1,pf['shadowRays']):6.2f} closest_ms/frame {k['trace_closest']['ms_per_frame']:.4f} shade {k['shade']['ms_per_frame']:.4f} shadow {k['trace_shadow']['ms_per_frame']:.4f} primary {k['trace_primary']['ms_per_frame']:.4f} first {k['shade_first']['ms_per_frame']:.4f}")
except Exception as e:
    print("RESULT", tag, "FAILED", e)
PY
}
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03_gputest.txt 2>&1; echo "GPUTEST rc=$? $(tail -1 $O/r03_gputest.txt)"
timeout 600 python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err; echo "BENCH rc=$?"; cut -c1-600 $O/r03_bench_default.json
B="--workload atrium --steps 5 --warmup 1 --no-cpu-baseline --also none"
timeout 120 python bench.py $B > $O/r03a_atrium_base.json 2>/dev/null; summ base $O/r03a_atrium_base.json
for v in T12 T32 R8 R24 A16 B2; do
  MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_$v/libmi_pt.so timeout 120 python bench.py $B > $O/r03a_atrium_$v.json 2>/dev/null; summ $v $O/r03a_atrium_$v.json
done
for l in 1 3; do
  MI_PT_LEAF_TRIS=$l timeout 120 python bench.py $B > $O/r03a_atrium_leaf$l.json 2>/dev/null; summ leaf$l $O/r03a_atrium_leaf$l.json
done
MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_prof/libmi_pt.so timeout 120 python bench.py $B > $O/r03a_atrium_prof.json 2> $O/r03a_atrium_prof.err; grep "profile" $O/r03a_atrium_prof.err | tail -8
timeout 120 python bench.py --workload street --steps 3 --warmup 1 --no-cpu-baseline --also none > $O/r03a_street_base.json 2>/dev/null; summ street $O/r03a_street_base.json
