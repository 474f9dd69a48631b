#!/bin/bash
# Round-3 GPU call K: occupancy variants of the shade and packet kernels under the pixel-major layout.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O
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
run() { tag=$1; shift; timeout 200 "$@" > $O/r03k_$tag.json 2>$O/r03k_$tag.err; summ $tag $O/r03k_$tag.json; }
H="--workload helmet --steps 6 --warmup 1 --no-cpu-baseline --also none"
A="--workload atrium --steps 3 --warmup 1 --no-cpu-baseline --also none"
run helmet_base python bench.py $H
run atrium_base python bench.py $A
for v in sw4 sw2 pw3 pw6 pw8; do
  MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_$v/libmi_pt.so run helmet_$v python bench.py $H
  MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_$v/libmi_pt.so run atrium_$v python bench.py $A
done
