#!/bin/bash
# Round-3 GPU call O: LDS split of the trace kernels: traversal-stack depth against cached BVH8 nodes.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    k = j["kernels"]
    print(f"RESULT {tag:18s} value {j['value']:9.2f} closest {k['trace_closest']['ms_per_frame']:.4f} shadow {k['trace_shadow']['ms_per_frame']:.4f}")
except Exception as e:
    print("RESULT", tag, "FAILED", e)
PY
}
run() { tag=$1; shift; timeout 300 "$@" > $O/r03o_$tag.json 2>$O/r03o_$tag.err; summ $tag $O/r03o_$tag.json; }
N="--no-cpu-baseline --also none"
for v in base st10 st8 st6; do
  if [ $v = base ]; then unset MI_PT_LIB; else export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_$v/libmi_pt.so; fi
  run atrium_$v python bench.py --workload atrium --steps 3 --warmup 1 $N
  run street_$v python bench.py --workload street --steps 2 --warmup 1 $N
  run helmet_$v python bench.py --workload helmet --steps 6 --warmup 1 $N
  run glass_$v python bench.py --workload glass --steps 1 --warmup 1 $N
done
