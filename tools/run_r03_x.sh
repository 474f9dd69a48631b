#!/bin/bash
# Round-3 GPU call X: the shade kernels read their own descriptors as constant memory (scalar loads after stores and calls too).
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
run() { tag=$1; shift; timeout 300 "$@" > $O/r03x_$tag.json 2>$O/r03x_$tag.err; summ $tag $O/r03x_$tag.json; }
N="--no-cpu-baseline --also none"
lib() { if [ $1 = new ]; then unset MI_PT_LIB; else export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_$1/libmi_pt.so; fi; }
for v in base new; do lib $v; timeout 300 python tools/ident_render.py $v 2>&1 | tail -1; done
python3 - <<'PY'
import numpy as np, glob, os
for f in sorted(glob.glob("/tmp/ident/base_*.npy")):
    a, b = np.load(f), np.load(f.replace("base_", "new_"))
    print("RESULT identity", os.path.basename(f), bool(np.array_equal(a, b)), "max abs", float(np.abs(a - b).max()))
PY
for v in base new; do
  lib $v
  run helmet_$v python bench.py --workload helmet --steps 6 --warmup 1 $N
  run atrium_$v python bench.py --workload atrium --steps 3 --warmup 1 $N
  run street_$v python bench.py --workload street --steps 2 --warmup 1 $N
done
for v in base new; do
  lib $v
  run glass_$v python bench.py --workload glass --steps 1 --warmup 1 $N
done
