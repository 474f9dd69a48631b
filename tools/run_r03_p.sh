#!/bin/bash
# Round-3 GPU call P: texture prefetch of k_shade (MI_PT_TEX_PREFETCH 0 / 1 = first bounce / 3 = every bounce): A/B + bit identity.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    k = j["kernels"]
    print(f"RESULT {tag:18s} value {j['value']:9.2f} shade_first {k['shade_first']['ms_per_frame']:.4f} shade {k['shade']['ms_per_frame']:.4f} closest {k['trace_closest']['ms_per_frame']:.4f}")
except Exception as e:
    print("RESULT", tag, "FAILED", e)
PY
}
run() { tag=$1; shift; timeout 300 "$@" > $O/r03p_$tag.json 2>$O/r03p_$tag.err; summ $tag $O/r03p_$tag.json; }
N="--no-cpu-baseline --also none"
# bit identity: the same frames with and without the prefetch
cat > /tmp/ident.py <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import parity_util as pu, scenegen
os.makedirs("/tmp/ident", exist_ok=True)
hdr = os.path.join(os.getcwd(), "assets", "std_env.hdr")
paths = [scenegen.scene_helmet_class("/tmp/ident/helmet.glb", seed=7, tess=48, tex_size=256),
         scenegen.scene_material_zoo("/tmp/ident/zoo.glb", "texture_transform", tess=24),
         scenegen.scene_atrium_class("/tmp/ident/atrium.glb", seed=5, detail=0.2, tex_size=64)]
for k, p in enumerate(paths):
    s = pu.Setup(p, 320, 192, max_depth=4, hdr_path=hdr)
    g = pu.render_gpu(s, 4, in_flight=4)
    np.save(f"/tmp/ident/{sys.argv[1]}_{k}.npy", g["accum"])
print("rendered", sys.argv[1])
PY
for v in pf0 pf1 pf3; do
  if [ $v = pf1 ]; then unset MI_PT_LIB; else export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_$v/libmi_pt.so; fi
  timeout 300 python /tmp/ident.py $v 2>&1 | tail -1
done
python3 - <<'PY'
import numpy as np
for k in range(3):
    a = np.load(f"/tmp/ident/pf0_{k}.npy")
    for v in ("pf1", "pf3"):
        print("RESULT identity scene", k, v, bool(np.array_equal(a, np.load(f"/tmp/ident/{v}_{k}.npy"))))
PY
for v in pf0 pf1 pf3; do
  if [ $v = pf1 ]; then unset MI_PT_LIB; else export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_$v/libmi_pt.so; fi
  run helmet_$v python bench.py --workload helmet --steps 6 --warmup 1 $N
  run atrium_$v python bench.py --workload atrium --steps 3 --warmup 1 $N
  run street_$v python bench.py --workload street --steps 2 --warmup 1 $N
  run glass_$v python bench.py --workload glass --steps 1 --warmup 1 $N
done
unset MI_PT_LIB
run helmet4k_pf1 python bench.py --workload helmet --width 3840 --height 2160 --in-flight 64 --steps 3 --warmup 1 $N
MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_pf0/libmi_pt.so run helmet4k_pf0 python bench.py --workload helmet --width 3840 --height 2160 --in-flight 64 --steps 3 --warmup 1 $N
