#!/bin/bash
# Round-3 GPU call F: two-stage shade A/B (fused / staged at 4 waves / staged at 3 waves), bit-identity of the staged image.
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
python - <<'PY'
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import parity_util as pu
from vk_gltf_renderer_amd import scenegen
hdr = os.path.join("assets", "std_env.hdr")
for name, path, kw in (("helmet", scenegen.scene_helmet_class("/tmp/h.glb", seed=3, tess=48, tex_size=256), dict(hdr_path=hdr, max_depth=6)),
                       ("atrium", scenegen.scene_atrium_class("/tmp/a.glb", seed=5, detail=0.15, tex_size=64), dict(max_depth=8))):
    s = pu.Setup(path, 200, 120, **kw)
    a = pu.render_gpu(s, 4, in_flight=2)
    os.environ["MI_PT_FUSED_SHADE"] = "1"
    b = pu.render_gpu(s, 4, in_flight=2)
    del os.environ["MI_PT_FUSED_SHADE"]
    same = (a["accum"] == b["accum"]).all()
    m = pu.compare_images(b["accum"], a["accum"])
    print("STAGED-vs-FUSED", name, "bit-identical", bool(same), "rel_l2", m["rel_l2"], "exact", m["frac_exact"], {k: (a["stats"][k], b["stats"][k]) for k in ("segments", "surfaceHits", "shadowRays", "textureTaps")})
    o = pu.render_oracle(s, 4)
    print("  staged vs oracle", pu.compare_images(o["accum"], a["accum"])["rel_l2"], "fused vs oracle", pu.compare_images(o["accum"], b["accum"])["rel_l2"])
PY
run() { tag=$1; shift; timeout 200 "$@" > $O/r03f_$tag.json 2>$O/r03f_$tag.err; summ $tag $O/r03f_$tag.json; }
A="--workload atrium --steps 3 --warmup 1 --no-cpu-baseline --also none"
H="--workload helmet --steps 6 --warmup 1 --no-cpu-baseline --also none"
S="--workload street --steps 2 --warmup 1 --no-cpu-baseline --also none"
V=$PWD/vk_gltf_renderer_amd/lib/var_stage3/libmi_pt.so
MI_PT_FUSED_SHADE=1 run helmet_fused python bench.py $H
run helmet_staged4 python bench.py $H
MI_PT_LIB=$V run helmet_staged3 python bench.py $H
MI_PT_FUSED_SHADE=1 run atrium_fused python bench.py $A
run atrium_staged4 python bench.py $A
MI_PT_LIB=$V run atrium_staged3 python bench.py $A
MI_PT_FUSED_SHADE=1 run street_fused python bench.py $S
run street_staged4 python bench.py $S
