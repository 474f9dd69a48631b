#!/bin/bash
# Round-3 GPU call U: which part of the helper change makes the zoo materials depend on the render mode?
cd "$(dirname "$0")/.."; ulimit -c 0
for v in new nouc base; do
  if [ $v = new ]; then unset MI_PT_LIB; else export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_$v/libmi_pt.so; fi
  for g in anisotropy retroreflection; do timeout 200 python tools/zoo_modes.py $v $g 2>&1 | grep "^ZOO"; done
done
python3 - <<'PY'
import numpy as np
for g in ("anisotropy", "retroreflection"):
    a, b, c = (np.load(f"/tmp/zoo/{v}_{g}_a.npy") for v in ("base", "new", "nouc"))
    print("ZOO", g, "base vs new max abs", float(np.abs(a - b).max()), "base vs nouc", float(np.abs(a - c).max()))
PY
unset MI_PT_LIB
timeout 600 python -m pytest tests/test_gpu_lobes.py -q -m gpu -x -k "zoo" 2>&1 | tail -30
