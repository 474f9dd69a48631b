#!/bin/bash
# A/B of library builds on the GPU box: benches every vk_gltf_renderer_amd/lib/var_*/libmi_pt.so next to the product build.
for v in "" $(cd vk_gltf_renderer_amd/lib && ls -d var_* 2>/dev/null); do
  if [ -n "$v" ]; then export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/$v/libmi_pt.so; fi
  a=$(timeout 80 python bench.py --no-cpu-baseline --steps 8 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  b=$(timeout 80 python bench.py --no-cpu-baseline --steps 8 --workload atrium 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  echo "VARIANT ${v:-default} helmet $a atrium $b"
done
