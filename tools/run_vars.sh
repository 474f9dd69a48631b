timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for v in "" var_b4 var_b8l24 var_b4l16 var_b16l32; do
  if [ -n "$v" ]; then export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/$v/libmi_pt.so; fi
  a=$(timeout 80 python bench.py --no-cpu-baseline --steps 8 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  b=$(timeout 80 python bench.py --no-cpu-baseline --steps 8 --workload atrium 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  echo "VARIANT ${v:-default} helmet $a atrium $b"
done
MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/prof/libmi_pt.so timeout 100 python bench.py --no-cpu-baseline --workload atrium --steps 2 --warmup 1 2>&1 | grep -E "trace profile" | head -3 | cut -c1-400
