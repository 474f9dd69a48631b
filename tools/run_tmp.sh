timeout 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lobes.py -m gpu -q -x 2>&1 | tail -5
for w in glass atrium helmet; do
timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_${w}_c.json 2> gpurun_out/r02_bench_${w}_c.err; python - <<PY
import json
d=json.load(open("gpurun_out/r02_bench_${w}_c.json"))
print("$w", d["value"], d["frame_ms_device"], {k:v["ms_per_frame"] for k,v in d["kernels"].items()})
PY
done
MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_prof/libmi_pt.so timeout 300 python tools/diag_spans.py glass 32 2>&1 | grep profile
MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_prof/libmi_pt.so timeout 300 python tools/diag_spans.py atrium 32 2>&1 | grep profile
