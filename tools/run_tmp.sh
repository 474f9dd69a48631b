run() { tag=$1; w=$2; shift 2; timeout 300 python bench.py --workload $w --no-cpu-baseline "$@" > gpurun_out/tmp_bench.json 2> gpurun_out/tmp_bench.err; python - "$tag $w" "$*" <<PY
import json,sys
try:
    d=json.load(open("gpurun_out/tmp_bench.json"))
    print(sys.argv[1], sys.argv[2], d["value"], d["frame_ms_device"], {k:v["ms_per_frame"] for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e, open("gpurun_out/tmp_bench.err").read()[-500:])
PY
}
for v in var_P5 var_P6; do
  unset MI_PT_LIB; if [ -n "$v" ]; then export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/$v/libmi_pt.so; fi
  run "${v:-product}" helmet --steps 4 --warmup 1

done
