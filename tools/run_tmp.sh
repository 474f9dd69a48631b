run() { tag=$1; w=$2; shift 2; timeout -k 5 120 python bench.py --workload $w --no-cpu-baseline --also none "$@" > gpurun_out/tmp_bench.json 2> gpurun_out/tmp_bench.err; python - "$tag $w" "$*" <<PY
import json,sys
try:
    d=json.load(open("gpurun_out/tmp_bench.json"))
    print(sys.argv[1], d["value"], d["frame_ms_device"], d["config"]["frames_in_flight"], {k:v["ms_per_frame"] for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/tmp_bench.err").read()[-300:])
PY
}
for v in "" var_L14 var_L28 var_L36B8; do
  unset MI_PT_LIB; if [ -n "$v" ]; then export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/$v/libmi_pt.so; fi
  run "${v:-product}" atrium --steps 1 --warmup 1
done
