run() { tag=$1; w=$2; shift 2; timeout 300 python bench.py --workload $w --no-cpu-baseline --also none "$@" > gpurun_out/tmp_bench.json 2> gpurun_out/tmp_bench.err; python - "$tag $w" "$*" <<PY
import json,sys
try:
    d=json.load(open("gpurun_out/tmp_bench.json"))
    pf=d["per_frame"]
    print(sys.argv[1], d["value"], d["frame_ms_device"], {k:v["ms_per_frame"] for k,v in d["kernels"].items()}, "nodes/ray", round(pf["nodesClosest"]/max(pf["segments"]-pf["cameraPaths"],1),2), "tris/ray", round(pf["trisClosest"]/max(pf["segments"]-pf["cameraPaths"],1),2), "build_s", d["scene_build_s"])
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/tmp_bench.err").read()[-300:])
PY
}
for v in "" var_R32 var_R64 var_R128; do
  unset MI_PT_LIB; if [ -n "$v" ]; then export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/$v/libmi_pt.so; fi
  run "${v:-product}" atrium --steps 1 --warmup 1
  run "${v:-product}" street --steps 1 --warmup 1 --frames-per-step 64
  run "${v:-product}" helmet --steps 3 --warmup 1
done
