run() { w=$1; shift; timeout 300 python bench.py --workload $w --no-cpu-baseline "$@" > gpurun_out/tmp_bench.json 2> gpurun_out/tmp_bench.err; python - "$w" "$*" <<PY
import json,sys
try:
    d=json.load(open("gpurun_out/tmp_bench.json"))
    print(sys.argv[1], sys.argv[2], d["value"], d["frame_ms_device"], {k:v["ms_per_frame"] for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e, open("gpurun_out/tmp_bench.err").read()[-500:])
PY
}
run glass --steps 2 --warmup 1 --in-flight 64 --frames-per-step 192
run glass --steps 2 --warmup 1 --in-flight 128 --frames-per-step 256
run helmet --steps 6 --warmup 1 --in-flight 64 --frames-per-step 192
run helmet --steps 6 --warmup 1 --in-flight 96 --frames-per-step 192
run atrium --steps 2 --warmup 1 --in-flight 64 --frames-per-step 192
