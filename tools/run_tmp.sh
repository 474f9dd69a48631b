timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "transmissive or glass" 2>&1 | tail -3
run() { tag=$1; w=$2; shift 2; timeout 400 python bench.py --workload $w --no-cpu-baseline --also none "$@" > gpurun_out/tmp_bench.json 2> gpurun_out/tmp_bench.err; python - "$tag $w" "$*" <<PY
import json,sys
try:
    d=json.load(open("gpurun_out/tmp_bench.json"))
    print(sys.argv[1], d["value"], d["frame_ms_device"], d["config"]["frames_in_flight"], {k:v["ms_per_frame"] for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/tmp_bench.err").read()[-300:])
PY
}
run chunked glass --steps 2 --warmup 1
