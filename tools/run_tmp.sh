timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
run() { tag=$1; w=$2; shift 2; timeout -k 5 150 python bench.py --workload $w --no-cpu-baseline --also none "$@" > gpurun_out/tmp_bench.json 2> gpurun_out/tmp_bench.err; python - "$tag $w" "$*" <<PY
import json,sys
try:
    d=json.load(open("gpurun_out/tmp_bench.json"))
    print(sys.argv[1], d["value"], d["frame_ms_device"], d["config"]["frames_in_flight"], {k:v["ms_per_frame"] for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/tmp_bench.err").read()[-300:])
PY
}
run prefetch helmet --steps 6 --warmup 1
run prefetch atrium --steps 1 --warmup 1
