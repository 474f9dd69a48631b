#!/bin/bash
# Round-3 GPU call Z: interval node test of the packet walk (pt_packet.h): bit identity against the per-ray node test, A/B.
cd "$(dirname "$0")/.."; ulimit -c 0
O=$PWD/gpurun_out; mkdir -p $O
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    k = j["kernels"]
    print(f"RESULT {tag:18s} value {j['value']:9.2f} primary {k['trace_primary']['ms_per_frame']:.4f} shade_first {k['shade_first']['ms_per_frame']:.4f} nodesPrimary {j['per_frame'].get('nodesPrimary')} trisPrimary {j['per_frame'].get('trisPrimary')}")
except Exception as e:
    print("RESULT", tag, "FAILED", e)
PY
}
run() { tag=$1; shift; timeout 300 "$@" > $O/r03z_$tag.json 2>$O/r03z_$tag.err; summ $tag $O/r03z_$tag.json; }
N="--no-cpu-baseline --also none"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "packet_interval or frames_in_flight or packet_walk" 2>&1 | tail -15
for m in 1; do
  export MI_PT_PACKET_INTERVAL=$m
  run helmet_iv$m python bench.py --workload helmet --steps 6 --warmup 1 $N
  run atrium_iv$m python bench.py --workload atrium --steps 3 --warmup 1 $N
  run street_iv$m python bench.py --workload street --steps 2 --warmup 1 $N
  run helmet4k_iv$m python bench.py --workload helmet --width 3840 --height 2160 --in-flight 64 --steps 3 --warmup 1 $N
done
unset MI_PT_PACKET_INTERVAL
run helmet_f1 python bench.py --workload helmet --in-flight 1 --frames-per-step 64 --steps 3 --warmup 1 $N
