#!/bin/bash
# Round-3 GPU call G: technique-keyed queue append A/B + the full GPU suite.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    k = j["kernels"]
    print(f"RESULT {tag:18s} value {j['value']:9.2f} closest {k['trace_closest']['ms_per_frame']:.4f} shade {k['shade']['ms_per_frame']:.4f} shadow {k['trace_shadow']['ms_per_frame']:.4f} primary {k['trace_primary']['ms_per_frame']:.4f} first {k['shade_first']['ms_per_frame']:.4f}")
except Exception as e:
    print("RESULT", tag, "FAILED", e)
PY
}
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03g_gputest.txt 2>&1; echo "GPUTEST rc=$? $(tail -1 $O/r03g_gputest.txt)"
run() { tag=$1; shift; timeout 200 "$@" > $O/r03g_$tag.json 2>$O/r03g_$tag.err; summ $tag $O/r03g_$tag.json; }
A="--workload atrium --steps 3 --warmup 1 --no-cpu-baseline --also none"
H="--workload helmet --steps 6 --warmup 1 --no-cpu-baseline --also none"
S="--workload street --steps 2 --warmup 1 --no-cpu-baseline --also none"
MI_PT_QUEUE_KEY=0 run atrium_nokey python bench.py $A
run atrium_key python bench.py $A
MI_PT_QUEUE_KEY=0 run street_nokey python bench.py $S
run street_key python bench.py $S
MI_PT_QUEUE_KEY=0 run helmet_nokey python bench.py $H
run helmet_key python bench.py $H
tools/pmc.sh r03g_atrium_key "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" --workload atrium --steps 1 --warmup 1 | grep "k_shade<false\|k_trace_closest<true, true, false\|k_trace_shadow<true, 1, false"
MI_PT_QUEUE_KEY=0 tools/pmc.sh r03g_atrium_nokey "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" --workload atrium --steps 1 --warmup 1 | grep "k_shade<false\|k_trace_closest<true, true, false\|k_trace_shadow<true, 1, false"
