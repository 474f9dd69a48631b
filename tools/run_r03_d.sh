#!/bin/bash
# Round-3 GPU call D: GPU tests, frames-in-flight sweep.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03d_gputest.txt 2>&1; echo "GPUTEST rc=$? $(tail -1 $O/r03d_gputest.txt)"
grep "converged parity" $O/r03d_gputest.txt
timeout 300 python -m pytest tests/test_gpu_lobes.py -m gpu -x -q -s -k converged_atrium 2>&1 | grep "converged parity\|passed\|failed" | cut -c1-400
val() { python3 -c "
import json,sys
try:
    j=json.loads(open('$2').read().strip().splitlines()[-1]); print('RESULT $1', j['value'], 'F', j['config']['frames_in_flight'], 'ms/frame', j['ms_per_frame'])
except Exception as e: print('RESULT $1 FAILED', e)"; }
for F in 64 96 128; do
  timeout 200 python bench.py --workload helmet --steps 8 --warmup 1 --no-cpu-baseline --also none --in-flight $F --frames-per-step $((F*3)) > $O/r03d_helmet_f$F.json 2>/dev/null; val helmet_f$F $O/r03d_helmet_f$F.json
  timeout 200 python bench.py --workload atrium --steps 3 --warmup 1 --no-cpu-baseline --also none --in-flight $F --frames-per-step $((F*3)) > $O/r03d_atrium_f$F.json 2>/dev/null; val atrium_f$F $O/r03d_atrium_f$F.json
done
