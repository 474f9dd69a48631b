#!/bin/bash
# Round-3 GPU call N: hipGraph replay of small batches: full GPU suite (its small batches all run through graphs) + A/B at 1 / 2 / 4 / 8 frames in flight.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03n_gputest.txt 2>&1; echo "GPUTEST rc=$? $(tail -1 $O/r03n_gputest.txt)"
val() { python3 -c "
import json
try:
    j=json.loads(open('$2').read().strip().splitlines()[-1]); print('RESULT $1', j['value'], 'F', j['config']['frames_in_flight'], 'ms/frame', j['ms_per_frame'])
except Exception as e: print('RESULT $1 FAILED', e)"; }
for F in 1 2 4 8; do
  for G in 8 0; do
    MI_PT_GRAPH=$G timeout 200 python bench.py --workload helmet --in-flight $F --frames-per-step 64 --steps 4 --warmup 1 --no-cpu-baseline --also none > $O/r03n_helmet_f${F}_g$G.json 2>$O/r03n_helmet_f${F}_g$G.err; val helmet_f${F}_graph$G $O/r03n_helmet_f${F}_g$G.json
  done
done
for G in 8 0; do
  MI_PT_GRAPH=$G timeout 200 python bench.py --workload atrium --in-flight 1 --frames-per-step 16 --steps 2 --warmup 1 --no-cpu-baseline --also none > $O/r03n_atrium_f1_g$G.json 2>/dev/null; val atrium_f1_graph$G $O/r03n_atrium_f1_g$G.json
done
