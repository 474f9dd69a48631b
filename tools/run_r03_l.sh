#!/bin/bash
# Round-3 GPU call L: 256 frames in flight on the glass workload (its volume random walks leave a long tail per batch), and on the atrium.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O
val() { python3 -c "
import json
try:
    j=json.loads(open('$2').read().strip().splitlines()[-1]); print('RESULT $1', j['value'], 'F', j['config']['frames_in_flight'], 'ms/frame', j['ms_per_frame'], 'parity', (j.get('parity') or {}).get('rel_l2'))
except Exception as e: print('RESULT $1 FAILED', e)"; tail -2 ${2%.json}.err | cut -c1-200; }
timeout 300 python bench.py --workload glass --steps 2 --warmup 1 --no-cpu-baseline --also none > $O/r03l_glass128.json 2> $O/r03l_glass128.err; val glass128 $O/r03l_glass128.json
timeout 400 python bench.py --workload glass --steps 2 --warmup 1 --also none --in-flight 256 --frames-per-step 512 > $O/r03l_glass256.json 2> $O/r03l_glass256.err; val glass256 $O/r03l_glass256.json
timeout 400 python bench.py --workload glass --denoise --steps 2 --warmup 1 --no-cpu-baseline --also none --in-flight 256 --frames-per-step 512 > $O/r03l_glass256d.json 2> $O/r03l_glass256d.err; val glass256_denoise $O/r03l_glass256d.json
timeout 400 python bench.py --workload atrium --steps 2 --warmup 1 --no-cpu-baseline --also none --in-flight 256 --frames-per-step 512 > $O/r03l_atrium256.json 2> $O/r03l_atrium256.err; val atrium256 $O/r03l_atrium256.json
timeout 400 python bench.py --workload helmet --steps 4 --warmup 1 --no-cpu-baseline --also none --in-flight 256 --frames-per-step 512 > $O/r03l_helmet256.json 2> $O/r03l_helmet256.err; val helmet256 $O/r03l_helmet256.json
