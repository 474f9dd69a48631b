#!/bin/bash
# Round-3 last GPU call: the interval packet kernel at 7 waves per SIMD (72 VGPRs, 5-9 spilled) against 6 (78-84, none).
cd "$(dirname "$0")/.."; ulimit -c 0
O=$PWD/gpurun_out; mkdir -p $O
N="--no-cpu-baseline --also none"
for v in base w7; do
  if [ $v = base ]; then unset MI_PT_LIB; else export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_$v/libmi_pt.so; fi
  for w in helmet atrium; do
    timeout 120 python bench.py --workload $w --steps 3 --warmup 1 $N > $O/r03zz_${w}_$v.json 2>/dev/null
    python3 -c "
import json; j=json.loads(open('$O/r03zz_${w}_$v.json').read().strip().splitlines()[-1]); print('RESULT ${w}_$v', round(j['value'],1), j['kernels']['trace_primary']['ms_per_frame'])"
  done
done
