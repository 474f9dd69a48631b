#!/bin/bash
# Round-4 GPU calls (one gpurun each): tools/run_r04.sh <step> [args]
cd "$(dirname "$0")/.."; ulimit -c 0
O=$PWD/gpurun_out; mkdir -p $O
N="--no-cpu-baseline --also none"
val() { python3 -c "
import json,sys
j=json.loads(open('$1').read().strip().splitlines()[-1]); k=j['kernels']
print('RESULT $2', round(j['value'],1), ' '.join(f\"{n}={k[n]['ms_per_frame']:.4f}\" for n in k), 'visits', j.get('node_visits_per_secondary_ray'), 'tris', j.get('triangle_tests_per_secondary_ray'))"; }
ab() { # ab <tag> <workloads...>: every lib/var_* build next to the product build, --steps 3
  tag=$1; shift
  for v in base $(cd vk_gltf_renderer_amd/lib && ls -d var_* 2>/dev/null | sed s/var_//); do
    if [ $v = base ]; then unset MI_PT_LIB; else export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_$v/libmi_pt.so; fi
    for w in "$@"; do
      timeout 150 python bench.py --workload $w --steps 3 --warmup 1 $N > $O/${tag}_${w}_$v.json 2> $O/${tag}_${w}_$v.err && val $O/${tag}_${w}_$v.json ${w}_$v || { echo "FAILED ${w}_$v"; tail -3 $O/${tag}_${w}_$v.err; }
    done
  done
  unset MI_PT_LIB
}
case "$1" in
  first)  # VALU issue costs + the full default line
    timeout 120 tools/_scratch/mb_valu > $O/r04_mb_valu.txt 2>&1; grep "waves/SIMD=4" $O/r04_mb_valu.txt | cut -c1-110
    timeout 1500 python bench.py > $O/r04_bench_default_a.json 2> $O/r04_bench_default_a.err; echo "bench rc $?"; tail -c 600 $O/r04_bench_default_a.err
    python3 - <<'PY'
import json
j=json.loads(open('gpurun_out/r04_bench_default_a.json').read().strip().splitlines()[-1])
print('atrium', j['value'], j.get('parity',{}).get('by_spp'), j.get('cpu_baseline'))
for n,a in j.get('also',{}).items(): print(n, a['value'], a.get('parity',{}).get('by_spp'), a.get('cpu_baseline',{}).get('value'))
PY
    ;;
  second)  # cndmask variants of the VALU micro-benchmark; what eager triangle / alpha rounds do to the walks' counters; the GPU suite
    timeout 200 tools/_scratch/mb_valu > $O/r04_mb_valu2.txt 2>&1; grep "waves/SIMD=4" $O/r04_mb_valu2.txt | cut -c1-20,80-160
    ab r04b atrium
    timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ;;
  third)  # the leaf word (node visit 235 -> 205 vector instructions): GPU suite, then A/B against the previous commit's library
    timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
    ab r04c atrium helmet glass street ;;
  tests) timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ;;
  ab) shift; ab "$@" ;;
  *) echo "unknown step $1" ;;
esac
