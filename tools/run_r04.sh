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
  fourth)  # where the walks' time goes on the current build (section timers), what the alpha tests still cost, the new GPU test
    timeout 300 python -m pytest tests/test_alpha_cut.py -m gpu -x -q 2>&1 | tail -3
    B="--workload atrium --steps 3 --warmup 1 $N"
    MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/var_prof/libmi_pt.so timeout 150 python bench.py $B > $O/r04d_atrium_prof.json 2> $O/r04d_atrium_prof.err; grep "profile" $O/r04d_atrium_prof.err | tail -8
    MI_PT_DIAG_IGNORE_ALPHA=1 timeout 150 python bench.py $B > $O/r04d_atrium_noalpha.json 2> /dev/null; val $O/r04d_atrium_noalpha.json atrium_ignore_alpha
    timeout 150 python bench.py $B --alpha-cut 0 > $O/r04d_atrium_nocut.json 2> /dev/null; val $O/r04d_atrium_nocut.json atrium_no_cut
    timeout 150 python bench.py $B > $O/r04d_atrium_base.json 2> /dev/null; val $O/r04d_atrium_base.json atrium_base ;;
  fifth)  # own-lane pipelined triangle tests against the triangle rounds (var_rounds): parity suite, then the A/B
    timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
    ab r04e atrium helmet glass street ;;
  sixth)  # the fixed cost of the alpha kernels (no alpha candidate at all) against their dynamic cost; the seed in the queue entry
    timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
    B="--workload atrium --steps 3 --warmup 1 $N"
    timeout 150 python bench.py $B > $O/r04f_atrium_base.json 2> /dev/null; val $O/r04f_atrium_base.json atrium_base
    MI_PT_DIAG_ALL_OPAQUE_TRIS=1 timeout 150 python bench.py $B > $O/r04f_atrium_allopaque.json 2> /dev/null; val $O/r04f_atrium_allopaque.json atrium_all_opaque_tris
    MI_PT_DIAG_IGNORE_ALPHA=1 timeout 150 python bench.py $B > $O/r04f_atrium_noalpha.json 2> /dev/null; val $O/r04f_atrium_noalpha.json atrium_ignore_alpha
    timeout 150 python bench.py --workload street --steps 3 --warmup 1 $N > $O/r04f_street_base.json 2> /dev/null; val $O/r04f_street_base.json street_base ;;
  seventh)  # path state in the queue entry against state by slot (MI_PT_STATE_BY_SLOT=1, same library)
    timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
    for w in atrium helmet glass street; do
      timeout 150 python bench.py --workload $w --steps 3 --warmup 1 $N > $O/r04g_${w}_queue.json 2> /dev/null; val $O/r04g_${w}_queue.json ${w}_state_in_queue
      MI_PT_STATE_BY_SLOT=1 timeout 150 python bench.py --workload $w --steps 3 --warmup 1 $N > $O/r04g_${w}_slot.json 2> /dev/null; val $O/r04g_${w}_slot.json ${w}_state_by_slot
    done ;;
  tests) timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ;;
  ab) shift; ab "$@" ;;
  *) echo "unknown step $1" ;;
esac
