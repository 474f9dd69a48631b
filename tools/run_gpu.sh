#!/bin/bash
# The GPU runner (rounds 4 and 5), one gpurun call per step:  gpurun -- 'tools/run_gpu.sh <step> [args]'
#   tests                      the GPU suite
#   line                       the full default bench line
#   ab <tag> <workloads...>    every vk_gltf_renderer_amd/lib/var_*/libmi_pt.so (tools/build_variant.sh, tools/build_rev_variant.sh) next to the product build
#   evidence <tag> <args...>   kernel-trace stats + counter passes of one configuration -> pmc_latest_<tag>.json (copy to profiles/)
#   timeline <tag> <args...>   per-bounce launch durations of one configuration (rocprofv3 --kernel-trace, tools/launch_timeline.py)
#   sweep                      frames in flight 1 / 8 / 64 / 128 at 1080p and 4K with device memory
#   mbvalu                     tools/microbench_valu.hip (build it first: hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o tools/_scratch/mb_valu)
cd "$(dirname "$0")/.."; ulimit -c 0
O=$PWD/gpurun_out; mkdir -p $O
N="--no-cpu-baseline --also none"
val() { python3 -c "
import json,sys
j=json.loads([l for l in open('$1'.replace('.json','.err')) if l.startswith('{')][-1]); k=j['kernels']  # (the full record goes to stderr; stdout ends with the compact line)
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
envab() { # envab <tag> <VAR> "<values...>" <workloads...>: the product build with VAR set to each value (run-time switches), --steps 3
  tag=$1; var=$2; values=$3; shift 3
  for v in $values; do
    for w in "$@"; do
      env $var=$v timeout 200 python bench.py --workload $w --steps 3 --warmup 1 $N > $O/${tag}_${w}_$v.json 2> $O/${tag}_${w}_$v.err && val $O/${tag}_${w}_$v.json ${w}_${var}=$v || { echo "FAILED ${w}_$v"; tail -3 $O/${tag}_${w}_$v.err; }
    done
  done
}
case "$1" in
  envab) shift; envab "$@" ;;
  line)  # the full default line (headline + every other configuration with CPU legs), as the driver runs it
    timeout 1500 python bench.py --steps 20 --warmup 5 > $O/r06_bench_default_a.json 2> $O/r06_bench_default_a.err; echo "bench rc $?"; tail -c 600 $O/r06_bench_default_a.err
    python3 - <<'PY'
import json
j=json.loads(open('gpurun_out/r06_bench_default_a.json').read().strip().splitlines()[-1])
print('atrium', j['value'], j.get('parity'), j.get('cpu_baseline'))
for n,a in j.get('also',{}).items(): print(n, a.get('value'), a.get('parity_rel_l2'), a.get('cpu_baseline'))
PY
    ;;
  evidence)  # tools/run_gpu.sh evidence <tag> <bench args...>: kernel-trace stats + counter passes of ONE configuration (tag = the pmc file's name:
             # a workload of bench.py, or helmet_4k); leaves gpurun_out/r06_<tag>_kernel_stats.csv, r06_<tag>_pmc_summary.json and pmc_latest_<tag>.json (copy to profiles/)
    shift; tag=$1; shift
    tools/profile.sh r06_$tag "$@" --steps 2 --warmup 1 > /dev/null 2>&1
    python tools/summarize_pmc.py $O/prof_r06_$tag $O/r06_${tag}_pmc_summary.json > /dev/null
    cp "$(find $O/prof_r06_$tag/stats -name '*kernel_stats.csv' | head -1)" $O/r06_${tag}_kernel_stats.csv
    python3 - $tag <<'PY'
import json, subprocess, sys
tag = sys.argv[1]
line = [l for l in open(f"gpurun_out/prof_r06_{tag}/stats.log") if l.startswith("{")][-1]
j = json.loads(line); c = j["config"]
wl = "helmet" if tag == "helmet_4k" else tag
out = subprocess.run([sys.executable, "tools/make_pmc_latest.py", f"gpurun_out/r06_{tag}_pmc_summary.json", wl, str(c["frames_in_flight"]), "6",
                      "profiles/r03_fetch_calibration.json", str(c["resolution"][0]), str(c["resolution"][1])], capture_output=True, text=True)
open(f"gpurun_out/pmc_latest_{tag}.json", "w").write(out.stdout)
k = json.loads(out.stdout)["kernels"]
print("EVIDENCE", tag, j["value"], {n: (v.get("issue_frac"), v.get("active_lanes"), round(v["hbm_bytes_per_launch"] / 1e9, 2)) for n, v in k.items()}, out.stderr[-300:])
PY
    rm -rf $O/prof_r06_$tag  # (the per-dispatch counter CSVs are tens of MB per configuration; gpurun copies back at most 64 MiB)
    ;;
  sweep)  # frames in flight 1 / 8 / 64 / 128 at 1080p and 4K: what a maintainer gets per onRender batch size, and the memory it takes; then the same
          # through ONE mi_pt_render_frame CALL PER FRAME with mi_pt_set_frame_queue (bench.py --frame-queue)
    pr() { python3 -c "
import json,sys; j=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', j['config']['frames_in_flight'], j['value'], j['device_memory_GB'])"; }
    for w in helmet atrium; do for f in 1 8 64 128; do
      timeout 200 python bench.py --workload $w --in-flight $f --exact-in-flight --frames-per-step $((f * 2)) --steps 4 --warmup 1 --no-uncut $N > $O/r06_sweep_${w}_f$f.json 2> /dev/null
      pr $O/r06_sweep_${w}_f$f.json "SWEEP ${w} 1080p render_frames in_flight"
    done; done
    for f in 1 8 64; do
      timeout 200 python bench.py --workload helmet --width 3840 --height 2160 --in-flight $f --exact-in-flight --frames-per-step $((f * 2)) --steps 4 --warmup 1 --no-uncut $N > $O/r06_sweep_helmet4k_f$f.json 2> /dev/null
      pr $O/r06_sweep_helmet4k_f$f.json "SWEEP helmet 4K render_frames in_flight"
    done
    for w in helmet atrium; do for f in 1 8 64; do
      timeout 200 python bench.py --workload $w --frame-queue $f --exact-in-flight --frames-per-step $((f * 2)) --steps 4 --warmup 1 --no-uncut $N > $O/r06_sweepq_${w}_f$f.json 2> /dev/null
      pr $O/r06_sweepq_${w}_f$f.json "SWEEP ${w} 1080p render_frame x1 + frame_queue"
    done; done ;;
  overlap)  # MI_PT_OVERLAP: the shadow stage of small batches on a second stream -- on (default 16 frames) against off, 1 / 4 / 8 / 16 frames in flight
    for w in ${OVERLAP_WORKLOADS:-helmet atrium}; do for f in ${OVERLAP_FRAMES:-1 4 8 16}; do for o in 0 1024; do
      MI_PT_OVERLAP=$o timeout 200 python bench.py --workload $w --in-flight $f --frames-per-step $((f * 4 > 256 ? 256 : f * 4)) --steps 4 --warmup 1 $N > $O/r04_overlap_${w}_f${f}_o$o.json 2> /dev/null
      python3 -c "
import json; j=json.loads(open('$O/r04_overlap_${w}_f${f}_o$o.json').read().strip().splitlines()[-1]); print('OVERLAP ${w} in_flight $f overlap_up_to $o', j['value'])"
    done; done; done ;;
  timeline)  # tools/run_gpu.sh timeline <tag> <bench args...>: per-dispatch kernel trace of one configuration -> per-bounce launch durations (tools/launch_timeline.py)
    shift; tag=$1; shift
    ( cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$tag -- python $OLDPWD/bench.py "$@" --no-uncut $N > $O/tl_$tag.log 2>&1 )
    python3 tools/launch_timeline.py $O/tl_$tag ${SHORT_US:-200} | tee $O/r06_timeline_$tag.txt
    grep -h -o 'k_[a-z_]*<[^>]*>' $O/tl_$tag/*/*kernel_trace.csv | sort | uniq -c | sort -rn | head -12; rm -rf $O/tl_$tag ;;
  mbvalu) timeout 200 tools/_scratch/mb_valu > $O/r04_mb_valu.txt 2>&1; grep "waves/SIMD=4" $O/r04_mb_valu.txt | cut -c1-20,80-160 ;;
  tests) timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ;;
  ab) shift; ab "$@" ;;
  *) echo "unknown step $1" ;;
esac
