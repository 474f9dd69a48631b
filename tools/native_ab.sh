#!/bin/bash
# A/B of environment switches and variant libraries through the headless app -- no Python, no torch: a GPU call of this costs seconds, not minutes
# (round 4's last two GPU-minutes paid for three of them).  Steady state is read off the HEADLESS_BATCH lines: median ms per batch after the first two
# batches (allocations, first touch; every process after the first of a call pays more there), as Msamples/s.
#
#   tools/native_ab.sh --prepare                       (here, no GPU) copies / generates bench.py's scenes into tools/_scratch/scenes/
#   gpurun -- 'tools/native_ab.sh atrium street -- base "MI_PT_REINSERT=16" "MI_PT_SHADOW_FAR_FIRST=1" "LIB=part" "COUNTERS=1 MI_PT_REINSERT=16"'
#       workloads before `--` (atrium street helmet glass), one run per quoted case after it:  ENV=VALUE ...   LIB=<name>: vk_gltf_renderer_amd/lib/var_<name>
#       COUNTERS=1: --ptCounters 1 and few frames (node visits per ray; slower kernels, no throughput figure)
cd "$(dirname "$0")/.."
S=tools/_scratch/scenes; L=$PWD/vk_gltf_renderer_amd/lib
if [ "$1" = "--prepare" ]; then
  mkdir -p $S
  python3 - <<'PY'
import os, shutil, sys
sys.path.insert(0, os.getcwd())
import bench
for w in ("atrium", "street", "helmet", "glass"):
    shutil.copy(bench.scene_path(w, 0), f"tools/_scratch/scenes/{w}.glb")
shutil.copy("assets/std_env.hdr", "tools/_scratch/scenes/std_env.hdr")
print(sorted(os.listdir("tools/_scratch/scenes")))
PY
  exit 0
fi
W=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do W+=("$1"); shift; done; shift
declare -A SIZE=([atrium]="1920 1080" [street]="3840 2160" [helmet]="1920 1080" [glass]="1920 1080")
declare -A DEPTH=([atrium]=12 [street]=8 [helmet]=8 [glass]=24) FLIGHT=([atrium]=128 [street]=64 [helmet]=128 [glass]=256) ENVSYS=([atrium]=0 [street]=0 [helmet]=1 [glass]=1)
for w in "${W[@]}"; do
  for c in "$@"; do
    lib=""; counters=0; envs=()
    for kv in $c; do
      case "$kv" in base) ;; LIB=*) lib=${kv#LIB=} ;; COUNTERS=1) counters=1 ;; *) envs+=("$kv") ;; esac
    done
    f=${FLIGHT[$w]}; frames=$((1 + 7 * f)); extra=""
    [ $counters = 1 ] && { frames=$((1 + 2 * f)); extra="--ptCounters 1"; }
    out=$(env "${envs[@]}" ${lib:+LD_LIBRARY_PATH=$L/var_$lib} timeout 120 $L/mi_gltf_renderer --headless --size ${SIZE[$w]} --scenefile $S/$w.glb --hdrfile $S/std_env.hdr \
          --envSystem ${ENVSYS[$w]} --frames $frames --maxFrames $frames --framesInFlight $f --ptMaxDepth ${DEPTH[$w]} --ptSamples 1 --useOpacityMicromap 1 --alphaCut 4 $extra 2>&1)
    echo "$out" | python3 -c "
import re, statistics, sys
t = sys.stdin.read(); w, case, size = sys.argv[1], sys.argv[2], sys.argv[3].split()
b = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r'HEADLESS_BATCH first_frame=\d+ frames=(\d+) ms=([0-9.]+)', t)]
steady = [ms / n for n, ms in b[3:]] or [ms / n for n, ms in b[1:]]
c = re.search(r'HEADLESS_COUNTERS.*', t)
if not b: print('NATIVE_AB', w, '|', case, '| FAILED:', t[-300:].replace(chr(10), ' '))
else: print('NATIVE_AB', w, '|', case, '| ms/frame %.4f | %.1f Msamples/s | batches' % (statistics.median(steady), int(size[0]) * int(size[1]) / statistics.median(steady) / 1e3), len(b), '|', (c.group(0)[18:] if c else ''),
            '|', ' '.join(l for l in t.splitlines() if 'reinsertion' in l))
" "$w" "$c" "${SIZE[$w]}"
  done
done
