#!/bin/bash
# Native A/B of MI_PT_REINSERT through the headless app (no Python start-up: fits a GPU call of seconds): atrium 1080p depth 12 F=128, street 4K depth 8 F=64.
# Scenes: bench.py's generated stand-ins copied to tools/_scratch/reinsert/{atrium,street}.glb (python -c 'import bench; print(bench.scene_path("atrium", 0))').
# Read the per-batch times off the HEADLESS_PROGRESS lines: the first batches of every process after the first carry one-off costs.
L=vk_gltf_renderer_amd/lib; S=tools/_scratch/reinsert
run() { # tag passes scene w h depth frames inflight
  echo "== $1 MI_PT_REINSERT=$2"
  MI_PT_BUILD_TIMING=1 MI_PT_REINSERT=$2 timeout 40 $L/mi_gltf_renderer --headless --size $4 $5 --scenefile $S/$3 --frames $7 --maxFrames $7 --framesInFlight $8 --ptMaxDepth $6 --ptSamples 1 --useOpacityMicromap 1 --alphaCut 4 $C 2>&1 | grep -E "HEADLESS_BATCH|HEADLESS_SUMMARY|HEADLESS_COUNTERS|reinsert|error|failed"
}
run atrium 0 atrium.glb 1920 1080 12 641 128
run atrium 24 atrium.glb 1920 1080 12 641 128
run street 0 street.glb 3840 2160 8 257 64
run street 16 street.glb 3840 2160 8 257 64
# the same with the traversal counters (slower kernels): node visits per ray before / after
C="--ptCounters 1"
run atrium 0 atrium.glb 1920 1080 12 129 128
run atrium 24 atrium.glb 1920 1080 12 129 128
