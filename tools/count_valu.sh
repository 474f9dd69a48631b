#!/bin/bash
# Prints the number of vector-ALU instructions (v_* minus the memory ones) of each probe kernel of tools/count_valu.hip.
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ivk_gltf_renderer_amd/csrc/device -Wno-unused-function -fno-hip-fp32-correctly-rounded-divide-sqrt -freciprocal-math -fapprox-func --cuda-device-only -S -o /tmp/count_valu.s tools/count_valu.hip || exit 1
awk '/^probe_[a-z_]*:/ {name=$1; v=0; m=0} /^[ \t]+v_/ {v++} /^[ \t]+(global_|flat_|buffer_|ds_|scratch_)/ {m++} /^[ \t]+s_endpgm/ {print name, "valu", v, "mem", m}' /tmp/count_valu.s
