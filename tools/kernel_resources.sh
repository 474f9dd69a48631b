#!/bin/bash
# Register / scratch / LDS use of every kernel of a device source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
cd "$(dirname "$0")/../vk_gltf_renderer_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -Idevice -Wno-unused-function $([ "${1:-pt_kernels}" = pt_kernels ] && echo -fno-hip-fp32-correctly-rounded-divide-sqrt -freciprocal-math -fapprox-func) -Rpass-analysis=kernel-resource-usage -c device/${1:-pt_kernels}.hip -o /dev/null 2>&1 \
 | awk '/Function Name:/ {name=$5} / VGPRs:/ {v=$4} /AGPRs:/ {a=$4} /ScratchSize/ {s=$5} /Occupancy/ {o=$5} /LDS Size/ {l=$6; print name, "vgpr", v, "agpr", a, "scratch", s, "occ", o, "lds", l}' \
 | c++filt | sed 's/pt::(anonymous namespace):://; s/(.*)//' | sort
