#!/bin/bash
# Builds a diagnostics variant of libmi_pt.so next to the product library: vk_gltf_renderer_amd/lib/var_<name>/libmi_pt.so, selected
# at run time with MI_PT_LIB=<path>.  Only pt_kernels.hip is recompiled with the extra flags; the other objects are the product's.
# usage: [VARIANT_SRC=bvh8] tools/build_variant.sh <name> <flags...>     e.g.  tools/build_variant.sh prof -DTRACE_PROFILE
#        (VARIANT_SRC: the device source that is recompiled with the flags, default pt_kernels)
#        tools/build_variant.sh NO_TEX -DMI_PT_DIAG_NO_TEX   (cost attribution, tools/attribution.sh; the images of such builds are wrong)
set -e
name=$1; shift
cd "$(dirname "$0")/../vk_gltf_renderer_amd/csrc"
make -s -j8
mkdir -p ../lib/var_$name
src=${VARIANT_SRC:-pt_kernels}
fp=""; [ "$src" = pt_kernels ] && fp="-fno-hip-fp32-correctly-rounded-divide-sqrt -freciprocal-math -fapprox-func"  # (csrc/Makefile: PT_KERNELS_FP)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I../../include -Idevice -Wno-unused-function $fp "$@" -c -o build/variant_$name.o device/$src.hip
objs=$(ls build/*.o | grep -v "build/$src.o" | grep -v "build/variant_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../lib/var_$name/libmi_pt.so build/variant_$name.o $objs
echo "built vk_gltf_renderer_amd/lib/var_$name/libmi_pt.so"
