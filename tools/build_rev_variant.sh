#!/bin/bash
# Builds libmi_pt.so of another git revision next to the product library (vk_gltf_renderer_amd/lib/var_<name>/libmi_pt.so, selected at
# run time with MI_PT_LIB=<path>), so that one GPU call can A/B the working tree against it (tools/run_gpu.sh ab).
# usage: tools/build_rev_variant.sh <name> <git revision>
set -e
name=$1; rev=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/mi_pt_rev_$name
rm -rf $tmp; mkdir -p $tmp
git -C $root archive $rev vk_gltf_renderer_amd/csrc include | tar -x -C $tmp
make -s -j8 -C $tmp/vk_gltf_renderer_amd/csrc $tmp/vk_gltf_renderer_amd/lib/libmi_pt.so ROOT=$tmp > $tmp/build.log 2>&1 || { tail -20 $tmp/build.log; exit 1; }
mkdir -p $root/vk_gltf_renderer_amd/lib/var_$name
cp $tmp/vk_gltf_renderer_amd/lib/libmi_pt.so $root/vk_gltf_renderer_amd/lib/var_$name/libmi_pt.so
echo "built vk_gltf_renderer_amd/lib/var_$name/libmi_pt.so from $rev"
