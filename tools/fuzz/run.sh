#!/bin/bash
# Builds the sanitizer driver and runs the fuzz campaign (CPU only, ~15 minutes).  usage: tools/fuzz/run.sh
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=/tmp/mi_fuzz; mkdir -p $out/seeds
cd $root/vk_gltf_renderer_amd/csrc
g++ -O1 -g -std=c++17 -fwrapv -fsanitize=address,undefined -fno-omit-frame-pointer -I$root/include -Ihost -o $out/driver $root/tools/fuzz/fuzz_driver.cpp \
    host/gltf_scene.cpp host/gltf_scene_animation.cpp host/alpha_cut.cpp host/image_loader.cpp host/jpeg_decoder.cpp host/dds_decoder.cpp host/bc7_decoder.cpp \
    host/ktx_decoder.cpp host/mikktspace_tangents.cpp host/meshopt_decoder.cpp host/mi_host.cpp -lz -ldl
python $root/tools/fuzz/make_seeds.py $out/seeds
python $root/tools/fuzz/fuzz_host.py $out/driver $out/seeds/*.glb $root/assets/Box.glb
